#!/bin/bash
# msm_accumulate at four waves per SIMD (amdgpu_waves_per_eu(4,4): 128 VGPRs) against the default build (147 VGPRs, three
# waves): headline A/B, then the occupancy each really gets (SQ_WAVE_CYCLES x 4 / SQ_BUSY_CU_CYCLES = waves per busy CU)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
B="--steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm"
show='import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"],1), round(d["ms_per_step"],2), round(d["top_kernels_ms"]["msm_accumulate"]/d["steps"],2))'
for rep in 1 2; do
  timeout 300 python bench.py $B 2>/dev/null | python -c "$show" default
  CAPGPU_LIBRARY=$PWD/tools/libcapgpu_w4.so timeout 300 python bench.py $B 2>/dev/null | python -c "$show" waves4
done
mkdir -p gpurun_out/w4
for lib in default w4; do
  [ $lib = w4 ] && export CAPGPU_LIBRARY=$PWD/tools/libcapgpu_w4.so
  for ctr in SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVES; do
    timeout 240 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/w4/${lib}_$ctr -- python3 bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm > /dev/null 2> gpurun_out/w4/${lib}_$ctr.err
  done
done
python3 - <<'PY'
import csv, glob, collections
for lib in ("default", "w4"):
    tot = {}
    for ctr in ("SQ_WAVE_CYCLES", "SQ_BUSY_CU_CYCLES", "SQ_WAVES"):
        v = n = 0
        for f in glob.glob(f"gpurun_out/w4/{lib}_{ctr}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                if "msm_accumulate" in r["Kernel_Name"] and float(r["Counter_Value"]) > 1e8 or (ctr == "SQ_WAVES" and "msm_accumulate" in r["Kernel_Name"]):
                    v += float(r["Counter_Value"]); n += 1
        tot[ctr] = (v, n)
    w, b = tot["SQ_WAVE_CYCLES"], tot["SQ_BUSY_CU_CYCLES"]
    print(lib, tot, "waves per busy CU (x4):", 4 * w[0] / b[0] if b[0] else None)
PY
find gpurun_out/w4 -name "*.csv" -size +2M -delete
