#!/bin/bash
# Instruction counters of a single 2^20-point MSM's kernels (why is msm_reduce_segments slow there?)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/msm20pmc; mkdir -p $O
for ctr in "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQ_WAVES SQ_BUSY_CYCLES"; do
  name=$(echo $ctr | tr ' ' '_')
  MSM_LOGS=${MSM_LOGS:-20} rocprofv3 --pmc $ctr --output-format csv -d $O/$name -- python3 tools/gpu_msm_profile.py > /dev/null 2> $O/$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("$O/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::", "").split("(")[0]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0])[0])[:6]:
    print(k[:40], {c: (round(v[0] / v[1]), v[1]) for c, v in agg[k].items()})
PY
