#!/bin/bash
# Instruction counters of the hot kernels (separate PMC passes, no trace domains): how many VALU instructions does a
# mixed addition really cost, and how busy are the SIMDs?
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/insts_$tag
for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
  name=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/insts_$tag/$name -- python3 bench.py --one-context --steps 1 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/insts_$tag/$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/insts_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", [0])[0])[:10]:
    print(k[:40], {c: (round(v[0] / v[1]), v[1]) for c, v in agg[k].items()})
PY
find gpurun_out/insts_$tag -name "*counter_collection.csv" -size +5M -delete
