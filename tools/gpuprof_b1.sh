#!/bin/bash
# What the latency-bound kernels of a single proof wait for: instruction-cache and wait counters of a batch-1 run
# (separate PMC passes, no trace domains).  tools/gpuprof_b1.sh TAG
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/b1pmc_$tag
for ctr in "SQ_INSTS_VALU SQ_WAVE_CYCLES" "SQC_ICACHE_REQ SQC_ICACHE_MISSES" "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_IFETCH SQ_BUSY_CYCLES" "SQ_WAVES SQ_WAIT_ANY"; do
  name=$(echo $ctr | tr ' ' '_')
  rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/b1pmc_$tag/$name -- python3 bench.py --batch 1 --steps 3 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/b1pmc_$tag/$name.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/b1pmc_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_WAVE_CYCLES", [0])[0])[:12]:
    print(k[:40], {c: (round(v[0] / v[1]), v[1]) for c, v in agg[k].items()})
PY
find gpurun_out/b1pmc_$tag -name "*counter_collection.csv" -size +5M -delete
