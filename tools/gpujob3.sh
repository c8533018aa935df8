#!/bin/bash
mkdir -p gpurun_out
for P in 8 32 64; do
python bench.py --steps 3 --warmup 1 --batch $P --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_P$P.json 2> gpurun_out/bench_P$P.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_P$P.json"))
print("P=$P value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2)); print(d["top_kernels_ms"])
PY
tail -2 gpurun_out/bench_P$P.err
done
