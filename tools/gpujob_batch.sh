#!/bin/bash
mkdir -p gpurun_out
for b in "$@"; do
python bench.py --batch $b --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_b$b.json 2> gpurun_out/bench_b$b.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_b$b.json"))
print("batch $b", "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "kernel share", d["roofline"]["share_of_kernel_time"], d["top_kernels_ms"]["msm_accumulate"])
PY
tail -1 gpurun_out/bench_b$b.err
done
