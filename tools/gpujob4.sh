#!/bin/bash
mkdir -p gpurun_out
for T in 11 10 9; do
CAPGPU_NTT_TILE_LOG=$T python bench.py --steps 3 --warmup 1 --batch 32 --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_T$T.json 2> gpurun_out/bench_T$T.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_T$T.json"))
print("T=$T value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2)); print({k:v for k,v in d["top_kernels_ms"].items() if k.startswith("ntt")})
PY
done
