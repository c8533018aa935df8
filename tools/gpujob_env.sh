#!/bin/bash
# bench under an environment override: tools/gpujob_env.sh TAG VAR=VALUE...
tag=$1; shift
mkdir -p gpurun_out
for kv in "$@"; do export "$kv"; done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("$tag", "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2)); print({k:round(v/3,1) for k,v in d["top_kernels_ms"].items()})
PY
for kv in "$@"; do unset "${kv%%=*}"; done
