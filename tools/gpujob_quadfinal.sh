#!/bin/bash
# msm_reduce_final on quads (CAPGPU_MSM_QUAD_FINAL): parity, then the batch-256 step and the single MSMs of 2^17 .. 2^24
# points with and without  -> gpurun_out/quadfinal_{0,1}.json
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_configs.py tests/test_gpu_plonk.py -x -q -m gpu 2>&1 | tail -3
for q in 0 1 0 1; do
  CAPGPU_MSM_QUAD_FINAL=$q timeout 900 python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-extras --no-mixed > $OUT/quadfinal_$q.json 2> $OUT/quadfinal_$q.err
  python - <<PY
import json
b = json.load(open("$OUT/quadfinal_$q.json"))
print("quad_final=$q", round(b["value"], 1), {m["log_n"]: round(m["ms"], 3) for m in b["roofline"].get("msm", [])}, {k: round(v, 2) for k, v in list(b["roofline"]["top_kernels_ms"].items())[:8]})
PY
done
