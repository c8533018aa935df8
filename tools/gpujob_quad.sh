#!/bin/bash
# Quad-lane tails of the small MSM launches (msm.hip: msm_*_quad): parity first, then the same-box A/B
#   CAPGPU_MSM_QUAD_MAX = 0 (one-lane tails) / 16 / 23   -> gpurun_out/quad_ab.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_plonk.py tests/test_gpu_graphs.py -x -q -m gpu 2>&1 | tail -5
: > $OUT/quad_ab.jsonl
for q in 0 16 23 0 16; do
  CAPGPU_MSM_QUAD_MAX=$q CAPGPU_AB_NAME=quad_max_$q timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/quad_ab.jsonl
done
cat $OUT/quad_ab.jsonl
