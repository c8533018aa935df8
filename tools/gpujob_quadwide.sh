#!/bin/bash
# Quad tails on the wide table's launches (24 .. 63 MSMs): one context, batches of 5 .. 12 proofs
#   CAPGPU_MSM_QUAD_MAX_WIDE = 0 / 63   -> gpurun_out/quadwide_ab.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_plonk.py -x -q -m gpu 2>&1 | tail -3
: > $OUT/quadwide_ab.jsonl
for q in 0 63 40 0 63; do
  CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_AB_BATCHES=5,6,8,10,12,16 CAPGPU_MSM_QUAD_MAX_WIDE=$q CAPGPU_AB_NAME=quad_wide_$q timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/quadwide_ab.jsonl
done
cat $OUT/quadwide_ab.jsonl
