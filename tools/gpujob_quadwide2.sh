#!/bin/bash
# Where the quad tails hand over to the running sums: launches of 65 .. 160 MSMs on the wide table, one context
#   CAPGPU_MSM_QUAD_MAX_WIDE = 63 / 100 / 130 / 160   -> gpurun_out/quadwide2_ab.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/quadwide2_ab.jsonl
for q in 63 100 130 160 63; do
  CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_AB_BATCHES=13,16,20,24,32 CAPGPU_MSM_QUAD_MAX_WIDE=$q CAPGPU_AB_NAME=quad_wide_$q timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/quadwide2_ab.jsonl
done
cat $OUT/quadwide2_ab.jsonl
