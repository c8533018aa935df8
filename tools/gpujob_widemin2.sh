#!/bin/bash
# Narrow (c = 13) against wide (c = 15) table for small launches, now that both tails run on quads: CAPGPU_MSM_WIDE_MIN
#   -> gpurun_out/widemin2_ab.jsonl   (one context; batches 1 .. 5)
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
: > $OUT/widemin2_ab.jsonl
for w in 24 5 10 15 20 24; do
  CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_AB_BATCHES=1,2,3,4,5 CAPGPU_MSM_WIDE_MIN=$w CAPGPU_AB_NAME=wide_min_$w timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/widemin2_ab.jsonl
done
cat $OUT/widemin2_ab.jsonl
