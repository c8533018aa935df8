#!/bin/bash
# Segment length of the running-sum reduction for launches of 131 .. 255 MSMs (batches of 27 .. 51 proofs), with the
# finish on quads: CAPGPU_MSM_SEG_TUNE  -> gpurun_out/segtune2_ab.jsonl   (one context)
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
: > $OUT/segtune2_ab.jsonl
for t in 0 1 0 1; do
  CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_AB_BATCHES=27,32,40,51,64 CAPGPU_MSM_SEG_TUNE=$t CAPGPU_AB_NAME=seg_tune_$t timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/segtune2_ab.jsonl
done
cat $OUT/segtune2_ab.jsonl
