#!/bin/bash
# Closed-loop callers (64 threads x 8 single-proof calls) against the number of device contexts and the cut of a gathered
# batch, with each part's callers released when their part is done  -> gpurun_out/coalesce_ab3.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/coalesce_ab3.jsonl
for cfg in "2 4" "4 3" "4 4" "3 4" "6 4" "2 4" "4 4" "8 4"; do
  set -- $cfg
  CAPGPU_CONTEXTS_PER_DEVICE=$1 CAPGPU_COALESCE_SPLIT=$2 CAPGPU_AB_NAME="contexts_$1_split_$2" timeout 600 python tools/gpu_coalesce_ab.py --child 8 2>&1 | grep '^{' >> $OUT/coalesce_ab3.jsonl
done
cat $OUT/coalesce_ab3.jsonl
