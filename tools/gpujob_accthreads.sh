#!/bin/bash
# msm_accumulate's workgroup size for the small launches (CAPGPU_ACC_THREADS) -> gpurun_out/accthreads_ab.jsonl
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
: > $OUT/accthreads_ab.jsonl
for w in 256 128 64 256 128; do
  CAPGPU_ACC_THREADS=$w CAPGPU_AB_BATCHES=1,2,4 CAPGPU_AB_NAME=acc_threads_$w timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/accthreads_ab.jsonl
done
cat $OUT/accthreads_ab.jsonl
