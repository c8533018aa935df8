#!/bin/bash
# Item length of the small MSM launches once their tails run on quads: waves of work items per SIMD (CAPGPU_MSM_ITEM_WAVES)
# and the largest launch, in buckets, that takes the rule (CAPGPU_MSM_ITEM_SMALL_BUCKETS)  -> gpurun_out/itemwaves_ab.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
: > $OUT/itemwaves_ab.jsonl
for cfg in "1 8192" "2 8192" "3 8192" "1 20480" "2 20480" "3 20480" "1 8192"; do
  set -- $cfg
  CAPGPU_MSM_ITEM_WAVES=$1 CAPGPU_MSM_ITEM_SMALL_BUCKETS=$2 CAPGPU_AB_NAME="waves_$1_buckets_$2" timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/itemwaves_ab.jsonl
done
cat $OUT/itemwaves_ab.jsonl
