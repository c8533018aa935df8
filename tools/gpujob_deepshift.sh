#!/bin/bash
# Round 4: the deep plan's window size and the shifted top window, same box (tools/gpujob_deepshift.sh)
#   c = 20 without / with the shift, c = 22 with / without it, single MSMs of 2^22 .. 2^24 points
OUT=gpurun_out/deepshift_r04.jsonl
: > $OUT
for cfg in "20 0" "20 1" "22 1" "22 0"; do
  set -- $cfg
  echo "{\"config\": {\"deep_c\": $1, \"top_shift\": $2}}" >> $OUT
  CAPGPU_MSM_DEEP_C=$1 CAPGPU_MSM_DEEP_SHIFT=$2 MINLOG=${MINLOG:-22} python tools/gpu_msm_deep_ab.py 24 >> $OUT 2>> gpurun_out/deepshift_r04.err
done
cat $OUT
