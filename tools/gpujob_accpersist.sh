#!/bin/bash
# Round 4: msm_accumulate as a launch of k workgroups per CU that walk the chunks of 256 items themselves - by a fixed
# stride (CAPGPU_ACC_PERSISTENT=k) or from an atomic counter (+ CAPGPU_ACC_DYNAMIC=1) - against one workgroup per chunk
for cfg in "0 0" "3 1" "4 1" "6 1" "3 0" "0 0"; do
  set -- $cfg
  CAPGPU_ACC_PERSISTENT=$1 CAPGPU_ACC_DYNAMIC=$2 python bench.py --one-context --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('persistent $1 dynamic $2', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'accum' in a or 'segments' in a})"
done
