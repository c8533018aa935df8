#!/bin/bash
# Round 4: segment length of the running-sum bucket reduction sized to the launch (same box A/B)
for st in 0 1 0 1; do
  CAPGPU_MSM_SEG_TUNE=$st python bench.py --one-context --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('seg_tune $st', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'accum' in a or 'reduce' in a})"
done
