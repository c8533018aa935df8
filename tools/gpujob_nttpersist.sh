#!/bin/bash
# Round 4: NTT passes as persistent launches (k workgroups per CU taking tiles from an atomic counter), same box A/B
for k in 0 4 5 8 0; do
  CAPGPU_NTT_PERSISTENT=$k python bench.py --one-context --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('ntt persistent $k', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'ntt' in a or 'accum' in a})"
done
