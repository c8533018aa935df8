#!/bin/bash
# Round 4: chunk size of the persistent msm_accumulate (workgroup = chunk = 256 / 128 / 64 items), same box
for cfg in "256 4" "128 8" "64 16" "256 6" "256 4"; do
  set -- $cfg
  CAPGPU_ACC_THREADS=$1 CAPGPU_ACC_PERSISTENT=$2 python bench.py --one-context --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-extras --msm-log-n 24 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('threads $1 per_cu $2', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'accum' in a}, [(m['log_n'], round(m['ms'],3)) for m in d['msm']])"
done
