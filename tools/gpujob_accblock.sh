#!/bin/bash
# A/B: workgroup size of msm_accumulate (dispatch granularity of its length-sorted items) on the headline bench
for t in 256 64 128 256 64; do
  CAPGPU_ACC_THREADS=$t timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('threads $t', round(d['value'],1), round(d['ms_per_step'],2), {k: round(v/d['steps'],2) for k,v in list(d['top_kernels_ms'].items())[:3]})"
done
