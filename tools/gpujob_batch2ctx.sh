#!/bin/bash
# Round 4: the two-context headline at different step sizes (same box)
for b in 256 384 512 256; do
  python bench.py --batch $b --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
print('batch $b two_ctx', round(d['value'],1), 'one_ctx_profiled', round(d['one_context_profiled_pass']['proofs_per_s'],1), 'ms/step', round(d['ms_per_step'],2))"
done
