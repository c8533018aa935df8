#!/bin/bash
# Round 4: from how many MSMs per launch does the wide (c = 15) table pay, now that its 16384 buckets reduce through the
# grid form?  One-context rates at small batches for CAPGPU_MSM_WIDE_MIN = 32 (the round-1 choice), 20, 10
for wm in 32 20 10; do
  for b in 1 2 4 6 8; do
    CAPGPU_MSM_WIDE_MIN=$wm python bench.py --batch $b --one-context --steps 12 --warmup 3 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('wide_min $wm batch $b', round(d['value'],1))"
  done
done
