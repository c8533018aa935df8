#!/bin/bash
# proofs/s against the batch size on the round-5 binary: one context (--one-context, profiler on) and the default two-context headline
O=gpurun_out/r05_batchsweep; mkdir -p $O
for b in 1 2 4 8 16 32 64 128 256 512; do
  st=$(( b < 16 ? 30 : (b < 128 ? 10 : 4) ))
  python bench.py --batch $b --steps $st --warmup 2 --one-context --no-cpu-baseline --no-reference-schedule --no-msm --no-extras --no-mixed --no-realistic > $O/one_$b.json 2> $O/one_$b.err
  python bench.py --batch $b --steps $st --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras --no-mixed --no-realistic > $O/two_$b.json 2> $O/two_$b.err
done
python - <<PY
import json
for b in (1, 2, 4, 8, 16, 32, 64, 128, 256, 512):
    try:
        a = json.load(open("$O/one_%d.json" % b)); t = json.load(open("$O/two_%d.json" % b))
        print("batch %4d: one context %7.1f proofs/s (%.2f ms per batch), two contexts %7.1f" % (b, a["value"], a["ms_per_step"], t["value"]))
    except Exception as e:
        print("batch", b, "failed", e)
PY
