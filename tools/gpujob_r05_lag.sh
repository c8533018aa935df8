#!/bin/bash
# round 5: Lagrange-form wire commitments - parity tests, then the bench with the realistic_witness leg
OUT=gpurun_out/r05_lag
mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_lagrange.py -x -q > $OUT/pytest_lagrange.txt 2>&1
tail -15 $OUT/pytest_lagrange.txt
timeout 900 python bench.py --steps 6 --warmup 2 > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
try:
    d = json.load(open("$OUT/bench_default.json"))
    print(round(d["value"], 1), "proofs/s; one ctx", round(d.get("one_context_profiled_pass", {}).get("proofs_per_s", 0), 1),
          {k: round(v / d["top_kernels_steps"], 2) for k, v in d["top_kernels_ms"].items()})
    print(json.dumps(d["config"].get("legs"), indent=1))
    print(json.dumps(d.get("realistic_witness"), indent=1)[:3000])
    print(json.dumps(d.get("cpu_baseline"), indent=1))
except Exception as e:
    print("failed", e, open("$OUT/bench_default.err").read()[-2500:])
PY
