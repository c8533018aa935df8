#!/bin/bash
# round 5, first GPU contact: (1) A/B of the headline batch with 4 cycled witnesses (rounds 1-4) against 256 distinct ones,
# same box, same binary; (2) the default bench line with the new legs (realistic_witness, hipEvent-timed msm / ntt legs).
OUT=gpurun_out/r05_first
mkdir -p $OUT
AB="--steps 6 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-extras --no-msm --no-mixed"
python bench.py $AB --n-wit 4 > $OUT/nwit4_a.json 2> $OUT/nwit4_a.err
python bench.py $AB > $OUT/nwit256_a.json 2> $OUT/nwit256_a.err
python bench.py $AB --n-wit 4 > $OUT/nwit4_b.json 2> $OUT/nwit4_b.err
python bench.py $AB > $OUT/nwit256_b.json 2> $OUT/nwit256_b.err
python bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err
python - <<PY
import json
for n in ("nwit4_a", "nwit256_a", "nwit4_b", "nwit256_b", "bench_default"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, round(d["value"], 1), "proofs/s; one ctx", round(d.get("one_context_profiled_pass", {}).get("proofs_per_s", 0), 1),
              {k: round(v / d["top_kernels_steps"], 2) for k, v in d["top_kernels_ms"].items()})
        if n == "bench_default":
            print(json.dumps(d["config"].get("legs"), indent=1))
            print(json.dumps(d.get("realistic_witness"), indent=1)[:3000])
            print(json.dumps(d.get("cpu_baseline"), indent=1))
    except Exception as e:
        print(n, "failed", e, open("$OUT/%s.err" % n).read()[-1500:])
PY
