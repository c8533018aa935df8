#!/bin/bash
# A/B of two library builds on the same box: tools/gpujob_ab.sh TAG LIB_B [bench args]
#   A = cap_amd/libcapgpu.so (default), B = LIB_B (path relative to the repo root)
TAG=$1; LIBB=$2; shift 2
OUT=gpurun_out/ab_$TAG
mkdir -p $OUT
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm "$@" > $OUT/a.json 2> $OUT/a.err
CAPGPU_LIBRARY=$PWD/$LIBB python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm "$@" > $OUT/b.json 2> $OUT/b.err
python bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm "$@" > $OUT/a2.json 2> $OUT/a2.err
python - <<PY
import json
for n in ("a", "b", "a2"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, round(d["value"], 1), "proofs/s", {k: round(v / d["steps"], 2) for k, v in d["top_kernels_ms"].items()})
    except Exception as e:
        print(n, "failed", e, open("$OUT/%s.err" % n).read()[-500:])
PY
