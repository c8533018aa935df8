#!/bin/bash
# Same-box comparison of several library builds: tools/gpujob_abn.sh TAG LIB1 LIB2 ...  (the in-tree library runs first and last)
TAG=$1; shift
OUT=gpurun_out/abn_$TAG
mkdir -p $OUT
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras"
python bench.py $ARGS > $OUT/base.json 2> $OUT/base.err
i=0
for L in "$@"; do
  i=$((i+1))
  CAPGPU_LIBRARY=$PWD/$L python bench.py $ARGS > $OUT/v$i.json 2> $OUT/v$i.err
done
python bench.py $ARGS > $OUT/base2.json 2> $OUT/base2.err
python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), round(d["value"], 1), "proofs/s", {k: round(v / d["steps"], 2) for k, v in d["top_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY
