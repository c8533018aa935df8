#!/bin/bash
OUT=gpurun_out/r05_spin
mkdir -p $OUT
rm -f $OUT/small_ab.jsonl
run() { tag=$1; shift; env "$@" python tools/gpu_small_ab.py $tag >> $OUT/small_ab.jsonl 2>> $OUT/small_ab.err; }
run block CAPGPU_SPIN_SYNC_MAX=0
run spin CAPGPU_SPIN_SYNC_MAX=16
run block2 CAPGPU_SPIN_SYNC_MAX=0
run spin2 CAPGPU_SPIN_SYNC_MAX=16
run spin_ov16 CAPGPU_SPIN_SYNC_MAX=16 CAPGPU_R1_OVERLAP_MAX=16
python - <<PY
import json
for ln in open("$OUT/small_ab.jsonl"):
    d = json.loads(ln)
    print(d["tag"], "msm15 %.3f msm17 %.3f | prove 1/2/4/8/16: %s | %s" % (d["msm_2^15_ms"]["median"], d["msm_2^17_ms"]["median"],
          " ".join("%.2f" % d["prove_batch%d_ms" % p]["median"] for p in (1, 2, 4, 8, 16)), d["first_proof_sha256_16"]))
PY
tail -3 $OUT/small_ab.err
