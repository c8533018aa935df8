#!/bin/bash
# the whole GPU suite + the small-launch A/B (tools/gpu_small_ab.py) on the current tree
OUT=gpurun_out/r05_suite
mkdir -p $OUT
rm -f $OUT/small_ab.jsonl
run() { tag=$1; shift; env "$@" python tools/gpu_small_ab.py $tag >> $OUT/small_ab.jsonl 2>> $OUT/small_ab.err; }
run base_old CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=0
run both CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16
run both_nographs CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16 CAPGPU_GRAPH_MAX_BATCH=0
run base_old2 CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=0
run both2 CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16
run overlap64 CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=64
python - <<PY
import json
for ln in open("$OUT/small_ab.jsonl"):
    d = json.loads(ln)
    print(d["tag"], "msm15 %.3f msm17 %.3f | prove 1/2/4/8/16: %s | %s" % (d["msm_2^15_ms"]["median"], d["msm_2^17_ms"]["median"],
          " ".join("%.2f" % d["prove_batch%d_ms" % p]["median"] for p in (1, 2, 4, 8, 16)), d["first_proof_sha256_16"]))
PY
tail -3 $OUT/small_ab.err
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -8 $OUT/pytest_gpu.txt
