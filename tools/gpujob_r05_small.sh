#!/bin/bash
# round 5: small-launch latencies, same box: chained sort kernels on / off, round-1 side stream on / off
OUT=gpurun_out/r05_small
mkdir -p $OUT
LIB=${1:-cap_amd/libcapgpu.so}
run() { tag=$1; shift; env CAPGPU_LIBRARY=$PWD/$LIB "$@" python tools/gpu_small_ab.py $tag >> $OUT/small_ab.jsonl 2>> $OUT/small_ab.err; }
run base_old CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=0
run chained CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=0
run overlap CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=16
run both CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16
run both_coeffcommit CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16 CAPGPU_WIRE_COMMIT=coeffs
run base_old2 CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=0
run both_cap CAPGPU_MSM_CHAINED=1 CAPGPU_R1_OVERLAP_MAX=16 SMALL_AB_CIRCUIT=cap
python - <<PY
import json
for ln in open("$OUT/small_ab.jsonl"):
    d = json.loads(ln)
    print(d["tag"], "msm15 %.3f msm17 %.3f | prove 1/2/4/8/16: %s | %s" % (d["msm_2^15_ms"]["median"], d["msm_2^17_ms"]["median"],
          " ".join("%.2f" % d["prove_batch%d_ms" % p]["median"] for p in (1, 2, 4, 8, 16)), d["first_proof_sha256_16"]))
PY
tail -5 $OUT/small_ab.err
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_lagrange.py tests/test_gpu_graphs.py tests/test_gpu_fuzz.py -x -q 2>&1 | tail -5
