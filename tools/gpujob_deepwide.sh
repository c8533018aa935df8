#!/bin/bash
# Deep plan with 22-bit windows, the shifted top window and the heavy-bucket combine for tables of >= 2^24 points
# (default) against c = 20 (CAPGPU_MSM_DEEP_WIDE=0): parity, then 2^24-point MSMs  -> gpurun_out/deepwide_ab.jsonl
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_primitives.py -x -q -m gpu -k "deep or skew or 2p24 or config5 or special or sharded or known" > $OUT/deepwide_pytest.txt 2>&1; grep -E "passed|failed|error" $OUT/deepwide_pytest.txt | tail -3
: > $OUT/deepwide_ab.jsonl
for w in 0 1 0 1; do
  echo "{\"config\": {\"deep_wide\": $w}}" >> $OUT/deepwide_ab.jsonl
  CAPGPU_MSM_DEEP_WIDE=$w MINLOG=24 timeout 600 python tools/gpu_msm_deep_ab.py 24 >> $OUT/deepwide_ab.jsonl 2>> $OUT/deepwide.err
done
python - <<PY
import json
for ln in open("$OUT/deepwide_ab.jsonl"):
    d = json.loads(ln)
    if "ms_deep" in d:
        print(d["n"], d["plan_deep"]["c"], d["ms_deep"], d["same_result"], d["deep_kernels_ms"])
    else:
        print(d)
PY
