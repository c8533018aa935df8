#!/bin/bash
# c = 22 (shifted top window, heavy-bucket combine) against c = 20 at 2^22 and 2^23 points -> gpurun_out/deepwide2_ab.jsonl
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
: > $OUT/deepwide2_ab.jsonl
for c in 20 22 20 22; do
  echo "{\"config\": {\"deep_c\": $c}}" >> $OUT/deepwide2_ab.jsonl
  CAPGPU_MSM_DEEP_C=$c MINLOG=22 timeout 600 python tools/gpu_msm_deep_ab.py 23 >> $OUT/deepwide2_ab.jsonl 2>> $OUT/deepwide2.err
done
python - <<PY
import json
for ln in open("$OUT/deepwide2_ab.jsonl"):
    d = json.loads(ln)
    if "ms_deep" in d:
        print(d["n"], d["plan_deep"]["c"], d["ms_deep"], d["same_result"], d["deep_kernels_ms"])
    else:
        print(d)
PY
