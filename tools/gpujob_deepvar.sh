mkdir -p gpurun_out; rm -f gpurun_out/deep_var.log
fmt='import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r["n"], r["plan_deep"]["c"], r["plan_deep"].get("sort"), r["ms_deep"], r["ms_parts"], r["same_result"], r["deep_kernels_ms"])'
for c in 17 20; do echo "DEEP_C=$c DEEP_MIN=300000" >> gpurun_out/deep_var.log; MINLOG=${MINLOG:-19} CAPGPU_MSM_DEEP_MIN=300000 CAPGPU_MSM_DEEP_C=$c timeout 300 python tools/gpu_msm_deep_ab.py ${MAXLOG:-21} 2>&1 | python -c "$fmt" >> gpurun_out/deep_var.log; done; cat gpurun_out/deep_var.log
