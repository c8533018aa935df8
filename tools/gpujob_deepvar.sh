mkdir -p gpurun_out; rm -f gpurun_out/deep_var.log
MINLOG=21 timeout 300 python tools/gpu_msm_deep_ab.py 24 2>&1 | python -c "
import sys, json
for l in sys.stdin:
    r = json.loads(l); print(r['n'], r['plan_deep']['c'], r['ms_deep'], r['ms_parts'], r['same_result'], r['deep_kernels_ms'])
" >> gpurun_out/deep_var.log; cat gpurun_out/deep_var.log
