mkdir -p gpurun_out
CAPGPU_ACC_LDS_VARIANT=1 timeout 600 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_plonk.py -q -m gpu --timeout=300 -x -k "msm or batch_vs_c_oracle or golden or wide_window" 2>&1 | tail -3
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras"
for V in 0 1 0 1; do
  CAPGPU_ACC_LDS_VARIANT=$V python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('LDS_VARIANT=$V', round(d['value'], 1), {k: round(v / d['steps'], 2) for k, v in d['top_kernels_ms'].items()})"
done
