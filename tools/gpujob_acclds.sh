mkdir -p gpurun_out
CAPGPU_ACC_PREFETCH=1 CAPGPU_ACC_LDS=65536 timeout 600 python -m pytest tests/test_gpu_primitives.py -q -m gpu --timeout=300 -x -k "msm" 2>&1 | tail -2
ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras"
for V in "0 0" "1 0" "1 65536" "0 65536" "1 40000" "0 0"; do set -- $V
  CAPGPU_ACC_PREFETCH=$1 CAPGPU_ACC_LDS=$2 python bench.py $ARGS 2>/dev/null | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print('PREFETCH=$1 LDS=$2', round(d['value'], 1), {k: round(v / d['steps'], 2) for k, v in list(d['top_kernels_ms'].items())[:3]})"
done
