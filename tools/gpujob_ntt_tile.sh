for t in 10 9 8; do
  CAPGPU_NTT_TILE_LOG=$t python bench.py --one-context --steps 3 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('tile_log $t', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'ntt' in a or 'quot' in a})"
done
