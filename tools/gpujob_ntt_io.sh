#!/bin/bash
# Round 4: global loads of the NTT passes issued in batches (same box A/B; variants built by
#   make -C cap_amd/csrc OUT=../../tools/libcapgpu_X.so OBJDIR=_obj_X EXTRA=-D...)
for v in io1 io2 default io4w4 io1 default; do
  if [ $v = default ]; then lib=cap_amd/libcapgpu.so; else lib=tools/libcapgpu_$v.so; fi
  CAPGPU_LIBRARY=$PWD/$lib python bench.py --one-context --steps 4 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['top_kernels_ms']; s=d['top_kernels_steps']
print('$v', round(d['value'],1), {a:round(b/s,2) for a,b in k.items() if 'ntt' in a or 'quot' in a or 'accum' in a})"
done
