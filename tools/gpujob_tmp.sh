#!/bin/bash
O=gpurun_out/r06_g; mkdir -p $O
bash tools/gpujob.sh ntt madwide tools/libcapgpu_nomadwide.so | python3 -c "
import sys, json
for ln in sys.stdin:
    if ln.startswith('{'):
        d = json.loads(ln); print(d['tag'], {k: (round(v['ms'], 3), round(v['col_ms'], 3), round(v['row_ms'], 3)) for k, v in d.items() if isinstance(v, dict)})
"
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_plonk.py -x -q -m gpu --timeout=240 2>&1 | tail -3
for lib in cap_amd/libcapgpu.so tools/libcapgpu_nomadwide.so cap_amd/libcapgpu.so tools/libcapgpu_nomadwide.so; do
  CAPGPU_LIBRARY=$PWD/$lib timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm --no-mixed > $O/bench_$(basename $lib .so).json 2>/dev/null
  python - <<PY
import json
d = json.load(open("$O/bench_$(basename $lib .so).json"))
print("$lib", round(d["value"], 1), {k: round(v / d["top_kernels_steps"], 2) for k, v in d["top_kernels_ms"].items()})
PY
done
