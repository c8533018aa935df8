#!/bin/bash
O=gpurun_out/r06_r; mkdir -p $O
for i in 1 2 3 4 5 6; do
  timeout 900 python -m pytest tests -m gpu -q -x --timeout=300 > $O/run$i.txt 2>&1
  echo "run$i rc=$? $(grep -E 'passed|failed' $O/run$i.txt | tail -1)"
  if grep -q "native stack\|Segmentation\|Fatal Python" $O/run$i.txt; then grep -B5 -A60 "native stack" $O/run$i.txt | head -120; cp cap_amd/libcapgpu.so $O/ 2>/dev/null; break; fi
done
