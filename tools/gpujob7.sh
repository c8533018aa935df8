#!/bin/bash
tag=$1
mkdir -p gpurun_out
nproc; lscpu | grep -E "Model name|^CPU\(s\)"
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$tag.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.txt
tail -3 gpurun_out/pytest_gpu_$tag.txt
for P in 64 128; do
python bench.py --batch $P --no-msm > gpurun_out/bench_${tag}_P$P.json 2> gpurun_out/bench_${tag}_P$P.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_${tag}_P$P.json"))
print("P=$P value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "ref", d.get("reference_schedule",{}).get("proofs_per_s"), "parity", d.get("cpu_baseline",{}).get("gpu_proof_bit_exact_vs_cpu"))
print(d["top_kernels_ms"])
PY
done
CAPGPU_HOST_THREADS=1 python bench.py --batch 64 --no-msm --no-cpu-baseline --no-reference-schedule > gpurun_out/bench_${tag}_1thr.json 2>/dev/null
python -c "
import json; d=json.load(open('gpurun_out/bench_${tag}_1thr.json')); print('1 host thread P=64 value', round(d['value'],1))"
