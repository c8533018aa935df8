#!/bin/bash
tag=$1
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$tag.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.txt
tail -3 gpurun_out/pytest_gpu_$tag.txt
python bench.py > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "ref_sched", d.get("reference_schedule",{}).get("proofs_per_s"), "cpu", d.get("cpu_baseline",{}).get("value"), "parity", d.get("cpu_baseline",{}).get("gpu_proof_bit_exact_vs_cpu"), "speedup", d.get("speedup_vs_cpu_1core"))
print(d["roofline"]); print(d["top_kernels_ms"]); print(d.get("msm"))
PY
tail -2 gpurun_out/bench_$tag.err
python bench.py --workload mixed64 --steps 4 --warmup 1 --no-msm > gpurun_out/bench_mixed_$tag.json 2> gpurun_out/bench_mixed_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_mixed_$tag.json')); print('mixed64 value', round(d['value'],1), 'ms/step', round(d['ms_per_step'],2))"
tail -2 gpurun_out/bench_mixed_$tag.err
