#!/bin/bash
tag=$1
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$tag.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.txt
tail -3 gpurun_out/pytest_gpu_$tag.txt
python bench.py --steps 4 --warmup 1 > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "ref_sched", d.get("reference_schedule",{}).get("proofs_per_s"), "parity", d.get("cpu_baseline",{}).get("gpu_proof_bit_exact_vs_cpu"))
print(d["top_kernels_ms"]); print(d.get("msm"))
PY
tail -3 gpurun_out/bench_$tag.err
timeout 600 python bench.py --steps 1 --warmup 0 --batch 2 --no-cpu-baseline --no-reference-schedule --msm-log-n 24 > gpurun_out/bench_msm24_$tag.json 2> gpurun_out/bench_msm24_$tag.err
python -c "
import json; d=json.load(open('gpurun_out/bench_msm24_$tag.json')); print(d.get('msm'))"
tail -3 gpurun_out/bench_msm24_$tag.err
