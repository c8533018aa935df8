#!/bin/bash
# usage: bash tools/gpujob.sh <tag> [bench args...]   (runs on the GPU box via gpurun)
tag=$1; shift
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_$tag.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_$tag.txt
tail -4 gpurun_out/pytest_gpu_$tag.txt
python bench.py --steps 4 --warmup 1 "$@" > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2), "ref_sched", d.get("reference_schedule",{}).get("proofs_per_s"), "cpu", d.get("cpu_baseline",{}).get("value"), "parity", d.get("cpu_baseline",{}).get("gpu_proof_bit_exact_vs_cpu"))
print("roofline", d["roofline"]["kernel"], d["roofline"]["achieved"], d["roofline"]["frac"])
print(d["top_kernels_ms"])
PY
tail -3 gpurun_out/bench_$tag.err
