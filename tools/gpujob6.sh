#!/bin/bash
mkdir -p gpurun_out
for P in 96 128; do
python bench.py --steps 4 --warmup 1 --batch $P --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_P$P.json 2> gpurun_out/bench_P$P.err
python - <<PY
import json
d=json.load(open("gpurun_out/bench_P$P.json"))
print("P=$P value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2))
PY
tail -2 gpurun_out/bench_P$P.err
done
python bench.py --steps 4 --warmup 1 --log-n 16 --batch 32 --no-reference-schedule --no-msm > gpurun_out/bench_n16.json 2> gpurun_out/bench_n16.err
python -c "
import json; d=json.load(open('gpurun_out/bench_n16.json')); print('n=2^16 value', round(d['value'],1), 'cpu', d['cpu_baseline']['value'], 'parity', d['cpu_baseline']['gpu_proof_bit_exact_vs_cpu'])"
tail -2 gpurun_out/bench_n16.err
