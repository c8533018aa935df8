#!/bin/bash
# socket power and clocks while the headline batch runs (rocm-smi samples every 0.5 s), against the idle reading
rocm-smi --showpower --showclocks --showmaxpower --showperflevel 2>&1 | grep -v "^=\|^$" | head -30
python bench.py --steps 40 --warmup 2 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_power.json 2>/dev/null &
BP=$!
sleep 25
for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>&1 | grep -i "power\|sclk\|fclk\|mclk" | tr '\n' ' '; echo; sleep 0.5; done
wait $BP
python -c "import json; d=json.load(open('gpurun_out/bench_power.json')); print('value', d['value'])"
