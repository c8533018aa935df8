#!/bin/bash
# Round-end evidence on one GPU box: GPU test suite, default bench, mixed workload, single-MSM profile, rocprofv3
# kernel stats + PMC traffic passes, instruction counters, world-2 (gloo, one GPU) runs.  Outputs under gpurun_out/final_$1.
tag=$1
O=gpurun_out/final_$tag
mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q --durations=10 --timeout=600 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -c 300 $O/bench.err
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm > $O/bench_20steps.json 2>/dev/null
timeout 600 python bench.py --workload mixed64 --steps 12 --warmup 3 --no-msm > $O/bench_mixed64.json 2>/dev/null
CAPGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --single-process --devices 0,0 --batch 128 --steps 4 --warmup 1 --msm-log-n 22 > $O/bench_single_process.json 2>/dev/null
MINLOG=21 timeout 300 python tools/gpu_msm_deep_ab.py 24 > $O/msm_deep_ab.jsonl 2>/dev/null
timeout 300 python tools/gpu_two_ctx.py 15 256 2>/dev/null | tail -1 > $O/two_ctx.json
timeout 600 python tools/gpu_latency_ab.py --quick > $O/latency_ab.jsonl 2>/dev/null
MSM_LOGS=15,17,20,22,24 timeout 600 python tools/gpu_msm_profile.py > $O/msm_single_profile.json 2>/dev/null
bash tools/gpuprof.sh $tag > $O/gpuprof.log 2>&1
python tools/make_traffic.py gpurun_out/prof_$tag 256 $O/traffic.json > $O/traffic.log 2>&1
bash tools/gpuprof_insts.sh $tag > $O/insts.txt 2>&1
bash tools/gpuprof_clock.sh $tag > $O/clock.log 2>&1; cp gpurun_out/clock_$tag/clock.json $O/clock.json
timeout 120 tools/ubench_mix.bin > $O/ubench_mix.txt 2>&1
bash tools/gpujob_w2.sh > $O/w2.log 2>&1; cp gpurun_out/bench_w2.json $O/bench_w2.json; cp gpurun_out/bench_w2_mixed.json $O/bench_w2_mixed.json
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v / d["steps"], 2) for k, v in d["top_kernels_ms"].items()})
for k in ("alu_roofline", "reference_schedule", "latency_ms_batch1", "pcie_inclusive", "n2p16", "cpu_baseline", "cpu_baseline_64_threads", "two_contexts_per_device", "mixed64", "coalesced_single_calls"):
    print(k, d.get(k))
print([ (l.get("log_n"), round(l.get("ms", 0), 3), l.get("identity_check")) for l in d.get("msm", [])])
print("20 steps:", round(json.load(open("$O/bench_20steps.json"))["value"], 1), "mixed64:", round(json.load(open("$O/bench_mixed64.json"))["value"], 1))
PY
