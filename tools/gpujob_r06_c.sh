#!/bin/bash
# round 6: coalescer with native callers (pre-staging A/B, in-flight limit), default bench (N=1) and the world-2 (shared device, gloo) line with wall times
O=gpurun_out/r06_c
mkdir -p $O; rm -f $O/phase.jsonl
timeout 600 python tools/gpu_phase_trace.py resident base >> $O/phase.jsonl 2>> $O/phase.err
for cfg in "pre1" "pre0 CAPGPU_COALESCE_PRESTAGE=0" "pre1_if3 CAPGPU_COALESCE_INFLIGHT=3" "pre1_if4 CAPGPU_COALESCE_INFLIGHT=4" "pre0_if4 CAPGPU_COALESCE_INFLIGHT=4 CAPGPU_COALESCE_PRESTAGE=0" "pre1_again"; do
  set -- $cfg; tag=$1; shift
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce $tag >> $O/phase.jsonl 2>> $O/phase.err
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce ${tag}_w200 --window-us 200 >> $O/phase.jsonl 2>> $O/phase.err
done
env CAPGPU_X=1 timeout 600 python tools/gpu_phase_trace.py coalesce pre1_python --python-threads >> $O/phase.jsonl 2>> $O/phase.err
env CAPGPU_X=1 timeout 600 python tools/gpu_phase_trace.py coalesce pre1_t128 --threads 128 --calls 4 >> $O/phase.jsonl 2>> $O/phase.err
python - <<PY
import json
for ln in open("$O/phase.jsonl"):
    d = json.loads(ln)
    print(d["mode"], d["tag"], round(d["proofs_per_s"], 1), {k: v for k, v in d.items() if k in ("batch_size", "batches_in_flight_share_of_wall", "per_batch_ms_median", "leader_ms_mean", "caller_latency_ms", "device_batches")})
PY
tail -5 $O/phase.err
S0=$SECONDS; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench wall $((SECONDS-S0)) s"; tail -2 $O/bench.err
tail -c 2000 $O/bench.json; echo
S0=$SECONDS; CAPGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 900 python bench.py --gpus 2 > $O/bench_w2.json 2> $O/bench_w2.err; echo "bench --gpus 2 wall $((SECONDS-S0)) s"; tail -2 $O/bench_w2.err
tail -c 2000 $O/bench_w2.json; echo
