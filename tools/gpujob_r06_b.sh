#!/bin/bash
# round 6, second call: ubench v2 (row-wise Shoup, Fr digit), host-witness A/B (first chunk, uneven deal), coalescer in-flight limit
O=gpurun_out/r06_b
mkdir -p $O
timeout 300 tools/ubench_shoup29.bin > $O/ubench_shoup29.txt 2>&1; cat $O/ubench_shoup29.txt
timeout 600 python tools/gpu_phase_trace.py resident base >> $O/phase.jsonl 2>> $O/phase.err
for cfg in "f16_s9" "f0_s8 CAPGPU_PROVE_FIRST_CHUNK=0 CAPGPU_DEAL_FIRST_SIXTEENTHS=8" "f16_s8 CAPGPU_DEAL_FIRST_SIXTEENTHS=8" "f0_s9 CAPGPU_PROVE_FIRST_CHUNK=0" "f16_s10 CAPGPU_DEAL_FIRST_SIXTEENTHS=10" "f8_s9 CAPGPU_PROVE_FIRST_CHUNK=8" "f16_s9_again"; do
  set -- $cfg; tag=$1; shift
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py host $tag --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
done
timeout 600 python tools/gpu_phase_trace.py resident base2 >> $O/phase.jsonl 2>> $O/phase.err
for cfg in "if2" "if3 CAPGPU_COALESCE_INFLIGHT=3" "if4 CAPGPU_COALESCE_INFLIGHT=4" "if2_again"; do
  set -- $cfg; tag=$1; shift
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce $tag >> $O/phase.jsonl 2>> $O/phase.err
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce ${tag}_w200 --window-us 200 >> $O/phase.jsonl 2>> $O/phase.err
done
env CAPGPU_X=1 timeout 600 python tools/gpu_phase_trace.py coalesce if2_t128 --threads 128 --calls 4 >> $O/phase.jsonl 2>> $O/phase.err
env CAPGPU_COALESCE_INFLIGHT=3 timeout 600 python tools/gpu_phase_trace.py coalesce if3_t128 --threads 128 --calls 4 >> $O/phase.jsonl 2>> $O/phase.err
python - <<PY
import json
for ln in open("$O/phase.jsonl"):
    d = json.loads(ln)
    print(d["mode"], d["tag"], round(d["proofs_per_s"], 1), {k: v for k, v in d.items() if k in ("batch_size", "batches_in_flight_share_of_wall", "per_batch_ms_median", "leader_ms_mean", "caller_latency_ms", "device_batches")})
    for b in d.get("batches", [])[1:2]:
        print("   ", json.dumps(b))
PY
tail -5 $O/phase.err
timeout 1200 python -m pytest tests/test_gpu_plonk.py tests/test_gpu_input_forms.py -x -q -m gpu 2>&1 | tail -4
