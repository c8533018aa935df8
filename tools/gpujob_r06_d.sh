#!/bin/bash
# round 6: host-witness path with two bound callers (chunking, pinned memory); coalescer with the arrival-aware window
O=gpurun_out/r06_d
mkdir -p $O; rm -f $O/phase.jsonl
timeout 600 python tools/gpu_phase_trace.py resident base --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
for cfg in "c4" "c1 CAPGPU_PROVE_CHUNKS=1" "c2 CAPGPU_PROVE_CHUNKS=2" "c4_again"; do
  set -- $cfg; tag=$1; shift
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py host2 $tag --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py host2 ${tag}_pin --pin --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
done
timeout 600 python tools/gpu_phase_trace.py host sync --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
timeout 600 python tools/gpu_phase_trace.py host sync_pin --pin --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
timeout 600 python tools/gpu_phase_trace.py resident base2 --reps 6 >> $O/phase.jsonl 2>> $O/phase.err
for cfg in "pre1" "pre0 CAPGPU_COALESCE_PRESTAGE=0" "pre1_if3 CAPGPU_COALESCE_INFLIGHT=3" "pre1_again"; do
  set -- $cfg; tag=$1; shift
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce $tag >> $O/phase.jsonl 2>> $O/phase.err
  env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py coalesce ${tag}_c16 --calls 16 >> $O/phase.jsonl 2>> $O/phase.err
done
python - <<PY
import json
for ln in open("$O/phase.jsonl"):
    d = json.loads(ln)
    print(d["mode"], d["tag"], round(d["proofs_per_s"], 1), {k: v for k, v in d.items() if k in ("batch_size", "batches_in_flight_share_of_wall", "per_batch_ms_median", "leader_ms_mean", "caller_latency_ms", "device_batches", "host_register_rc")})
PY
tail -5 $O/phase.err
