#!/bin/bash
# round 6, first call: Shoup go/no-go microbenchmark, phase traces of the host-witness and coalesced paths (A/B of the
# H2D part order), the extended world-2 bench test
O=gpurun_out/r06_a
mkdir -p $O
timeout 300 tools/ubench_shoup29.bin > $O/ubench_shoup29.txt 2>&1; cat $O/ubench_shoup29.txt
for cfg in "order1 CAPGPU_H2D_PART_ORDER=1" "order0 CAPGPU_H2D_PART_ORDER=0" "order1_c8 CAPGPU_H2D_PART_ORDER=1 CAPGPU_PROVE_CHUNKS=8" "order1_c2 CAPGPU_H2D_PART_ORDER=1 CAPGPU_PROVE_CHUNKS=2"; do
  set -- $cfg; tag=$1; shift
  env "$@" timeout 600 python tools/gpu_phase_trace.py host $tag >> $O/phase.jsonl 2>> $O/phase.err
done
timeout 600 python tools/gpu_phase_trace.py resident base >> $O/phase.jsonl 2>> $O/phase.err
timeout 600 python tools/gpu_phase_trace.py coalesce base >> $O/phase.jsonl 2>> $O/phase.err
timeout 600 python tools/gpu_phase_trace.py coalesce w200 --window-us 200 >> $O/phase.jsonl 2>> $O/phase.err
timeout 600 python tools/gpu_phase_trace.py coalesce t128 --threads 128 --calls 4 >> $O/phase.jsonl 2>> $O/phase.err
python - <<PY
import json
for ln in open("$O/phase.jsonl"):
    d = json.loads(ln)
    print(d["mode"], d["tag"], round(d["proofs_per_s"], 1), {k: v for k, v in d.items() if k in ("batch_size", "batches_in_flight_share_of_wall", "per_batch_ms_median", "leader_ms_median", "leader_ms_mean", "caller_latency_ms", "device_batches")})
    for b in d.get("batches", [])[:2]:
        print("   ", json.dumps(b))
PY
tail -5 $O/phase.err
timeout 1200 python -m pytest tests/test_bench_launch.py tests/test_gpu_plonk.py -x -q -m gpu 2>&1 | tail -8
