#!/bin/bash
# Smaller NTT tiles for launches that do not fill the chip (CAPGPU_NTT_TILE_ADAPT): parity, then single 2^15 .. 2^18
# transforms and the single-proof latency with and without  -> gpurun_out/nttadapt_ab.jsonl
OUT=gpurun_out
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_gpu_primitives.py tests/test_gpu_plonk.py -x -q -m gpu -k "ntt or plonk or note or proof" 2>&1 | tail -3
: > $OUT/nttadapt_ab.jsonl
for a in 0 1 0 1; do
  CAPGPU_NTT_TILE_ADAPT=$a python - >> $OUT/nttadapt_ab.jsonl <<PY
import json, sys
sys.path.insert(0, ".")
import bench
from cap_amd import lib as cg, bench_utils as bu
cg.init(0); cg.set_device(0)
out = {"adapt": $a}
for lg in (15, 16, 17, 18):
    legs = bench.ntt_leg(cg, bu, lg, iters=30)
    out[f"ntt_2p{lg}_us"] = round(legs[0]["ms_per_call"] * 1e3, 1)
    out[f"ntt_2p{lg}_x64_ms"] = round(legs[1]["ms_per_call"], 3)
    assert legs[0]["round_trip_identity"] and legs[1]["round_trip_identity"]
print(json.dumps(out))
PY
  CAPGPU_NTT_TILE_ADAPT=$a CAPGPU_AB_BATCHES=1,2,4 CAPGPU_AB_NAME=ntt_adapt_$a timeout 600 python tools/gpu_latency_ab.py --child 2>&1 | grep '^{' >> $OUT/nttadapt_ab.jsonl
done
cat $OUT/nttadapt_ab.jsonl
