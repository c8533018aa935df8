#!/bin/bash
# Single-proof latency of several library builds on one box: tools/gpujob_lat.sh TAG [LIB ...] (in-tree library first)
TAG=$1; shift
OUT=gpurun_out/lat_$TAG; mkdir -p $OUT
run() {  # name, library ("" = in-tree)
  CAPGPU_LIBRARY=$2 python - "$1" <<'PY'
import os, sys, time, json
import numpy as np
if not os.environ.get("CAPGPU_LIBRARY"): os.environ.pop("CAPGPU_LIBRARY", None)
from cap_amd import lib as cg, bench_utils as bu
cg.init(0)
log_n, ni = 15, 27
n = 1 << log_n
srs = cg.srs_generate(bu.SplitMix64(0xCA9).field(), n + 3)
sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
pk, vk = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
w, pubs = sc.witness(3)
wires = sc.wires_mont(w)[None]; pubs = bu.to_mont_array(pubs)[None]; bl = bu.to_mont_array(bu.blinders(7000))[None]
d = cg.DevBuf.from_numpy(wires)
for _ in range(5): cg.plonk_prove_batch_dev(pk, d, pubs, bl, b"x" * 32, 1)
ts = []
for _ in range(15):
    t = time.perf_counter(); cg.plonk_prove_batch_dev(pk, d, pubs, bl, b"x" * 32, 1); ts.append((time.perf_counter() - t) * 1e3)
print(sys.argv[1], "median ms", round(float(np.median(ts)), 3), "min", round(min(ts), 3))
cg.profile_enable(True); cg.profile_reset()
for _ in range(5): cg.plonk_prove_batch_dev(pk, d, pubs, bl, b"x" * 32, 1)
st = cg.profile_stats()
print("   ", [(nm, round(ms / 5, 3)) for nm, (ms, cnt) in sorted(st.items(), key=lambda kv: -kv[1][0])[:9]])
PY
}
run base ""
for L in "$@"; do run $L $PWD/$L; done
run base2 ""
