#!/usr/bin/env python3
"""One proof at a time, N times (for kernel traces of a single proof: rocprofv3 --kernel-trace -- python3 tools/gpu_prove1.py)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
P = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cg.init(0)
cg.set_device(0)
log_n, nin = 15, 27
n = 1 << log_n
srs = cg.srs_generate(bu.SplitMix64(0xCA9).field(), n + 3)
sc = bu.synthetic_circuit(log_n, nin, seed=2 + log_n + nin)
pk, _ = cg.plonk_preprocess(srs, n, nin, sc.selectors_mont(), sc.sigma_mont())
wm, pm = sc.witnesses_mont([3 + i for i in range(P)])
bl = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
d_w = cg.DevBuf.from_numpy(wm)
ts = []
for i in range(reps):
    t0 = time.perf_counter()
    cg.plonk_prove_batch_dev(pk, d_w, pm, bl, b"x" * 32, P)
    ts.append((time.perf_counter() - t0) * 1e3)
ts = sorted(ts[5:])
print("median ms", ts[len(ts) // 2], file=sys.stderr)
