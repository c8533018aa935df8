#!/usr/bin/env python3
"""Device time of batches of NTTs (HIP events, median of 30): tools/gpu_ntt_time.py TAG - one JSON line.  For A/B runs of
library builds (CAPGPU_LIBRARY) on one box: tools/gpujob.sh ntt."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

cg.init(0)
cg.set_device(0)
out = {"tag": sys.argv[1] if len(sys.argv) > 1 else "run", "lib": os.environ.get("CAPGPU_LIBRARY", "default")}
import numpy as np  # noqa: E402

for log_n, count, padded in ((16, 768, False), (16, 768, True), (17, 64, False), (15, 1280, False)):
    n = 1 << log_n
    a = bu.random_canonical_scalars(7, count * n)
    if padded:      # the prover's forward transforms: a polynomial of n / 2 + 2 coefficients, the rest zeros
        a = a.reshape(count, n, 4)
        a[:, n // 2 + 2:] = 0
        a = np.ascontiguousarray(a.reshape(-1, 4))
    d = cg.DevBuf.from_numpy(a)
    for _ in range(5):
        if padded:
            d.upload(a)
        cg.ntt_fr_dev(d, log_n, count=count, coset=True)
    ts = []
    cg.profile_reset()
    cg.profile_enable(True)
    for _ in range(12 if padded else 30):
        if padded:
            d.upload(a)      # (in place: the zeros have to come back)
        cg.timer_begin()
        cg.ntt_fr_dev(d, log_n, count=count, coset=True)
        ts.append(cg.timer_end())
    st = cg.profile_stats()
    cg.profile_enable(False)
    ts.sort()
    reps = len(ts)
    out[f"2^{log_n}_x{count}" + ("_half_zero" if padded else "")] = {"ms": ts[reps // 2], "col_ms": st.get("ntt_col_pass", (0, 1))[0] / reps, "row_ms": st.get("ntt_row_pass", (0, 1))[0] / reps,
                                  "GBps": 64.0 * n * count / ts[reps // 2] / 1e6}
    d.free()
print(json.dumps(out))
