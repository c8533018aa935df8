#!/usr/bin/env python3
"""Device time of batches of NTTs (HIP events, median of 30): tools/gpu_ntt_time.py TAG - one JSON line.  For A/B runs of
library builds (CAPGPU_LIBRARY) on one box: tools/gpujob_r05_nttexp.sh."""
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
for log_n, count in ((16, 768), (17, 64), (15, 1280)):
    n = 1 << log_n
    d = cg.DevBuf.from_numpy(bu.random_canonical_scalars(7, count * n))
    for _ in range(5):
        cg.ntt_fr_dev(d, log_n, count=count, coset=True)
    ts = []
    cg.profile_reset()
    cg.profile_enable(True)
    for _ in range(30):
        cg.timer_begin()
        cg.ntt_fr_dev(d, log_n, count=count, coset=True)
        ts.append(cg.timer_end())
    st = cg.profile_stats()
    cg.profile_enable(False)
    ts.sort()
    out[f"2^{log_n}_x{count}"] = {"ms": ts[15], "col_ms": st.get("ntt_col_pass", (0, 1))[0] / 30, "row_ms": st.get("ntt_row_pass", (0, 1))[0] / 30,
                                  "GBps": 64.0 * n * count / ts[15] / 1e6}
    d.free()
print(json.dumps(out))
