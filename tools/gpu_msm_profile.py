"""Per-kernel times of single MSMs (2^17, 2^20 points) through the library's HIP-event profiler."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg  # noqa: E402
from cap_amd import bench_utils as bu  # noqa: E402

cg.init(0)
out = {}
import os
for log_n in [int(x) for x in os.environ.get('MSM_LOGS','17,20').split(',')]:
    n = 1 << log_n
    h = cg.srs_generate(0x1234567, n)
    rng = np.random.default_rng(log_n)
    sc = rng.integers(0, 1 << 62, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= (1 << 60) - 1
    d = cg.DevBuf.from_numpy(sc)
    for _ in range(3):
        cg.msm_g1_dev(h, d, n)
    cg.sync()
    t0 = time.perf_counter()
    for _ in range(10):
        cg.msm_g1_dev(h, d, n)
    cg.sync()
    wall = (time.perf_counter() - t0) / 10 * 1e3
    cg.profile_enable(True); cg.profile_reset()
    for _ in range(10):
        cg.msm_g1_dev(h, d, n)
    cg.sync()
    st = cg.profile_stats(); cg.profile_enable(False)
    out[log_n] = {"wall_ms": wall, "kernels_ms": {k: round(v[0] / 10, 4) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}}
    cg.srs_free(h)
print(json.dumps(out, indent=1))
