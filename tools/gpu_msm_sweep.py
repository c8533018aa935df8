"""Cliff detector: single-MSM time over sizes (and a few batch counts) - ms and ns per point; a size that costs more
per point than its smaller neighbour by a wide margin is a plan or kernel falling off its path."""
import json
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg  # noqa: E402
from cap_amd import bench_utils as bu  # noqa: E402

cg.init(0)
rows = []
sizes = [1 << k for k in range(10, 25)] + [(1 << 15) + 2, (1 << 16) + 2, (1 << 18) + 1, 3 << 18, 5 << 19]
hmax = cg.srs_generate_affine_seq(12345, 67, max(sizes))
for n in sorted(sizes):
    h = hmax if n > (1 << 18) else cg.srs_generate_affine_seq(12345, 67, n)     # the table's window follows the SRS size
    sc = bu.random_canonical_scalars(n % 97, n)
    d = cg.DevBuf.from_numpy(sc)
    for _ in range(2):
        cg.msm_g1_dev(h, d, n)
    cg.sync()
    it = 10 if n <= (1 << 20) else 4
    t0 = time.perf_counter()
    for _ in range(it):
        cg.msm_g1_dev(h, d, n)
    cg.sync()
    ms = (time.perf_counter() - t0) / it * 1e3
    rows.append({"n": n, "log2": round(float(np.log2(n)), 2), "ms": round(ms, 3), "ns_per_point": round(ms * 1e6 / n, 2),
                 "plan": cg.msm_plan(h, n, 1)})
    d.free()
    if h != hmax:
        cg.srs_free(h)
for r in rows:
    print(json.dumps(r))
