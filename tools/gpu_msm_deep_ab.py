"""A/B of the deep-window plan (one bucket set, windows of log2(n) - 2 bits, three-level sort) against the sub-MSM plan
on the c = 15 table for single MSMs of 2^19 .. 2^24 points: time per MSM under both plans and agreement of the two results
(Jacobian triples compared by cross-multiplication).  CAPGPU_MSM_DEEP is read per call, the third table is built with the
SRS.  Usage: python tools/gpu_msm_deep_ab.py [max_log_n]"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47


def ints(w):
    w = np.asarray(w, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192 for r in w]


def same_point(a, b):
    x1, y1, z1 = ints(a)
    x2, y2, z2 = ints(b)
    if z1 == 0 or z2 == 0:
        return z1 == z2
    return (x1 * z2 * z2 - x2 * z1 * z1) % P == 0 and (y1 * z2 ** 3 - y2 * z1 ** 3) % P == 0


cg.init(0)
max_log = int(sys.argv[1]) if len(sys.argv) > 1 else 24
for log_n in range(int(os.environ.get("MINLOG", "19")), max_log + 1):
    for n in ([(1 << log_n)] if log_n > 19 else [(1 << 19) + 1]):
        h = cg.srs_generate_affine_seq(12345, 67, n)
        sc = bu.random_canonical_scalars(log_n, n)
        d = cg.DevBuf.from_numpy(sc)
        row = {"n": n}
        res = {}
        for mode in ("1", "0"):
            os.environ["CAPGPU_MSM_DEEP"] = mode
            row["plan_deep" if mode == "1" else "plan_parts"] = cg.msm_plan(h, n, 1)
            for _ in range(2):
                out = cg.msm_g1_dev(h, d, n)
            cg.sync()
            it = 8 if n <= (1 << 21) else 4
            t0 = time.perf_counter()
            for _ in range(it):
                out = cg.msm_g1_dev(h, d, n)
            cg.sync()
            row["ms_deep" if mode == "1" else "ms_parts"] = round((time.perf_counter() - t0) / it * 1e3, 3)
            res[mode] = out.to_numpy()
        os.environ["CAPGPU_MSM_DEEP"] = "1"
        row["same_result"] = same_point(res["1"], res["0"])
        # per-kernel time of the deep plan
        cg.profile_enable(True)
        cg.profile_reset()
        cg.msm_g1_dev(h, d, n)
        cg.sync()
        st = cg.profile_stats()
        cg.profile_enable(False)
        row["deep_kernels_ms"] = {k: round(v[0], 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])[:8]}
        print(json.dumps(row), flush=True)
        d.free()
        cg.srs_free(h)
