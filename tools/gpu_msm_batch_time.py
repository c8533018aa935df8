#!/usr/bin/env python3
"""Device time of `count` MSMs of 2^log_n points in one call (HIP events, median of 30), for the plan knobs read once per
process (CAPGPU_MSM_WIDE_MIN ...): python tools/gpu_msm_batch_time.py LOG_N COUNT[,COUNT...] - one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 17
counts = [int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "1,5").split(",")]
cg.init(0)
cg.set_device(0)
n = 1 << log_n
srs = cg.srs_generate_affine_seq(0x1234567890ABCDEF % bu.R, 0xFEDCBA0987654321 % bu.R, n)
out = {"log_n": log_n, "knobs": {k: v for k, v in os.environ.items() if k.startswith("CAPGPU_MSM")}}
for count in counts:
    sc = bu.random_canonical_scalars(500 + count, count * n).reshape(count, n, 4)
    d_sc, d_out = cg.DevBuf.from_numpy(sc), cg.DevBuf(96 * count)
    for _ in range(8):
        cg.msm_g1_dev(srs, d_sc, n, count=count, d_out=d_out)
    cg.sync()
    ts = []
    for _ in range(30):
        cg.timer_begin()
        cg.msm_g1_dev(srs, d_sc, n, count=count, d_out=d_out)
        ts.append(cg.timer_end())
    ts.sort()
    out[f"x{count}"] = {"ms": round(ts[15], 4), "ms_per_msm": round(ts[15] / count, 4), "plan": cg.msm_plan(srs, n, count)}
    d_sc.free()
    d_out.free()
print(json.dumps(out))
