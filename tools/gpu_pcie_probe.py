#!/usr/bin/env python3
"""Why does bench.py's pcie_inclusive leg read ~0.94 of the resident rate when tools/gpu_phase_trace.py host reads 0.97 on
the same box?  Replays the bench's sequence piece by piece in ONE process: resident two-context rate, host batch calls,
alternating in one process, then again after the single-proof latency loop the bench runs before that leg."""
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

cg.init(0)
P, log_n, ni = 256, 15, 27
n = 1 << log_n
tau = bu.SplitMix64(0xCA9).field()
srs = cg.srs_generate(tau, n + 3)
sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
pk, _ = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
wires, pubs = sc.witnesses_mont([3 + i for i in range(P)])
blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
msg = bytes(range(32))
d = cg.DevBuf.from_numpy(wires)
per = 5 * n * 32


def resident(steps=4):
    bar = threading.Barrier(3)

    def run(i):
        cg.set_device(i)
        lo, hi = (0, P // 2) if i == 0 else (P // 2, P)
        buf = d.view(lo * per, (hi - lo) * per)
        cg.plonk_prove_batch_dev(pk, buf, pubs[lo:hi], blind[lo:hi], msg, hi - lo)
        bar.wait()
        for _ in range(steps):
            cg.plonk_prove_batch_dev(pk, buf, pubs[lo:hi], blind[lo:hi], msg, hi - lo)
        bar.wait()

    th = [threading.Thread(target=run, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return P * steps / dt


def host(reps=6):
    cg.set_device(-1)
    cg.plonk_prove_batch(pk, wires, pubs, blind, msg, P)
    t0 = time.perf_counter()
    for _ in range(reps):
        cg.plonk_prove_batch(pk, wires, pubs, blind, msg, P)
    r = P * reps / (time.perf_counter() - t0)
    cg.set_device(0)
    return r


out = {}
cg.set_device(0)
out["resident_0"] = resident()
out["host_0"] = host()
for k in range(1, 4):
    out[f"resident_{k}"] = resident(6)
    out[f"host_{k}"] = host()
d1 = cg.DevBuf.from_numpy(wires[:1])
for _ in range(60):
    cg.plonk_prove_batch_dev(pk, d1, pubs[:1], blind[:1], msg, 1)
out["resident_after_latency_leg"] = resident(6)
out["host_after_latency_leg"] = host()
# what else the bench does before that leg: the whole batch on ONE context with the HIP-event profiler on (context 0's scratch
# grows to 256 proofs' worth), the issue-rate microbenchmarks
cg.set_device(0)
cg.profile_reset()
cg.profile_enable(True)
for _ in range(3):
    cg.plonk_prove_batch_dev(pk, d, pubs, blind, msg, P)
cg.profile_stats()
cg.profile_enable(False)
out["resident_after_profiled_pass"] = resident(6)
out["host_after_profiled_pass"] = host()
cg.ubench_mad_rate()
cg.ubench_issue_rates()
out["resident_after_ubench"] = resident(6)
out["host_after_ubench"] = host()
out["resident_3steps"] = resident(3)
for k in ("0", "1", "2", "3", "after_latency_leg", "after_profiled_pass", "after_ubench"):
    out["ratio_" + k] = out["host_" + k] / out["resident_" + k]
print(json.dumps({k: round(v, 4 if k.startswith("ratio") else 1) for k, v in out.items()}))
