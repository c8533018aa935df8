#!/usr/bin/env python3
"""Same-box A/B: host-resident batches of 4 .. 64 proofs (capgpu_plonk_prove_batch, unbound caller: the library deals the
batch over its two contexts) for CAPGPU_DEAL_MIN = 8 (a part holds at least 8 proofs: batches below 16 stay whole) and 4
/ 2.  One JSON line per setting, each in a process of its own.   python tools/gpu_deal_ab.py"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child():
    import numpy as np
    from cap_amd import bench_utils as bu
    from cap_amd import lib as cg
    cg.init(0)
    log_n, ni = 15, 27
    n = 1 << log_n
    tau = bu.SplitMix64(0xCA9).field()
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
    pk, _ = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
    wit = [sc.witness(3 + i) for i in range(4)]
    out = {"deal_min": os.environ.get("CAPGPU_DEAL_MIN", "8")}
    for P in (4, 8, 16, 32, 64):
        wires = np.stack([sc.wires_mont(wit[i % 4][0]) for i in range(P)])
        pubs = np.stack([bu.to_mont_array(wit[i % 4][1]) for i in range(P)])
        blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
        for _ in range(4):
            cg.plonk_prove_batch(pk, wires, pubs, blind, b"d", P)
        reps = max(4, 64 // P)
        t0 = time.perf_counter()
        for _ in range(reps):
            cg.plonk_prove_batch(pk, wires, pubs, blind, b"d", P)
        out[f"batch{P}_proofs_per_s"] = round(P * reps / (time.perf_counter() - t0), 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child()
    else:
        for dm in ("8", "4", "2"):
            e = dict(os.environ)
            e["CAPGPU_DEAL_MIN"] = dm
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(line[-1] if line else json.dumps({"deal_min": dm, "error": r.stderr[-400:]}), flush=True)
