#!/usr/bin/env python3
"""Small-launch latencies on one GPU, one JSON line: single MSMs of 2^15 / 2^17 points (device time, HIP events) and proof
batches of 1 .. 16 (host wall time per call).  For same-box A/B runs of the knobs that are read once per process
(CAPGPU_MSM_CHAINED, CAPGPU_R1_OVERLAP_MAX, CAPGPU_WIRE_COMMIT, CAPGPU_LIBRARY ...): tools/gpujob.sh env."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402


def med(xs):
    xs = sorted(xs)
    return {"median": xs[len(xs) // 2], "p10": xs[len(xs) // 10], "p90": xs[(9 * len(xs)) // 10]}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "run"
    cg.init(0)
    cg.set_device(0)
    tau = bu.SplitMix64(0xCA9).field()
    out = {"tag": tag, "env": {k: v for k, v in os.environ.items() if k.startswith("CAPGPU_")}}
    for ln in (15, 17):
        n = 1 << ln
        srs = cg.srs_generate_affine_seq(0x1234567890ABCDEF % bu.R, 0xFEDCBA0987654321 % bu.R, n)
        d_sc = cg.DevBuf.from_numpy(bu.random_canonical_scalars(5, n))
        d_out = cg.DevBuf(96)
        for _ in range(10):
            cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)
        ts = []
        for _ in range(60):
            cg.timer_begin()
            cg.msm_g1_dev(srs, d_sc, n, d_out=d_out)
            ts.append(cg.timer_end())
        out[f"msm_2^{ln}_ms"] = med(ts)
        first = d_out.to_numpy().copy()
        out[f"msm_2^{ln}_plan"] = cg.msm_plan(srs, n, 1)
        cg.srs_free(srs)
        del first
    log_n, nin = 15, 27
    n = 1 << log_n
    srs = cg.srs_generate(tau, n + 3)
    kind = os.environ.get("SMALL_AB_CIRCUIT", "uniform")
    sc = bu.cap_like_circuit("transfer_2x2", seed=7) if kind == "cap" else bu.synthetic_circuit(log_n, nin, seed=2 + log_n + nin)
    pk, _ = cg.plonk_preprocess(srs, n, nin, sc.selectors_mont(), sc.sigma_mont())
    P = 16
    wm, pm = sc.witnesses_mont([3 + i for i in range(P)])
    bl = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
    d_w = cg.DevBuf.from_numpy(wm)
    digest = None
    for p in (1, 2, 4, 8, 16):
        view = d_w.view(0, p * 5 * n * 32)
        for _ in range(6):
            pr = cg.plonk_prove_batch_dev(pk, view, pm[:p], bl[:p], b"x" * 32, p)
        ts = []
        for _ in range(40 if p <= 4 else 20):
            t0 = time.perf_counter()
            pr = cg.plonk_prove_batch_dev(pk, view, pm[:p], bl[:p], b"x" * 32, p)
            ts.append((time.perf_counter() - t0) * 1e3)
        out[f"prove_batch{p}_ms"] = med(ts)
        if p == 1:
            import hashlib
            digest = hashlib.sha256(bytes(pr[0])).hexdigest()[:16]
    out["first_proof_sha256_16"] = digest
    out["circuit"] = kind
    print(json.dumps(out))


if __name__ == "__main__":
    main()
