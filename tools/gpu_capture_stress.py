#!/usr/bin/env python3
"""Stress of the small-batch path that captures hipGraphs with the side-stream fork / join (rounds 1-2 on two streams):
fresh keys of tiny domains, batches of 1 .. 3 proofs, evaluation- and coefficient-form calls alternating so that every
third call captures.  Hunting a one-in-ten crash seen once in tests/test_gpu_input_forms.py[6-0-3] (round 6).
    CAPGPU_SEGV_BACKTRACE=1 python tools/gpu_capture_stress.py [ITERATIONS]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cg.init(0)
tau = bu.SplitMix64(0xCA9).field()
ok = 0
for it in range(iters):
    log_n, nin, P = [(4, 1, 1), (6, 0, 3), (5, 2, 2), (7, 0, 3), (6, 3, 1)][it % 5]
    sc = bu.synthetic_circuit(log_n, nin, seed=40 + log_n + it)
    n = sc.n
    h = cg.srs_generate(tau, n + 3)
    pk, _ = cg.plonk_preprocess(h, n, nin, sc.selectors_mont(), sc.sigma_mont())
    ws, ps = sc.witnesses_mont([300 + p + it for p in range(P)])
    bls = np.stack([bu.to_mont_array(bu.blinders(1300 + p)) for p in range(P)])
    d = cg.DevBuf.from_numpy(ws)
    cg.ntt_fr_dev(d, log_n, count=5 * P, inverse=True)
    wc = d.to_numpy().reshape(ws.shape)
    d.free()
    base = [bytes(p) for p in cg.plonk_prove_batch(pk, ws, ps, bls, b"memo", P)]
    for rep in range(3):
        assert [bytes(p) for p in cg.plonk_prove_batch(pk, wc, ps, bls, b"memo", P, input_form="coeffs")] == base
        assert [bytes(p) for p in cg.plonk_prove_batch(pk, ws, ps, bls, b"memo", P)] == base
    cg.plonk_free_key(pk)
    cg.srs_free(h)
    ok += 1
print("capture stress:", ok, "iterations ok; graph stats", cg.plonk_graph_stats())
