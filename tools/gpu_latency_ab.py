#!/usr/bin/env python3
"""Same-box A/B of the small-launch paths (round 4): one proof at a time (what the reference's criterion bench times,
benches/transfer.rs:103-105), small batches, and single MSMs, with the round-4 mechanisms switched on and off:
  CAPGPU_GRAPH_MAX_BATCH     hipGraph replay of the prover's kernel segments (0 = direct launches)
  CAPGPU_MSM_GRID_REDUCE     2-D grid bucket reduction for small batches (0 = bit planes)
  CAPGPU_PERM_INV_ON_DEVICE  round 2's inversion on a device thread (1) instead of the host (0)
  CAPGPU_MSM_SMALL_C / _MAX  window bits of the small-launch table (0 = none) and the largest launch, in MSMs, that takes it
  CAPGPU_MSM_QUAD_MAX / _QUAD_MAX_WIDE / _QUAD_FINAL   the tails of small launches on quads (0 = the one-lane kernels)
Every configuration runs in a process of its own (the switches are read once).  One JSON line per configuration.

    python tools/gpu_latency_ab.py [--quick]          (driver)
    python tools/gpu_latency_ab.py --child            (one configuration, taken from the environment)
"""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CONFIGS = [
    ("round3", {"CAPGPU_GRAPH_MAX_BATCH": "0", "CAPGPU_MSM_GRID_REDUCE": "0", "CAPGPU_PERM_INV_ON_DEVICE": "1",
                "CAPGPU_MSM_SMALL_C": "0"}),
    ("graphs_only", {"CAPGPU_GRAPH_MAX_BATCH": "16", "CAPGPU_MSM_GRID_REDUCE": "0", "CAPGPU_PERM_INV_ON_DEVICE": "1",
                     "CAPGPU_MSM_SMALL_C": "0"}),
    ("grid_only", {"CAPGPU_GRAPH_MAX_BATCH": "0", "CAPGPU_MSM_GRID_REDUCE": "1", "CAPGPU_PERM_INV_ON_DEVICE": "1",
                   "CAPGPU_MSM_SMALL_C": "0"}),
    ("hostinv_only", {"CAPGPU_GRAPH_MAX_BATCH": "0", "CAPGPU_MSM_GRID_REDUCE": "0", "CAPGPU_PERM_INV_ON_DEVICE": "0",
                      "CAPGPU_MSM_SMALL_C": "0"}),
    ("round4_small_table_c11", {"CAPGPU_MSM_SMALL_C": "11"}),
    ("round4_small_table_c12", {"CAPGPU_MSM_SMALL_C": "12"}),
    ("round4_one_lane_tails", {"CAPGPU_MSM_QUAD_MAX": "0", "CAPGPU_MSM_QUAD_MAX_WIDE": "0", "CAPGPU_MSM_QUAD_FINAL": "0",
                               "CAPGPU_NTT_TILE_ADAPT": "0"}),
    ("round4", {}),
]
R3 = {"CAPGPU_GRAPH_MAX_BATCH": "0", "CAPGPU_MSM_GRID_REDUCE": "0", "CAPGPU_PERM_INV_ON_DEVICE": "1", "CAPGPU_MSM_SMALL_C": "0"}


def child():
    import numpy as np
    from cap_amd import bench_utils as bu
    from cap_amd import lib as cg
    cg.init(0)
    cg.set_device(0)
    log_n, ni = 15, 27
    n = 1 << log_n
    tau = bu.SplitMix64(0xCA9).field()
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
    pk, _ = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
    wit = [sc.witness(3 + i) for i in range(4)]
    out = {"config": os.environ.get("CAPGPU_AB_NAME", "?")}
    ref = None
    for P in [int(x) for x in os.environ.get("CAPGPU_AB_BATCHES", "1,2,4,8,16").split(",")]:
        wires = np.stack([sc.wires_mont(wit[i % 4][0]) for i in range(P)])
        pubs = np.stack([bu.to_mont_array(wit[i % 4][1]) for i in range(P)])
        blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
        d = cg.DevBuf.from_numpy(wires)
        for _ in range(4):
            pr = cg.plonk_prove_batch_dev(pk, d, pubs, blind, b"ab", P)
        lat = []
        for _ in range(15):
            t0 = time.perf_counter()
            pr = cg.plonk_prove_batch_dev(pk, d, pubs, blind, b"ab", P)
            lat.append((time.perf_counter() - t0) * 1e3)
        lat.sort()
        out[f"batch{P}_ms"] = round(lat[len(lat) // 2], 4)
        if "proof0_sha" not in out:
            out["proof0_sha"] = __import__("hashlib").sha256(bytes(pr[0])).hexdigest()[:16]
        d.free()
    out["graph_stats"] = cg.plonk_graph_stats()
    # single MSMs: the latency chain of the 1-MSM launch
    for lg in (15, 17):
        m = 1 << lg
        h = cg.srs_generate_affine_seq(0x1234567, 0x89ABCDEF, m)
        sc_m = bu.random_canonical_scalars(5, m)
        d_sc, d_out = cg.DevBuf.from_numpy(sc_m), cg.DevBuf(96)
        for _ in range(3):
            cg.msm_g1_dev(h, d_sc, m, d_out=d_out)
        cg.sync()
        t0 = time.perf_counter()
        for _ in range(20):
            cg.msm_g1_dev(h, d_sc, m, d_out=d_out)
        cg.sync()
        out[f"msm_2p{lg}_ms"] = round((time.perf_counter() - t0) / 20 * 1e3, 4)
        out[f"msm_2p{lg}_sha"] = __import__("hashlib").sha256(
            __import__("numpy").ascontiguousarray(d_out.to_numpy()).tobytes()).hexdigest()[:12]
        d_sc.free()
        cg.srs_free(h)
    print(json.dumps(out), flush=True)


def main():
    if "--child" in sys.argv:
        return child()
    quick = "--quick" in sys.argv
    for name, env in ([CONFIGS[0], CONFIGS[4]] + CONFIGS[-2:] if quick else CONFIGS):
        e = dict(os.environ)
        e.update(env)
        e["CAPGPU_AB_NAME"] = name
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child"], env=e, capture_output=True, text=True)
        line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
        print(line[-1] if line else json.dumps({"config": name, "error": r.stderr[-600:]}), flush=True)


if __name__ == "__main__":
    main()
