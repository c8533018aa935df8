#!/usr/bin/env python3
"""Same-box A/B of the call coalescer under closed-loop callers (round 4): 64 host threads, each calling
capgpu_plonk_prove_ex for one proof (host wires) again as soon as it has the last one - the reference's rayon pattern
(src/utils/params_builder.rs:194-226).  CAPGPU_COALESCE_SPLIT = eighths of a gathered batch that go to the first of two
free contexts (4 = even halves, the round-3 behaviour; 3 = the default); CAPGPU_COALESCE_EARLY = 0: the callers of both
parts of a cut batch are released together (the behaviour before the second half of round 4).  One JSON line per configuration, each in a
process of its own.
    python tools/gpu_coalesce_ab.py [calls_per_thread]"""
import ctypes
import json
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(per_thread):
    import numpy as np
    from cap_amd import bench_utils as bu
    from cap_amd import lib as cg
    cg.init(0)
    log_n, ni, T = 15, 27, 64
    n = 1 << log_n
    tau = bu.SplitMix64(0xCA9).field()
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
    pk, _ = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
    wit = [sc.witness(3 + i) for i in range(4)]
    W = 64
    wires = np.stack([sc.wires_mont(wit[i % 4][0]) for i in range(W)])
    pubs = np.stack([bu.to_mont_array(wit[i % 4][1]) for i in range(W)])
    blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(W)])
    msg = bytes(range(32))
    L = cg.load()
    mbuf = (ctypes.c_uint8 * len(msg)).from_buffer_copy(msg)
    u64p = ctypes.POINTER(ctypes.c_uint64)
    out = {"config": os.environ.get("CAPGPU_AB_NAME"), "threads": T, "calls_per_thread": per_thread}
    for window in (500,):
        cg.plonk_set_coalescing(window, 256)
        proofs = [[cg.Proof() for _ in range(per_thread)] for _ in range(T)]
        calls = [[(ctypes.c_uint64(pk), wires[(t + k) % W].ctypes.data_as(u64p), pubs[(t + k) % W].ctypes.data_as(u64p),
                   ctypes.c_size_t(ni), mbuf, ctypes.c_size_t(len(msg)), blind[(t + k) % W].ctypes.data_as(u64p),
                   ctypes.c_int(0), ctypes.byref(proofs[t][k])) for k in range(per_thread)] for t in range(T)]
        for rep in range(2):                       # the first repetition warms every buffer up
            bar = threading.Barrier(T + 1)
            errs = []

            def worker(t):
                bar.wait()
                for a in calls[t]:
                    rc = L.capgpu_plonk_prove_ex(*a)
                    if rc:
                        errs.append(rc)
                        return

            b0, p0 = cg.plonk_coalescing_stats()
            ths = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
            for th in ths:
                th.start()
            bar.wait()
            t0 = time.perf_counter()
            for th in ths:
                th.join()
            dt = time.perf_counter() - t0
            b1, p1 = cg.plonk_coalescing_stats()
        out[f"window_{window}us"] = {"proofs_per_s": round(T * per_thread / dt, 1), "device_batches": b1 - b0,
                                     "errors": errs[:2]}
        cg.plonk_set_coalescing(0)
    # reference points on the same box: resident batches of 64 on one context, and as 2 x 32 on two
    d = cg.DevBuf.from_numpy(wires)
    cg.set_device(0)
    cg.plonk_prove_batch_dev(pk, d, pubs, blind, msg, W)
    t0 = time.perf_counter()
    for _ in range(4):
        cg.plonk_prove_batch_dev(pk, d, pubs, blind, msg, W)
    out["resident_batch64_one_context"] = round(4 * W / (time.perf_counter() - t0), 1)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    if "--child" in sys.argv:
        child(int(sys.argv[sys.argv.index("--child") + 1]))
    else:
        per = sys.argv[1] if len(sys.argv) > 1 else "8"
        for name, env in (("late_release", {"CAPGPU_COALESCE_EARLY": "0"}), ("default", {}),
                          ("even_halves", {"CAPGPU_COALESCE_SPLIT": "4"}), ("quarter", {"CAPGPU_COALESCE_SPLIT": "2"}),
                          ("three_contexts", {"CAPGPU_CONTEXTS_PER_DEVICE": "3"}),
                          ("four_contexts", {"CAPGPU_CONTEXTS_PER_DEVICE": "4"}),
                          ("one_context", {"CAPGPU_CONTEXTS_PER_DEVICE": "1"}), ("late_release", {"CAPGPU_COALESCE_EARLY": "0"}),
                          ("default", {})):
            e = dict(os.environ)
            e.update(env)
            e["CAPGPU_AB_NAME"] = name
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--child", per], env=e, capture_output=True, text=True)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            print(line[-1] if line else json.dumps({"config": name, "error": r.stderr[-500:]}), flush=True)
