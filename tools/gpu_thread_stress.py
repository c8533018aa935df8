"""Concurrency stress of the C ABI: threads making different calls at once (MSM, NTT, single proofs with and without
coalescing, batch proofs, verification) must all get the values a lone caller gets, and nothing may deadlock.
python tools/gpu_thread_stress.py [seconds]
(also run, bounded, by tests/test_gpu_fuzz.py)"""
import sys
import threading
import time

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu  # noqa: E402
from oracle import capref as cr  # noqa: E402  (checker: Jacobian -> affine)

def run(secs=10.0, log=print):
    """returns (calls per kind, list of errors)"""
    cg.init(0)
    tau = bu.SplitMix64(5).field()
    n = 1 << 9
    srs = cg.srs_generate(tau, n + 3)
    circuits = [bu.synthetic_circuit(9, 3, seed=4), bu.synthetic_circuit(9, 5, seed=5)]
    keys = [cg.plonk_preprocess(srs, n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont()) for sc in circuits]
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    wit = [sc.witness(1) for sc in circuits]
    wm = [sc.wires_mont(w[0]) for sc, w in zip(circuits, wit)]
    pm = [bu.to_mont_array(w[1]) for w in wit]
    bl = bu.to_mont_array(bu.blinders(1))
    scal = bu.random_canonical_scalars(1, n)
    data = bu.to_mont_array(list(range(1, 1 << 10)) + [7])
    ref = {
        "msm": cr.g1_to_affine(cg.msm_g1(srs, scal)).tobytes(),
        "ntt": cg.ntt_fr(data.copy(), 10, inverse=False, coset=True).tobytes(),
        "p0": bytes(cg.plonk_prove(keys[0][0], wm[0], pm[0], bl, b"a")),
        "p1": bytes(cg.plonk_prove(keys[1][0], wm[1], pm[1], bl, None)),
    }
    stop = [time.time() + secs]
    errors, counts = [], {}
    lock = threading.Lock()


    def worker(kind):
        k = 0
        try:
            while time.time() < stop[0]:
                if kind == "msm":
                    ok = cr.g1_to_affine(cg.msm_g1(srs, scal)).tobytes() == ref["msm"]   # a Jacobian triple is not unique
                elif kind == "ntt":
                    ok = cg.ntt_fr(data.copy(), 10, inverse=False, coset=True).tobytes() == ref["ntt"]
                elif kind == "p0":
                    ok = bytes(cg.plonk_prove(keys[0][0], wm[0], pm[0], bl, b"a")) == ref["p0"]
                elif kind == "p1":
                    ok = bytes(cg.plonk_prove(keys[1][0], wm[1], pm[1], bl, None)) == ref["p1"]
                elif kind == "batch":
                    got = cg.plonk_prove_batch(keys[0][0], np.stack([wm[0]] * 3), np.stack([pm[0]] * 3), np.stack([bl] * 3), b"a", 3)
                    ok = all(bytes(g) == ref["p0"] for g in got)
                else:
                    pr = cg.Proof.from_buffer_copy(ref["p0"])
                    ok = cg.plonk_verify(keys[0][1], h2, bh, pm[0], pr, b"a")
                if not ok:
                    errors.append(kind)
                k += 1
        except Exception as e:  # noqa: BLE001
            errors.append(f"{kind}: {e}")
        with lock:
            counts[kind] = counts.get(kind, 0) + k


    for coalesce in (0, 800):
        cg.plonk_set_coalescing(coalesce, 64)
        stop[0] = time.time() + secs / 2
        ths = [threading.Thread(target=worker, args=(k,)) for k in ("msm", "ntt", "p0", "p0", "p1", "p1", "batch", "verify")]
        for t in ths:
            t.start()
        for t in ths:
            t.join(timeout=secs * 10)
        assert not any(t.is_alive() for t in ths), "deadlock"
    cg.plonk_set_coalescing(0)
    log(f"calls: {counts} errors: {errors[:5]}")
    for k in keys:
        cg.plonk_free_key(k[0])
    cg.srs_free(srs)
    return counts, errors


if __name__ == "__main__":
    _, errs = run(float(sys.argv[1]) if len(sys.argv) > 1 else 10)
    sys.exit(1 if errs else 0)
