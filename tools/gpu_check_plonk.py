"""Ad-hoc GPU check: device prover vs the Python oracle prover on small domains (development aid)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402
from oracle import bn254 as bn  # noqa: E402
from oracle import capref as cr  # noqa: E402
from oracle import plonk as pl  # noqa: E402

cg.init(0)
tau = bn.SplitMix64(0xCA9).field(bn.R)
ok = True


def pt(words):
    return cr.affine_to_ints(np.asarray(words, dtype=np.uint64))


def fr(words):
    return bn.from_mont(cr.array_to_ints(np.asarray(words, dtype=np.uint64))[0], bn.R)


for log_n, nin in [(5, 3), (6, 0), (8, 7)]:
    sc = bu.synthetic_circuit(log_n, nin, seed=2)
    n = sc.n
    h = cg.srs_generate(tau, n + 3)
    pkh, vk = cg.plonk_preprocess(h, n, nin, sc.selectors_mont(), sc.sigma_mont())
    c0 = pl.Circuit(n=n, num_inputs=nin, selectors=sc.selectors, sigma=sc.sigma)
    pko = pl.preprocess(c0, tau)
    vk_sel = [pt(vk.selector_comms[i]) for i in range(13)]
    vk_sig = [pt(vk.sigma_comms[i]) for i in range(5)]
    good = vk_sel == pko.selector_comms and vk_sig == pko.sigma_comms
    print(f"log_n={log_n} vk commitments", "OK" if good else "MISMATCH")
    ok &= good
    P = 3
    ws, pubs, bls = [], [], []
    for p in range(P):
        w, pu = sc.witness(100 + p)
        ws.append(w); pubs.append(pu); bls.append(bu.blinders(200 + p))
    wires = np.stack([sc.wires_mont(w) for w in ws])
    pub_arr = np.stack([bu.to_mont_array(pu) if pu else np.zeros((0, 4), np.uint64) for pu in pubs])
    bl_arr = np.stack([bu.to_mont_array(b) for b in bls])
    t = time.time()
    proofs = cg.plonk_prove_batch(pkh, wires, pub_arr, bl_arr, ext_msg=b"memo-key", count=P)
    tg = time.time() - t
    for p in range(P):
        t = time.time()
        exp = pl.prove(pko, ws[p], pubs[p], bls[p], ext_msg=b"memo-key")
        tc = time.time() - t
        g = proofs[p]
        checks = {
            "wires": [pt(g.wires_poly_comms[i]) for i in range(5)] == exp.wires_poly_comms,
            "z": pt(g.prod_perm_poly_comm) == exp.prod_perm_poly_comm,
            "quot": [pt(g.split_quot_poly_comms[i]) for i in range(5)] == exp.split_quot_poly_comms,
            "w_evals": [fr(g.wires_evals[i]) for i in range(5)] == exp.wires_evals,
            "s_evals": [fr(g.wire_sigma_evals[i]) for i in range(4)] == exp.wire_sigma_evals,
            "z_next": fr(g.perm_next_eval) == exp.perm_next_eval,
            "open": pt(g.opening_proof) == exp.opening_proof,
            "shifted": pt(g.shifted_opening_proof) == exp.shifted_opening_proof,
        }
        good = all(checks.values())
        ok &= good
        print(f"log_n={log_n} proof {p}: {'OK' if good else 'MISMATCH ' + str(checks)} gpu batch {tg*1e3:.1f} ms, python {tc:.1f} s", flush=True)
    # unsatisfied witness must be refused
    bad = wires.copy()
    bad[0, 4, n // 2, 0] ^= 1
    try:
        cg.plonk_prove_batch(pkh, bad, pub_arr, bl_arr, count=P)
        print("unsatisfied witness: NOT refused"); ok = False
    except cg.CapGpuError as e:
        print("unsatisfied witness refused:", e)
    cg.plonk_free_key(pkh)
    cg.srs_free(h)
print("ALL OK" if ok else "FAILURES")
