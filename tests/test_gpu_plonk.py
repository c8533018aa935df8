"""Parity tests for the device prover (A6 / K7-K9): every element of the proof - 13 commitments in affine
form and 10 evaluations - must equal the oracle's on the same inputs, bit for bit."""
import numpy as np
import pytest

from cap_amd import bench_utils as bu
from cap_amd import proof as capproof
from oracle import bn254 as bn
from oracle import capref as cr
from oracle import plonk as pl
from tests import helpers as H

pytestmark = pytest.mark.gpu


def gpu_key(cg, tau, sc):
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
    return h, pkh, vk


def pubs_arr(pubs):
    return bu.to_mont_array(pubs) if pubs else np.zeros((0, 4), np.uint64)


def test_golden_proof_log5(cg, tau):
    g = H.load_golden("proof_log5.json")
    sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
    w, pubs = sc.witness(g["witness_seed"])
    bl = bu.blinders(g["blinder_seed"])
    h, pkh, vk = gpu_key(cg, tau, sc)
    vk_pts = [cr.affine_to_ints(np.ctypeslib.as_array(vk.selector_comms[i])) for i in range(13)] + \
             [cr.affine_to_ints(np.ctypeslib.as_array(vk.sigma_comms[i])) for i in range(5)]
    assert vk_pts == [H.unhex_pt(p) for p in g["selector_comms"] + g["sigma_comms"]]
    assert vk.domain_size == sc.n and vk.num_inputs == g["num_inputs"]
    pr = cg.plonk_prove_batch(pkh, sc.wires_mont(w)[None], pubs_arr(pubs)[None], bu.to_mont_array(bl)[None],
                              g["ext_msg"].encode(), 1)[0]
    pts, ev = H.proof_points(pr)
    exp_pts = [H.unhex_pt(p) for p in g["wires_poly_comms"]] + [H.unhex_pt(g["prod_perm_poly_comm"])] + \
        [H.unhex_pt(p) for p in g["split_quot_poly_comms"]] + [H.unhex_pt(g["opening_proof"]),
                                                              H.unhex_pt(g["shifted_opening_proof"])]
    assert pts == exp_pts
    assert ev == [int(x, 16) for x in g["wires_evals"] + g["wire_sigma_evals"] + [g["perm_next_eval"]]]
    # the oracle's verifier accepts the device proof, and rejects it under a wrong public input
    o = pl.Proof(pts[0:5], pts[5], pts[6:11], pts[11], pts[12], ev[0:5], ev[5:9], ev[9])
    sel, sig = vk_pts[:13], vk_pts[13:]
    assert pl.verify(sc.n, sc.num_inputs, sel, sig, pubs, o, tau, ext_msg=g["ext_msg"].encode())
    bad = list(pubs); bad[0] = (bad[0] + 1) % bn.R
    assert not pl.verify(sc.n, sc.num_inputs, sel, sig, bad, o, tau, ext_msg=g["ext_msg"].encode())
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


@pytest.mark.parametrize("log_n,nin,P", [(4, 1, 1), (6, 0, 3), (9, 27, 2), (11, 5, 4)])
def test_batch_vs_c_oracle(cg, tau, log_n, nin, P):
    sc = bu.synthetic_circuit(log_n, nin, seed=log_n)
    h, pkh, vk = gpu_key(cg, tau, sc)
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, nin, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = [], [], []
    for p in range(P):
        w, pubs = sc.witness(100 + p)
        ws.append(sc.wires_mont(w)); ps.append(pubs_arr(pubs)); bls.append(bu.to_mont_array(bu.blinders(200 + p)))
    msg = b"txn-memo-ver-key" if log_n != 6 else None
    proofs = cg.plonk_prove_batch(pkh, np.stack(ws), np.stack(ps), np.stack(bls), msg, P)
    for p in range(P):
        rc, comms, evals = key.prove(ws[p], ps[p], bls[p], msg)
        assert rc == 0
        assert H.proof_points(proofs[p]) == H.cref_proof_points(comms, evals), f"proof {p}"
    # the host-side API mirror gives the same object, and its verify() accepts / rejects like the reference's
    g2h = cg.g2_generator()
    srs = capproof.UniversalSrs(h, sc.n + 2, g2h, cg.g2_mul(g2h, tau))
    pk = capproof.ProvingKey(pkh, sc.n, nin, srs)
    single = capproof.prove(pk, ws[0], ps[0], bls[0], msg)
    assert H.proof_points(single) == H.proof_points(proofs[0])
    vkey = capproof.VerifyingKey(vk, sc.n, nin, srs.h, srs.beta_h)
    for p in range(P):
        capproof.verify(vkey, ps[p], proofs[p], msg)
    with pytest.raises(capproof.TxnApiError):
        capproof.verify(vkey, ps[0], proofs[0], b"other-bound-data")
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_batch_large_enough_for_the_wide_window_msm(cg, tau):
    """With P = 8 proofs of n = 2^12 the 5P-wide commitment launches (wires, quotient splits) take the c = 15 table and
    the two-level sort, while the P- and 2P-wide ones (z, opening proofs) stay on c = 13: both MSM plans inside one
    proof, every proof bit-exact vs the CPU restatement."""
    log_n, nin, P = 12, 9, 8
    sc = bu.synthetic_circuit(log_n, nin, seed=77)
    h, pkh, vk = gpu_key(cg, tau, sc)
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, nin, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = [], [], []
    for p in range(P):
        w, pubs = sc.witness(300 + p)
        ws.append(sc.wires_mont(w)); ps.append(pubs_arr(pubs)); bls.append(bu.to_mont_array(bu.blinders(400 + p)))
    proofs = cg.plonk_prove_batch(pkh, np.stack(ws), np.stack(ps), np.stack(bls), b"wide", P)
    for p in range(P):
        rc, comms, evals = key.prove(ws[p], ps[p], bls[p], b"wide")
        assert rc == 0
        assert H.proof_points(proofs[p]) == H.cref_proof_points(comms, evals), f"proof {p}"
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_transfer_note_shape_full_size(cg, tau):
    """BASELINE config 3: the full 2-in/2-out transfer-note proof (n = 2^15, 27 public inputs, 13 MSM of
    ~n+2 points, 7 iNTT(n), 26 coset (i)NTT(8n)) on one MI355X, bit-exact vs the CPU restatement."""
    sc = bu.note_circuit("transfer_2x2", seed=2)
    assert sc.n == 1 << 15 and sc.num_inputs == 27
    h, pkh, vk = gpu_key(cg, tau, sc)
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, 27, sc.selectors_mont(), sc.sigma_mont())
    vk_c = [cr.affine_to_ints(c) for c in key.vk_comms]
    vk_g = [cr.affine_to_ints(np.ctypeslib.as_array(vk.selector_comms[i])) for i in range(13)] + \
           [cr.affine_to_ints(np.ctypeslib.as_array(vk.sigma_comms[i])) for i in range(5)]
    assert vk_g == vk_c
    w, pubs = sc.witness(3)
    wm, pm, bm = sc.wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(3))
    w2, pubs2 = sc.witness(4)
    proofs = cg.plonk_prove_batch(pkh, np.stack([wm, sc.wires_mont(w2)]), np.stack([pm, pubs_arr(pubs2)]),
                                  np.stack([bm, bu.to_mont_array(bu.blinders(4))]), b"memo", 2)
    rc, comms, evals = key.prove(wm, pm, bm, b"memo")
    assert rc == 0
    assert H.proof_points(proofs[0]) == H.cref_proof_points(comms, evals)
    assert H.proof_points(proofs[1]) != H.proof_points(proofs[0])
    # prove -> verify round trip with the product's own (pairing) verifier, as the reference tests do
    # (src/proof/transfer.rs:599-760): good proofs pass; wrong public input / bound data / swapped proof fail
    g2h = cg.g2_generator()
    bh = cg.g2_mul(g2h, tau)
    assert cg.plonk_verify(vk, g2h, bh, pm, proofs[0], b"memo")
    assert cg.plonk_verify(vk, g2h, bh, pubs_arr(pubs2), proofs[1], b"memo")
    assert not cg.plonk_verify(vk, g2h, bh, pubs_arr(pubs2), proofs[0], b"memo")
    assert not cg.plonk_verify(vk, g2h, bh, pm, proofs[0], b"memO")
    bad = pm.copy(); bad[26, 0] ^= 1
    assert not cg.plonk_verify(vk, g2h, bh, bad, proofs[0], b"memo")
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


@pytest.mark.parametrize("kind", ["mint", "freeze_2"])
def test_mint_and_freeze_note_shapes_full_size(cg, tau, kind):
    """A2 / A3: mint (n = 2^14, 22 public inputs: src/proof/mint.rs:262-277) and freeze (2 inputs, n = 2^14,
    3 + 2k public inputs: src/proof/freeze.rs:331-344) go through the same prover; bit-exact vs the CPU restatement."""
    sc = bu.note_circuit(kind, seed=5)
    assert sc.n == 1 << 14
    h, pkh, vk = gpu_key(cg, tau, sc)
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
    w, pubs = sc.witness(8)
    wm, pm, bm = sc.wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(8))
    pr = cg.plonk_prove_batch(pkh, wm[None], pm[None], bm[None], None, 1)[0]   # mint has no extra bound data
    rc, comms, evals = key.prove(wm, pm, bm, None)
    assert rc == 0 and H.proof_points(pr) == H.cref_proof_points(comms, evals)
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_mixed_batch_shares_one_srs(cg, tau):
    """TxnsParams::generate_txns (src/utils/params_builder.rs:64-241): one SRS sized for the largest circuit, one
    proving key per note type, proofs of different types interleaved.  Small domains here; the full-size mixed
    batch is bench.py --workload mixed64."""
    shapes = {"transfer": (9, 27), "mint": (8, 22), "freeze": (8, 7)}
    h = cg.srs_generate(tau, (1 << 9) + 3)
    keys, circs = {}, {}
    for name, (log_n, nin) in shapes.items():
        sc = bu.synthetic_circuit(log_n, nin, seed=log_n + nin)
        circs[name] = sc
        keys[name] = cg.plonk_preprocess(h, sc.n, nin, sc.selectors_mont(), sc.sigma_mont())[0]
    srs_host = cg.srs_download(h, 0, (1 << 9) + 3)
    order = ["transfer", "mint", "transfer", "freeze", "freeze", "transfer"]
    for i, name in enumerate(order):
        sc = circs[name]
        w, pubs = sc.witness(50 + i)
        wm, pm, bm = sc.wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(60 + i))
        pr = cg.plonk_prove_batch(keys[name], wm[None], pm[None], bm[None], b"m", 1)[0]
        ck = cr.PlonkKey(srs_host, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
        rc, comms, evals = ck.prove(wm, pm, bm, b"m")
        assert rc == 0 and H.proof_points(pr) == H.cref_proof_points(comms, evals), name
    for k in keys.values():
        cg.plonk_free_key(k)
    cg.srs_free(h)


def test_unsatisfied_witness_and_bad_arguments(cg, tau):
    sc = bu.synthetic_circuit(6, 2, seed=9)
    h, pkh, vk = gpu_key(cg, tau, sc)
    w, pubs = sc.witness(1)
    wm = sc.wires_mont(w)
    bl = bu.to_mont_array(bu.blinders(1))
    bad = wm.copy(); bad[4, 20, 0] ^= 1
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_prove_batch(pkh, bad[None], pubs_arr(pubs)[None], bl[None], None, 1)
    assert e.value.code == -7
    with pytest.raises(capproof.TxnApiError) as e2:      # reference: TxnApiError::FailedSnark
        capproof.prove(capproof.ProvingKey(pkh, sc.n, 2, capproof.UniversalSrs(h, sc.n + 2)), bad, pubs_arr(pubs), bl)
    assert "FailedSnark" in str(e2.value)
    with pytest.raises(cg.CapGpuError) as e:             # wrong number of public inputs
        cg.plonk_prove_batch(pkh, wm[None], pubs_arr(pubs[:1])[None], bl[None], None, 1)
    assert e.value.code == -1
    # an 8-row domain is refused at preprocessing: its five split-quotient commitments would read 5 (n + 2) = 50
    # coefficients of a 48-point quotient array (found by tools/gpu_fuzz_prover.py; the smallest CAP circuit has 2^14 rows)
    sc8 = bu.synthetic_circuit(3, 1, seed=3)
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_preprocess(h, 8, 1, sc8.selectors_mont(), sc8.sigma_mont())
    assert e.value.code == -1
    with pytest.raises(cg.CapGpuError) as e:             # unknown key
        cg.plonk_prove_batch(424242, wm[None], pubs_arr(pubs)[None], bl[None], None, 1)
    assert e.value.code == -4
    # the C ABI takes bare pointers; the binding checks every array against the key's shape before the call
    assert cg.plonk_key_info(pkh) == (sc.n, 2, h)
    for bad_args in ((wm[None, :, :sc.n // 2], pubs_arr(pubs)[None], bl[None], 1),     # wires of another domain
                     (wm[None], pubs_arr(pubs)[None], bl[None], 2),                    # count > arrays
                     (wm[None], pubs_arr(pubs)[None], bl[None, :12], 1)):              # 12 blinders
        with pytest.raises(cg.CapGpuError) as e:
            cg.plonk_prove_batch(pkh, bad_args[0], bad_args[1], bad_args[2], None, bad_args[3])
        assert e.value.code == -1
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_prove_batch_dev(pkh, cg.DevBuf(32 * 5 * sc.n - 32), pubs_arr(pubs)[None], bl[None], None, 1)
    assert e.value.code == -1
    # and the library itself refuses a public-input count that differs from the key's
    pr_raw = (cg.Proof * 1)()
    import ctypes
    rc = cg.load().capgpu_plonk_prove_batch(ctypes.c_uint64(pkh), 1, cg._p(wm.reshape(-1)),
                                            cg._p(pubs_arr(pubs).reshape(-1)), ctypes.c_size_t(1), None,
                                            ctypes.c_size_t(0), cg._p(bl.reshape(-1)), pr_raw)
    assert rc == -1
    small = cg.srs_generate(tau, 10)                      # SRS too small for the circuit
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_preprocess(small, sc.n, 2, sc.selectors_mont(), sc.sigma_mont())
    assert e.value.code == -1
    # a good proof still goes through after the failures (no poisoned state)
    ok = cg.plonk_prove_batch(pkh, wm[None], pubs_arr(pubs)[None], bl[None], None, 1)[0]
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, 2, sc.selectors_mont(), sc.sigma_mont())
    rc, comms, evals = key.prove(wm, pubs_arr(pubs), bl, None)
    assert H.proof_points(ok) == H.cref_proof_points(comms, evals)
    for x in (small,):
        cg.srs_free(x)
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_concurrent_prove_calls_are_coalesced(cg, tau):
    """The reference proves notes from rayon worker threads, one prove() per note (src/utils/params_builder.rs:194-226).
    With coalescing on, 24 threads calling capgpu_plonk_prove at once - different witnesses, different bound data, one of
    them an unsatisfied witness - are served by a few device batches; every caller gets exactly the proof a lone call
    gives, and only the owner of the bad witness sees CAPGPU_ERR_PROOF."""
    import threading
    sc = bu.synthetic_circuit(10, 3, seed=44)
    h, pkh, vk = gpu_key(cg, tau, sc)
    T = 24
    ws, ps, bls, msgs = [], [], [], []
    for t in range(T):
        w, pubs = sc.witness(900 + t)
        ws.append(sc.wires_mont(w)); ps.append(pubs_arr(pubs)); bls.append(bu.to_mont_array(bu.blinders(950 + t)))
        msgs.append(b"memo-%d" % t if t % 3 else None)
    alone = [cg.plonk_prove(pkh, ws[t], ps[t], bls[t], msgs[t]) for t in range(T)]        # coalescing off
    ws_bad = [w.copy() for w in ws]
    ws_bad[5][4, 30, 0] ^= 1                                                              # caller 5's witness is wrong
    cg.plonk_set_coalescing(2000, 16)
    b0, p0 = cg.plonk_coalescing_stats()
    results = [None] * T
    start = threading.Barrier(T)

    def worker(t):
        start.wait()
        try:
            results[t] = cg.plonk_prove(pkh, ws_bad[t], ps[t], bls[t], msgs[t])
        except cg.CapGpuError as e:
            results[t] = e

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    cg.plonk_set_coalescing(0)
    b1, p1 = cg.plonk_coalescing_stats()
    for t in range(T):
        if t == 5:
            assert isinstance(results[t], cg.CapGpuError) and results[t].code == -7, results[t]
        else:
            assert not isinstance(results[t], Exception), (t, results[t])
            assert bytes(results[t]) == bytes(alone[t]), t
    assert p1 - p0 >= T - 1 and b1 - b0 <= 12, (b1 - b0, p1 - p0)      # a handful of device batches (typically 2-3), not 24
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_proofs_of_several_keys_in_one_device_batch(cg, tau):
    """capgpu_plonk_prove_multi: proofs of different circuits (keys) over one domain size share the device batch - what
    the reference's side-by-side proving of transfer / mint / freeze notes (src/utils/params_builder.rs:194-226) becomes
    on the device.  Every proof equals, bit for bit, the one its own key makes alone, whatever the order and the mix,
    with per-proof messages and different numbers of public inputs; keys of another domain size are refused."""
    log_n = 9
    n = 1 << log_n
    srs = cg.srs_generate(tau, n + 3)
    shapes = [(3, 11), (9, 12), (0, 13)]              # (public inputs, circuit seed): three different circuits
    circuits = [bu.synthetic_circuit(log_n, ni, seed=seed) for ni, seed in shapes]
    keys = [cg.plonk_preprocess(srs, n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont()) for sc in circuits]
    order = [0, 1, 1, 2, 0, 2, 1]
    max_in = max(ni for ni, _ in shapes)
    wires, rows, blinds, msgs, alone = [], [], [], [], []
    for i, k in enumerate(order):
        sc = circuits[k]
        w, pubs = sc.witness(500 + i)
        bl = bu.to_mont_array(bu.blinders(600 + i))
        msg = b"note-%d" % i if i != 3 else b""
        row = np.zeros((max_in, 4), np.uint64)
        if pubs:
            row[:len(pubs)] = bu.to_mont_array(pubs)
        row[len(pubs):] = 0xFFFF                       # garbage beyond a key's own inputs must be ignored
        wires.append(sc.wires_mont(w)); rows.append(row); blinds.append(bl); msgs.append(msg)
        alone.append(cg.plonk_prove_batch(keys[k][0], sc.wires_mont(w)[None], pubs_arr(pubs)[None], bl[None],
                                          msg or None, 1)[0])
    handles = [keys[k][0] for k in order]
    got = cg.plonk_prove_multi(handles, np.stack(wires), np.stack(rows), np.stack(blinds), msgs)
    for i in range(len(order)):
        assert bytes(got[i]) == bytes(alone[i]), i
    # the same through the device-resident form
    d = cg.DevBuf.from_numpy(np.stack(wires))
    got_dev = cg.plonk_prove_multi(handles, d, np.stack(rows), np.stack(blinds), msgs)
    assert [bytes(p) for p in got_dev] == [bytes(p) for p in alone]
    d.free()
    # and the product verifier accepts each under its own key
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    for i, k in enumerate(order):
        pubs = rows[i][:shapes[k][0]]
        assert cg.plonk_verify(keys[k][1], h2, bh, pubs, got[i], msgs[i] or None)
    # a key of another domain size cannot join the batch
    sc_small = bu.synthetic_circuit(log_n - 1, 3, seed=5)
    k_small = cg.plonk_preprocess(srs, n // 2, 3, sc_small.selectors_mont(), sc_small.sigma_mont())
    with pytest.raises((cg.CapGpuError, ValueError)):
        cg.plonk_prove_multi([handles[0], k_small[0]], np.stack(wires[:2]), np.stack(rows[:2]), np.stack(blinds[:2]))
    for pkh, _ in keys + [k_small]:
        cg.plonk_free_key(pkh)
    cg.srs_free(srs)


def test_concurrent_calls_for_different_keys_share_device_batches(cg, tau):
    """Coalescing gathers per (domain size, SRS), not per key: threads proving notes of three different circuits at once
    (the reference's generate_txns does exactly that) are served by a few mixed-key device batches, a circuit of another
    domain size by its own, and every caller still gets the proof a lone call gives."""
    import threading
    srs = cg.srs_generate(tau, (1 << 9) + 3)
    specs = [(9, 3, 21), (9, 8, 22), (9, 0, 23), (8, 2, 24)]          # (log n, public inputs, seed); the last: another domain
    circuits = [bu.synthetic_circuit(ln, ni, seed=sd) for ln, ni, sd in specs]
    keys = [cg.plonk_preprocess(srs, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())[0] for sc in circuits]
    T = 20
    args = []
    for t in range(T):
        k = t % len(circuits)
        w, pubs = circuits[k].witness(700 + t)
        args.append((keys[k], circuits[k].wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(800 + t)),
                     b"m%d" % t if t % 2 else None))
    alone = [cg.plonk_prove(*a) for a in args]
    cg.plonk_set_coalescing(3000, 64)
    b0, p0 = cg.plonk_coalescing_stats()
    results = [None] * T
    start = threading.Barrier(T)

    def worker(t):
        start.wait()
        try:
            results[t] = cg.plonk_prove(*args[t])
        except Exception as e:                                      # noqa: BLE001
            results[t] = e

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(T)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    cg.plonk_set_coalescing(0)
    b1, p1 = cg.plonk_coalescing_stats()
    for t in range(T):
        assert not isinstance(results[t], Exception), (t, results[t])
        assert bytes(results[t]) == bytes(alone[t]), t
    # 15 calls on n = 2^9 under three keys and 5 on n = 2^8: with per-key queues that would be at least four batches per
    # round of arrivals; mixed-key batches need two
    assert p1 - p0 == T and b1 - b0 <= 8, (b1 - b0, p1 - p0)
    for k in keys:
        cg.plonk_free_key(k)
    cg.srs_free(srs)


def test_batch_verifier_on_the_device_agrees_with_the_host_one(cg, tau):
    """capgpu_plonk_batch_verify_dev: the verifier's group arithmetic as two MSMs on the prover's
    kernels.  It accepts and rejects what the host batch verifier does - good batches of 1, 2 and 9 proofs under two keys,
    a wrong public input, a corrupted commitment (still on the curve), a proof under the wrong key, swapped messages."""
    srs = cg.srs_generate(tau, (1 << 8) + 3)
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    circuits = [bu.synthetic_circuit(8, 4, seed=61), bu.synthetic_circuit(7, 0, seed=62)]
    keys = [cg.plonk_preprocess(srs, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont()) for sc in circuits]
    vks, pubs_l, proofs, msgs = [], [], [], []
    for i in range(9):
        k = i % 2
        w, pubs = circuits[k].witness(300 + i)
        msg = b"n%d" % i if i % 3 else None
        pr = cg.plonk_prove(keys[k][0], circuits[k].wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(400 + i)), msg)
        vks.append(keys[k][1]); pubs_l.append(pubs_arr(pubs)); proofs.append(pr); msgs.append(msg)

    def both(v, p, pr, m):
        host = cg.plonk_batch_verify(v, h2, bh, p, pr, m)
        dev = cg.plonk_batch_verify(v, h2, bh, p, pr, m, on_device=True)
        assert host == dev
        return dev

    for cnt in (1, 2, 9):
        assert both(vks[:cnt], pubs_l[:cnt], proofs[:cnt], msgs[:cnt])
    assert cg.plonk_batch_verify([], h2, bh, [], [], [], on_device=True)
    bad_pub = [p.copy() for p in pubs_l]
    bad_pub[4][1, 0] ^= 1
    assert not both(vks, bad_pub, proofs, msgs)
    import copy
    bad_pr = [copy.deepcopy(p) for p in proofs]
    for k in range(8):
        bad_pr[3].wires_poly_comms[0][k] = proofs[5].wires_poly_comms[1][k]      # another point of the curve
    assert not both(vks, pubs_l, bad_pr, msgs)
    assert not both([vks[1]] + vks[1:], pubs_l, proofs, msgs) if circuits[0].num_inputs == circuits[1].num_inputs else True
    assert not both(vks, pubs_l, proofs, [msgs[1], msgs[0]] + msgs[2:])
    for pkh, _ in keys:
        cg.plonk_free_key(pkh)
    cg.srs_free(srs)
