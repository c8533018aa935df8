"""Row A12 (verify): the product's host-side verifier (pairing-based, no GPU needed) against the oracle.
Mirrors the reference's proof tests (src/proof/transfer.rs:599-760, mint.rs:344-471, freeze.rs:429-534):
a good proof verifies; wrong public input / proof / verifying key / bound data must fail.  (`-m "not gpu"`)"""
import copy
import ctypes

import numpy as np
import pytest

from cap_amd import bench_utils as bu
from cap_amd import lib as cg
from oracle import bn254 as bn
from oracle import capref as cr
from oracle import pairing as pr
from oracle import plonk as pl
from tests import helpers as H


def g2_words(q):
    """oracle G2 point ((x0, x1), (y0, y1)) -> 16 Montgomery words"""
    vals = [q[0][0], q[0][1], q[1][0], q[1][1]]
    return cr.ints_to_array([bn.to_mont(v, bn.P) for v in vals]).reshape(-1)


def g2_ints(w):
    v = [bn.from_mont(x, bn.P) for x in cr.array_to_ints(np.asarray(w, dtype=np.uint64).reshape(4, 4))]
    return ((v[0], v[1]), (v[2], v[3]))


def fill_words(dst, arr):
    arr = np.asarray(arr, dtype=np.uint64).reshape(-1)
    for i, v in enumerate(arr):
        dst[i] = int(v)


def make_vk(n, num_inputs, sel_pts, sig_pts):
    vk = cg.VerifyingKey()
    vk.domain_size, vk.num_inputs = n, num_inputs
    for i, k in enumerate(pl.K):
        fill_words(vk.k[i], cr.ints_to_array([bn.to_mont(k, bn.R)]))
    for i, p in enumerate(sel_pts):
        fill_words(vk.selector_comms[i], cr.points_to_array([p]))
    for i, p in enumerate(sig_pts):
        fill_words(vk.sigma_comms[i], cr.points_to_array([p]))
    return vk


def make_proof(pts, evals):
    pr_ = cg.Proof()
    for i in range(5):
        fill_words(pr_.wires_poly_comms[i], cr.points_to_array([pts[i]]))
        fill_words(pr_.split_quot_poly_comms[i], cr.points_to_array([pts[6 + i]]))
        fill_words(pr_.wires_evals[i], cr.ints_to_array([bn.to_mont(evals[i], bn.R)]))
    for i in range(4):
        fill_words(pr_.wire_sigma_evals[i], cr.ints_to_array([bn.to_mont(evals[5 + i], bn.R)]))
    fill_words(pr_.prod_perm_poly_comm, cr.points_to_array([pts[5]]))
    fill_words(pr_.opening_proof, cr.points_to_array([pts[11]]))
    fill_words(pr_.shifted_opening_proof, cr.points_to_array([pts[12]]))
    fill_words(pr_.perm_next_eval, cr.ints_to_array([bn.to_mont(evals[9], bn.R)]))
    return pr_


def test_g2_and_pairing_against_oracle():
    g2 = cg.g2_generator()
    assert g2_ints(g2) == pr.G2_GEN                       # EIP-197 generator
    a, b = 0x1234567, 0xABCDEF0123
    assert g2_ints(cg.g2_mul(g2, a)) == pr.g2_mul(pr.G2_GEN, a)
    assert not cg.g2_mul(g2, bn.R).any()                  # order r
    g1 = cr.points_to_array([bn.g1_mul(bn.G1_GEN, a), bn.g1_neg(bn.g1_mul(bn.G1_GEN, a * b % bn.R))])
    # bilinearity: e(aG, bH) * e(-abG, H) == 1, and it fails for a wrong exponent
    assert cg.pairing_check(g1, np.stack([cg.g2_mul(g2, b), g2]))
    assert not cg.pairing_check(g1, np.stack([cg.g2_mul(g2, b + 1), g2]))
    assert pr.pairing_product_is_one([(cr.affine_to_ints(g1[0]), pr.g2_mul(pr.G2_GEN, b)),
                                      (cr.affine_to_ints(g1[1]), pr.G2_GEN)])
    # non-degenerate: e(G, H) != 1
    assert not cg.pairing_check(cr.points_to_array([bn.G1_GEN]), g2[None])
    with pytest.raises(cg.CapGpuError):                   # off-curve input is refused, not mis-evaluated
        bad = g1.copy(); bad[0, 0] ^= 1
        cg.pairing_check(bad, np.stack([g2, g2]))


def test_oracle_pairing_verifier_matches_trapdoor_verifier(tau):
    sc = bu.synthetic_circuit(4, 2, seed=2)
    w, pubs = sc.witness(100)
    pk = pl.preprocess(pl.Circuit(n=sc.n, num_inputs=2, selectors=sc.selectors, sigma=sc.sigma), tau)
    proof = pl.prove(pk, w, pubs, bu.blinders(200), ext_msg=b"x")
    bh = pr.g2_mul(pr.G2_GEN, tau)
    args = (sc.n, 2, pk.selector_comms, pk.sigma_comms)
    assert pl.verify(*args, pubs, proof, tau, ext_msg=b"x")
    assert pl.verify_pairing(*args, pubs, proof, pr.G2_GEN, bh, ext_msg=b"x")
    bad = list(pubs); bad[0] = (bad[0] + 1) % bn.R
    assert not pl.verify_pairing(*args, bad, proof, pr.G2_GEN, bh, ext_msg=b"x")


def test_product_verifier_on_golden_proof(tau):
    """The committed golden proof (made by the Python oracle) is accepted by the product's C++ verifier, and
    every corruption the reference's tests try is rejected."""
    g = H.load_golden("proof_log5.json")
    n, nin = 1 << g["log_n"], g["num_inputs"]
    sc = bu.synthetic_circuit(g["log_n"], nin, seed=g["circuit_seed"])
    _, pubs = sc.witness(g["witness_seed"])
    sel = [H.unhex_pt(p) for p in g["selector_comms"]]
    sig = [H.unhex_pt(p) for p in g["sigma_comms"]]
    pts = [H.unhex_pt(p) for p in g["wires_poly_comms"]] + [H.unhex_pt(g["prod_perm_poly_comm"])] + \
        [H.unhex_pt(p) for p in g["split_quot_poly_comms"]] + [H.unhex_pt(g["opening_proof"]),
                                                              H.unhex_pt(g["shifted_opening_proof"])]
    evals = [int(h, 16) for h in g["wires_evals"] + g["wire_sigma_evals"] + [g["perm_next_eval"]]]
    vk, proof = make_vk(n, nin, sel, sig), make_proof(pts, evals)
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    assert np.array_equal(bh, g2_words(pr.g2_mul(pr.G2_GEN, tau)))
    pub_arr = bu.to_mont_array(pubs)
    msg = g["ext_msg"].encode()
    assert cg.plonk_verify(vk, h2, bh, pub_arr, proof, msg)
    # wrong public input
    bad_pub = pub_arr.copy(); bad_pub[1, 0] ^= 1
    assert not cg.plonk_verify(vk, h2, bh, bad_pub, proof, msg)
    # wrong bound data (ext_msg) / missing bound data
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, proof, msg + b"!")
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, proof, None)
    # wrong proof: an evaluation, a commitment, an opening
    p2 = make_proof(pts, [evals[0] + 1] + evals[1:]); assert not cg.plonk_verify(vk, h2, bh, pub_arr, p2, msg)
    pts3 = list(pts); pts3[7] = bn.g1_add(pts3[7], bn.G1_GEN)
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, make_proof(pts3, evals), msg)
    pts4 = list(pts); pts4[11] = bn.g1_add(pts4[11], bn.G1_GEN)
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, make_proof(pts4, evals), msg)
    pts5 = list(pts); pts5[0] = (pts5[0][0], (pts5[0][1] + 1) % bn.P)          # not even on the curve
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, make_proof(pts5, evals), msg)
    # wrong verifying key
    sig2 = list(sig); sig2[2] = bn.g1_add(sig2[2], bn.G1_GEN)
    assert not cg.plonk_verify(make_vk(n, nin, sel, sig2), h2, bh, pub_arr, proof, msg)
    # wrong SRS trapdoor
    assert not cg.plonk_verify(vk, h2, cg.g2_mul(h2, tau + 1), pub_arr, proof, msg)
    # malformed call: wrong number of public inputs is an argument error (reference: PlonkError)
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_verify(vk, h2, bh, pub_arr[:2], proof, msg)
    assert e.value.code == -1


def _add_to_words(words, modulus):
    """raw 256-bit integer held in 4 ctypes u64 words += modulus (must stay below 2^256)"""
    v = sum(int(words[i]) << (64 * i) for i in range(4)) + modulus
    assert v < 1 << 256
    for i in range(4):
        words[i] = (v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF


def test_verifier_rejects_non_canonical_encodings(tau):
    """arkworks field elements are canonical by construction, so v and v + r (or x and x + p) can never both be valid
    in the reference.  The raw Montgomery words of the C ABI could carry the second encoding (2r, 2p < 2^256):
    every field of the proof, every public input and every key point must be rejected in that form - otherwise a
    nullifier or Merkle root would have two accepted encodings (round-1 ADVICE, proof malleability)."""
    g = H.load_golden("proof_log5.json")
    n, nin = 1 << g["log_n"], g["num_inputs"]
    sc = bu.synthetic_circuit(g["log_n"], nin, seed=g["circuit_seed"])
    _, pubs = sc.witness(g["witness_seed"])
    sel = [H.unhex_pt(p) for p in g["selector_comms"]]
    sig = [H.unhex_pt(p) for p in g["sigma_comms"]]
    pts = [H.unhex_pt(p) for p in g["wires_poly_comms"]] + [H.unhex_pt(g["prod_perm_poly_comm"])] + \
        [H.unhex_pt(p) for p in g["split_quot_poly_comms"]] + [H.unhex_pt(g["opening_proof"]),
                                                              H.unhex_pt(g["shifted_opening_proof"])]
    evals = [int(h, 16) for h in g["wires_evals"] + g["wire_sigma_evals"] + [g["perm_next_eval"]]]
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    msg = g["ext_msg"].encode()
    pub_arr = bu.to_mont_array(pubs)
    vk0, proof0 = make_vk(n, nin, sel, sig), make_proof(pts, evals)
    assert cg.plonk_verify(vk0, h2, bh, pub_arr, proof0, msg)
    # public inputs: + r on each one in turn
    for i in range(nin):
        bad = pub_arr.copy()
        v = cr.array_to_ints(bad[i])[0] + bn.R
        bad[i] = cr.int_to_limbs(v)
        assert not cg.plonk_verify(vk0, h2, bh, bad, proof0, msg), f"public input {i} + r accepted"
        assert not cg.plonk_batch_verify([vk0], h2, bh, [bad], [proof0], [msg])
    # the 10 evaluations: + r
    ev_fields = [("wires_evals", i) for i in range(5)] + [("wire_sigma_evals", i) for i in range(4)] + \
        [("perm_next_eval", None)]
    for name, i in ev_fields:
        pr_ = make_proof(pts, evals)
        tgt = getattr(pr_, name) if i is None else getattr(pr_, name)[i]
        _add_to_words(tgt, bn.R)
        assert not cg.plonk_verify(vk0, h2, bh, pub_arr, pr_, msg), f"{name}[{i}] + r accepted"
    # the 13 proof points: + p on x, then on y
    pt_fields = [("wires_poly_comms", i) for i in range(5)] + [("prod_perm_poly_comm", None)] + \
        [("split_quot_poly_comms", i) for i in range(5)] + [("opening_proof", None), ("shifted_opening_proof", None)]
    for name, i in pt_fields:
        for off in (0, 4):
            pr_ = make_proof(pts, evals)
            tgt = getattr(pr_, name) if i is None else getattr(pr_, name)[i]
            view = (ctypes.c_uint64 * 4).from_buffer(tgt, 8 * off)
            _add_to_words(view, bn.P)
            assert not cg.plonk_verify(vk0, h2, bh, pub_arr, pr_, msg), f"{name}[{i}] coordinate {off // 4} + p accepted"
    # verifying-key points and coset constants
    for name, cnt in (("selector_comms", 13), ("sigma_comms", 5)):
        for i in range(cnt):
            vk = make_vk(n, nin, sel, sig)
            _add_to_words((ctypes.c_uint64 * 4).from_buffer(getattr(vk, name)[i], 0), bn.P)
            assert not cg.plonk_verify(vk, h2, bh, pub_arr, proof0, msg), f"vk.{name}[{i}].x + p accepted"
    vk = make_vk(n, nin, sel, sig)
    _add_to_words(vk.k[1], bn.R)
    assert not cg.plonk_verify(vk, h2, bh, pub_arr, proof0, msg)
    # G2 open key with a non-canonical coordinate is a malformed argument
    bad_h = np.asarray(h2, dtype=np.uint64).copy()
    bad_h[0:4] = cr.int_to_limbs(cr.array_to_ints(bad_h[0:4])[0] + bn.P)
    with pytest.raises(cg.CapGpuError):
        cg.plonk_verify(vk0, bad_h, bh, pub_arr, proof0, msg)
    # pairing_check refuses non-canonical G1 input
    g1 = cr.points_to_array([bn.G1_GEN])
    g1[0, 0:4] = cr.int_to_limbs(cr.array_to_ints(g1[0, 0:4])[0] + bn.P)
    with pytest.raises(cg.CapGpuError):
        cg.pairing_check(g1, np.asarray(h2)[None])


def test_batch_verify_and_proof_serialization(tau):
    """txn_batch_verify counterpart (src/lib.rs:455-529): proofs of two different circuits under one SRS."""
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    vks, pubs_l, proofs, msgs, oproofs = [], [], [], [], []
    for log_n, nin, seed in ((4, 2, 11), (5, 3, 12)):
        sc = bu.synthetic_circuit(log_n, nin, seed=seed)
        w, pubs = sc.witness(seed)
        pk = pl.preprocess(pl.Circuit(n=sc.n, num_inputs=nin, selectors=sc.selectors, sigma=sc.sigma), tau)
        msg = b"note-%d" % seed
        op = pl.prove(pk, w, pubs, bu.blinders(seed), ext_msg=msg)
        pts, ev = H.oracle_proof_points(op)
        vks.append(make_vk(sc.n, nin, pk.selector_comms, pk.sigma_comms))
        pubs_l.append(bu.to_mont_array(pubs))
        proofs.append(make_proof(pts, ev))
        msgs.append(msg)
        oproofs.append(op)
    assert all(cg.plonk_verify(vks[i], h2, bh, pubs_l[i], proofs[i], msgs[i]) for i in range(2))
    assert cg.plonk_batch_verify(vks, h2, bh, pubs_l, proofs, msgs)
    assert cg.plonk_batch_verify([], h2, bh, [], [], [])
    bad = pubs_l[1].copy(); bad[0, 0] ^= 1
    assert not cg.plonk_batch_verify(vks, h2, bh, [pubs_l[0], bad], proofs, msgs)
    assert not cg.plonk_batch_verify(vks, h2, bh, pubs_l, proofs, [msgs[0], b"x"])
    assert not cg.plonk_batch_verify(vks, h2, bh, pubs_l, [proofs[0], proofs[0]], msgs)
    # the device form of the batch verifier has no host path behind it: without an initialised GPU it says so
    if not H.gpu_present():
        with pytest.raises(cg.CapGpuError) as e:
            cg.plonk_batch_verify(vks, h2, bh, pubs_l, proofs, msgs, on_device=True)
        assert e.value.code in (-6, -2)
    # ark-serialize bytes of the proof = what the oracle's encoders give
    op = oproofs[0]
    exp = (5).to_bytes(8, "little") + b"".join(bn.g1_serialize_compressed(p) for p in op.wires_poly_comms)
    exp += bn.g1_serialize_compressed(op.prod_perm_poly_comm)
    exp += (5).to_bytes(8, "little") + b"".join(bn.g1_serialize_compressed(p) for p in op.split_quot_poly_comms)
    exp += bn.g1_serialize_compressed(op.opening_proof) + bn.g1_serialize_compressed(op.shifted_opening_proof)
    exp += (5).to_bytes(8, "little") + b"".join(bn.fr_to_bytes_le(v) for v in op.wires_evals)
    exp += (4).to_bytes(8, "little") + b"".join(bn.fr_to_bytes_le(v) for v in op.wire_sigma_evals)
    exp += bn.fr_to_bytes_le(op.perm_next_eval) + b"\x00"
    got = cg.proof_serialize(proofs[0])
    assert got == exp and len(got) == 769
    # and back (what reading a note from bytes does): same words, trailing bytes untouched, malformed encodings refused
    back, used = cg.proof_deserialize(got + b"trailer")
    assert used == 769 and bytes(back) == bytes(proofs[0])
    assert cg.plonk_verify(vks[0], h2, bh, pubs_l[0], back, msgs[0])
    bad_cases = {
        "short": got[:700],
        "wrong vector length": (4).to_bytes(8, "little") + got[8:],
        "scalar not canonical": got[:8 + 32 * 13 + 8 + 8] + bn.R.to_bytes(32, "little") + got[8 + 32 * 13 + 8 + 8 + 32:],
        "x off the curve": got[:8] + (4).to_bytes(32, "little") + got[40:],
        "both flags": got[:8 + 31] + bytes([got[8 + 31] | 0xC0]) + got[8 + 32:],
        "plookup proof": got[:-1] + b"\x01",
    }
    for name, blob in bad_cases.items():
        with pytest.raises(cg.CapGpuError) as e:
            cg.proof_deserialize(blob)
        assert e.value.code == cg.CAPGPU_ERR_SERIALIZATION, name
