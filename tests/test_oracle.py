"""CPU tests that pin the oracle: public known answers, Python <-> C agreement, golden fixtures,
and the PLONK restatement against the standard verifier equations.  (`-m "not gpu"`)"""
import copy

import numpy as np
import pytest

from cap_amd import bench_utils as bu
from oracle import bn254 as bn
from oracle import capref as cr
from oracle import plonk as pl
from tests import helpers as H


# ---- known answers that exist outside the reference ------------------------------------------------
def test_keccak256_known_answers():
    assert bn.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert bn.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    for msg in (b"", b"abc", b"x" * 135, b"y" * 136, b"z" * 137, bytes(range(256)) * 3):
        assert cr.keccak256(msg) == bn.keccak256(msg)


def test_bn254_constants_and_eip196_doubling():
    # SURVEY §7 step 1 constants
    assert bn.ROOT_OF_UNITY_28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert (-pow(bn.P, -1, 1 << 64)) % (1 << 64) == 0x87D20782E4866389
    assert (-pow(bn.R, -1, 1 << 64)) % (1 << 64) == 0xC2E1F593EFFFFFFF
    assert pow(bn.ROOT_OF_UNITY_28, 1 << 28, bn.R) == 1 and pow(bn.ROOT_OF_UNITY_28, 1 << 27, bn.R) != 1
    # EIP-196 (alt_bn128 precompile) known answer: 2 * (1, 2)
    assert bn.g1_add(bn.G1_GEN, bn.G1_GEN) == (
        1368015179489954701390400359078579693043519447331113978918064868415326638035,
        9918110051302171585080402603319702774565515993150576347155970296011118125764)
    assert bn.g1_mul(bn.G1_GEN, bn.R) is bn.INF
    assert bn.g1_mul(bn.G1_GEN, bn.R - 1) == bn.g1_neg(bn.G1_GEN)


def test_window_rule_matches_survey():
    # c = 12 for N = 16 386 and 32 768, 13 for 32 770, 65 538, 131 072, 18 for 2^24, 3 below 32
    for n, c in ((16386, 12), (32768, 12), (32770, 13), (65538, 13), (131072, 13), (1 << 24, 18), (31, 3), (32, 5)):
        assert bn.ark_window_size(n) == c
        assert cr.lib().capref_msm_window(n) == c


# ---- Python big ints <-> C restatement ------------------------------------------------------------
def test_field_ops_c_vs_python():
    rng = np.random.default_rng(1)
    for which, m in ((0, bn.P), (1, bn.R)):
        edge = [0, 1, 2, m - 1, m - 2, bn.MONT_R % m]
        vals = edge + [int.from_bytes(rng.bytes(32), "little") % m for _ in range(60)]
        rinv = bn.inv_mod(bn.MONT_R, m)
        for a in vals[:20]:
            for b in vals[::7]:
                assert cr.fp_op(which, "m", a, b) == a * b * rinv % m
                assert cr.fp_op(which, "a", a, b) == (a + b) % m
                assert cr.fp_op(which, "s", a, b) == (a - b) % m
            assert cr.fp_op(which, "n", a) == (-a) % m
            assert cr.fp_op(which, "t", a) == bn.to_mont(a, m)
            assert cr.fp_op(which, "f", a) == bn.from_mont(a, m)
            if a:
                assert bn.from_mont(cr.fp_op(which, "i", bn.to_mont(a, m)), m) == bn.inv_mod(a, m)
        assert cr.fp_op(which, "i", 0) == 0


def test_splitmix_inputs_agree():
    sm = bn.SplitMix64(7)
    assert cr.array_to_ints(cr.random_field(7, 1, 9, False)) == [sm.field(bn.R) for _ in range(9)]
    sm = bn.SplitMix64(8)
    assert cr.array_to_ints(cr.random_field(8, 0, 9, False)) == [sm.field(bn.P) for _ in range(9)]
    b = bu.SplitMix64(7)
    sm = bn.SplitMix64(7)
    assert [b.field() for _ in range(4)] == [sm.field(bn.R) for _ in range(4)]


def test_group_law_c_vs_python():
    ks = cr.random_field(5, 1, 6, False)
    pts_arr = cr.g1_fixed_base_batch(ks)
    pts = [cr.affine_to_ints(p) for p in pts_arr]
    kints = cr.array_to_ints(ks)
    assert pts == [bn.g1_mul(bn.G1_GEN, k) for k in kints]
    assert all(bn.is_on_curve(p) for p in pts)
    jac = [cr.g1_mul(pts_arr[i], 1) for i in range(6)]
    s = cr.g1_add(jac[0], jac[1])
    assert cr.affine_to_ints(cr.g1_to_affine(s)) == bn.g1_add(pts[0], pts[1])
    assert cr.affine_to_ints(cr.g1_to_affine(cr.g1_add(jac[2], jac[2]))) == bn.g1_add(pts[2], pts[2])
    neg = cr.g1_mul(pts_arr[3], bn.R - 1)
    assert cr.affine_to_ints(cr.g1_to_affine(cr.g1_add(jac[3], neg))) is None
    assert cr.affine_to_ints(cr.g1_to_affine(cr.g1_mul(pts_arr[4], 12345))) == bn.g1_mul(pts[4], 12345)


# ---- golden fixtures --------------------------------------------------------------------------------
@pytest.mark.parametrize("vec", H.load_golden("msm.json"), ids=lambda v: f"n{v['n']}{'e' if v['edge'] else ''}")
def test_msm_golden(vec):
    bases, scalars = H.msm_inputs(vec)
    exp = H.unhex_pt(vec["result"])
    n = vec["n"]
    assert bn.msm_pippenger(bases, scalars) == exp
    assert bn.msm_naive(bases, scalars) == exp
    got = cr.msm_g1(cr.points_to_array(bases), cr.ints_to_array(scalars))
    assert cr.affine_to_ints(cr.g1_to_affine(got)) == exp
    for c in (2, 5, 9):   # result does not depend on the window size
        got = cr.msm_g1(cr.points_to_array(bases), cr.ints_to_array(scalars), c)
        assert cr.affine_to_ints(cr.g1_to_affine(got)) == exp, (n, c)


def test_msm_empty_and_all_zero():
    assert cr.affine_to_ints(cr.g1_to_affine(cr.msm_g1(np.zeros((0, 8), np.uint64), np.zeros((0, 4), np.uint64)))) is None
    bases = cr.g1_fixed_base_batch(cr.random_field(3, 1, 50, False))
    assert cr.affine_to_ints(cr.g1_to_affine(cr.msm_g1(bases, np.zeros((50, 4), np.uint64)))) is None
    assert bn.msm_pippenger([cr.affine_to_ints(b) for b in bases], [0] * 50) is None


@pytest.mark.parametrize("vec", H.load_golden("ntt.json"), ids=lambda v: f"log{v['log_n']}")
def test_ntt_golden(vec):
    log_n = vec["log_n"]
    n = 1 << log_n
    rng = bn.SplitMix64(vec["seed"])
    a = [rng.field(bn.R) for _ in range(n)]
    arr = cr.ints_to_array([bn.to_mont(v, bn.R) for v in a])
    for key, inv, coset, fn in (("ntt", False, False, bn.ntt), ("intt", True, False, bn.intt),
                                ("coset_ntt", False, True, bn.coset_ntt), ("coset_intt", True, True, bn.coset_intt)):
        exp = [int(h, 16) for h in vec[key]]
        assert fn(a, log_n) == exp
        assert H.fr_to_ints(cr.ntt_fr(arr, log_n, inv, coset)) == exp
    if log_n <= 6:
        assert bn.ntt(a, log_n) == bn.dft_naive(a, bn.root_of_unity(log_n))   # the O(n^2) definition


def test_ntt_identities_medium():
    for log_n in (10, 13):
        a = cr.random_field(40 + log_n, 1, 1 << log_n, True)
        f = cr.ntt_fr(a, log_n, False, False)
        assert np.array_equal(cr.ntt_fr(f, log_n, True, False), a)
        assert np.array_equal(cr.ntt_fr(cr.ntt_fr(a, log_n, False, True), log_n, True, True), a)
        # out[j] = f(omega^j) and coset out[j] = f(5 * omega^j) for sampled j (Horner)
        w = bn.root_of_unity(log_n)
        fc = cr.ntt_fr(a, log_n, False, True)
        for j in (0, 1, 77, (1 << log_n) - 1):
            x = pow(w, j, bn.R)
            assert cr.poly_eval_fr(a, bn.to_mont(x, bn.R)) == cr.array_to_ints(f[j])[0]
            assert cr.poly_eval_fr(a, bn.to_mont(5 * x % bn.R, bn.R)) == cr.array_to_ints(fc[j])[0]


def test_known_tau_identity_cpu(tau):
    n = 300
    srs = H.srs_powers(tau, n)
    coef = cr.random_field(77, 1, n, False)
    got = cr.affine_to_ints(cr.g1_to_affine(cr.msm_g1(srs, coef)))
    assert got == bn.g1_mul(bn.G1_GEN, bn.poly_eval(cr.array_to_ints(coef), tau))


# ---- PLONK restatement ----------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def small_proof(tau):
    sc = bu.synthetic_circuit(4, 2, seed=2)
    w, pubs = sc.witness(100)
    bl = bu.blinders(200)
    c = pl.Circuit(n=sc.n, num_inputs=2, selectors=sc.selectors, sigma=sc.sigma, wires=w, pub_inputs=pubs)
    pl.check_circuit_satisfiability(c)
    pk = pl.preprocess(c, tau)
    pr = pl.prove(pk, w, pubs, bl, ext_msg=b"memo-key")
    return sc, c, pk, pr, w, pubs, bl


def test_plonk_oracle_proof_verifies(small_proof, tau):
    sc, c, pk, pr, w, pubs, bl = small_proof
    assert pl.verify(c.n, 2, pk.selector_comms, pk.sigma_comms, pubs, pr, tau, ext_msg=b"memo-key")


def test_plonk_oracle_rejects_corruptions(small_proof, tau):
    """mirrors the reference's negative tests (src/proof/transfer.rs:599-760): wrong public input,
    wrong proof, wrong verifying key, wrong bound data must all fail."""
    sc, c, pk, pr, w, pubs, bl = small_proof
    v = lambda pubs_=pubs, pr_=pr, sel=pk.selector_comms, sig=pk.sigma_comms, msg=b"memo-key": pl.verify(
        c.n, 2, sel, sig, pubs_, pr_, tau, ext_msg=msg)
    bad = list(pubs); bad[1] = (bad[1] + 1) % bn.R
    assert not v(pubs_=bad)
    assert not v(msg=b"memo-kex")
    assert not v(msg=None)
    p2 = copy.deepcopy(pr); p2.wires_evals[0] = (p2.wires_evals[0] + 1) % bn.R
    assert not v(pr_=p2)
    p3 = copy.deepcopy(pr); p3.opening_proof = bn.g1_add(p3.opening_proof, bn.G1_GEN)
    assert not v(pr_=p3)
    p4 = copy.deepcopy(pr); p4.split_quot_poly_comms[2] = bn.g1_add(p4.split_quot_poly_comms[2], bn.G1_GEN)
    assert not v(pr_=p4)
    sig = list(pk.sigma_comms); sig[0] = bn.g1_add(sig[0], bn.G1_GEN)
    assert not v(sig=sig)


def test_plonk_unsatisfied_witness_is_refused(small_proof):
    sc, c, pk, pr, w, pubs, bl = small_proof
    w2 = [list(col) for col in w]
    w2[4][5] = (w2[4][5] + 1) % bn.R
    with pytest.raises(pl.PlonkError):
        pl.check_circuit_satisfiability(pl.Circuit(n=c.n, num_inputs=2, selectors=sc.selectors, sigma=sc.sigma,
                                                   wires=w2, pub_inputs=pubs))
    with pytest.raises(pl.PlonkError):
        pl.prove(pk, w2, pubs, bl)


def test_plonk_golden_and_c_restatement(tau):
    g = H.load_golden("proof_log5.json")
    sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
    w, pubs = sc.witness(g["witness_seed"])
    bl = bu.blinders(g["blinder_seed"])
    n = sc.n
    exp_pts = [H.unhex_pt(p) for p in g["wires_poly_comms"]] + [H.unhex_pt(g["prod_perm_poly_comm"])] + \
        [H.unhex_pt(p) for p in g["split_quot_poly_comms"]] + [H.unhex_pt(g["opening_proof"]),
                                                              H.unhex_pt(g["shifted_opening_proof"])]
    exp_ev = [int(h, 16) for h in g["wires_evals"] + g["wire_sigma_evals"] + [g["perm_next_eval"]]]
    # C restatement (real Pippenger MSM + radix-2 FFT) reproduces the golden proof bit for bit
    key = cr.PlonkKey(H.srs_powers(tau, n + 3), n, g["num_inputs"], sc.selectors_mont(), sc.sigma_mont())
    assert [cr.affine_to_ints(c) for c in key.vk_comms] == [H.unhex_pt(p) for p in g["selector_comms"] + g["sigma_comms"]]
    rc, comms, evals = key.prove(sc.wires_mont(w), bu.to_mont_array(pubs), bu.to_mont_array(bl), g["ext_msg"].encode())
    assert rc == 0
    pts, ev = H.cref_proof_points(comms, evals)
    assert pts == exp_pts and ev == exp_ev
    # and refuses an unsatisfied witness with the error code the ABI uses
    wb = sc.wires_mont(w); wb[4, n // 2, 0] ^= 1
    assert key.prove(wb, bu.to_mont_array(pubs), bu.to_mont_array(bl))[0] == -7
    # a second witness / no public-input-free message under the same key: C vs Python
    w2, pubs2 = sc.witness(101)
    bl2 = bu.blinders(201)
    pk = pl.preprocess(pl.Circuit(n=n, num_inputs=g["num_inputs"], selectors=sc.selectors, sigma=sc.sigma), tau)
    pr = pl.prove(pk, w2, pubs2, bl2, ext_msg=None)
    rc, comms, evals = key.prove(sc.wires_mont(w2), bu.to_mont_array(pubs2), bu.to_mont_array(bl2), None)
    assert rc == 0 and H.cref_proof_points(comms, evals) == H.oracle_proof_points(pr)
