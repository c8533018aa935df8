"""Parity tests proper for K1-K6: the HIP NTT and MSM, called through the C ABI on a real MI355X, against the
oracle on the same seeded inputs; bit-exact (integer work).  Full BASELINE sizes go through size-independent
identities (round trips, Horner samples, the known-tau identity)."""
import numpy as np
import pytest

from oracle import bn254 as bn
from oracle import capref as cr
from tests import helpers as H

pytestmark = pytest.mark.gpu


# ---- NTT -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("vec", H.load_golden("ntt.json"), ids=lambda v: f"log{v['log_n']}")
def test_ntt_golden(cg, vec):
    log_n = vec["log_n"]
    rng = bn.SplitMix64(vec["seed"])
    a = [rng.field(bn.R) for _ in range(1 << log_n)]
    arr = cr.ints_to_array([bn.to_mont(v, bn.R) for v in a])
    for key, inv, coset in (("ntt", False, False), ("intt", True, False), ("coset_ntt", False, True),
                            ("coset_intt", True, True)):
        assert H.fr_to_ints(cg.ntt_fr(arr, log_n, inv, coset)) == [int(h, 16) for h in vec[key]], key


@pytest.mark.parametrize("log_n", [9, 10, 11, 12, 14, 15, 16])
def test_ntt_vs_c_oracle(cg, log_n):
    a = H.seeded_fr(100 + log_n, 1 << log_n)
    for inv in (False, True):
        for coset in (False, True):
            got = cg.ntt_fr(a, log_n, inv, coset)
            assert np.array_equal(got.reshape(-1), cr.ntt_fr(a, log_n, inv, coset).reshape(-1)), (inv, coset)


@pytest.mark.parametrize("log_n", [17, 18, 19, 21])
def test_ntt_full_size_identities(cg, log_n):
    """2^17 is the north-star NTT size, 2^18 / 2^19 the quotient domains of n = 2^15 / 2^16."""
    n = 1 << log_n
    a = H.seeded_fr(7 + log_n, n)
    f = cg.ntt_fr(a, log_n, False, False)
    assert np.array_equal(cg.ntt_fr(f, log_n, True, False), a)
    fc = cg.ntt_fr(a, log_n, False, True)
    assert np.array_equal(cg.ntt_fr(fc, log_n, True, True), a)
    w = bn.root_of_unity(log_n)
    for j in (0, 1, 12345 % n, n // 2 + 3, n - 1):
        x = pow(w, j, bn.R)
        assert cr.poly_eval_fr(a, bn.to_mont(x, bn.R)) == cr.array_to_ints(f[j])[0]
        assert cr.poly_eval_fr(a, bn.to_mont(5 * x % bn.R, bn.R)) == cr.array_to_ints(fc[j])[0]
    # linearity: NTT(a + b) = NTT(a) + NTT(b) on a sample
    b = H.seeded_fr(99 + log_n, n)
    ai, bi = cr.array_to_ints(a[:64]), cr.array_to_ints(b[:64])
    s = a.copy()
    s[:64] = cr.ints_to_array([(x + y) % bn.R for x, y in zip(ai, bi)])
    s[64:] = 0
    a0 = a.copy(); a0[64:] = 0
    b0 = b.copy(); b0[64:] = 0
    fs, fa, fb = (cg.ntt_fr(x, log_n) for x in (s, a0, b0))
    for j in (0, 5, n - 1):
        assert cr.array_to_ints(fs[j])[0] == (cr.array_to_ints(fa[j])[0] + cr.array_to_ints(fb[j])[0]) % bn.R


def test_ntt_batch_and_device_resident(cg):
    log_n = 12
    arrs = [H.seeded_fr(500 + i, 1 << log_n) for i in range(5)]
    outs = cg.ntt_fr_batch(arrs, log_n, inverse=True, coset=True)
    for a, o in zip(arrs, outs):
        assert np.array_equal(o.reshape(-1), cr.ntt_fr(a, log_n, True, True).reshape(-1))
    stride = (1 << log_n) + 8
    host = np.zeros((3, stride, 4), dtype=np.uint64)
    for i in range(3):
        host[i, :1 << log_n] = arrs[i]
    d = cg.DevBuf.from_numpy(host)
    cg.ntt_fr_dev(d, log_n, count=3, stride=stride, coset=True)
    back = d.to_numpy().reshape(3, stride, 4)
    for i in range(3):
        assert np.array_equal(back[i, :1 << log_n].reshape(-1), cr.ntt_fr(arrs[i], log_n, False, True).reshape(-1))
        assert not back[i, 1 << log_n:].any()


def test_ntt_rejects_bad_arguments(cg):
    a = H.seeded_fr(1, 8)
    with pytest.raises(cg.CapGpuError) as e:
        cg.check(cg.load().capgpu_ntt_fr(a.ctypes.data_as(cg.u64p), 29, 0, 0))
    assert e.value.code == -1
    with pytest.raises(cg.CapGpuError):
        cg.check(cg.load().capgpu_ntt_fr(a.ctypes.data_as(cg.u64p), 3, 2, 0))


# ---- MSM -------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("vec", H.load_golden("msm.json"), ids=lambda v: f"n{v['n']}{'e' if v['edge'] else ''}")
def test_msm_golden(cg, vec):
    bases, scalars = H.msm_inputs(vec)
    h = cg.srs_upload(cr.points_to_array(bases))
    got = cg.msm_g1(h, cr.ints_to_array(scalars))
    assert cr.affine_to_ints(cr.g1_to_affine(got)) == H.unhex_pt(vec["result"])
    cg.srs_free(h)


@pytest.fixture(scope="module")
def srs13(cg):
    n = 1 << 13
    bases = cr.g1_fixed_base_batch(cr.random_field(11, 1, n, False))
    bases[5] = 0              # point at infinity among the bases
    bases[7] = bases[6]       # duplicate base (forces P + P in a bucket)
    return cg.srs_upload(bases), bases


@pytest.mark.parametrize("n", [0, 1, 2, 3, 31, 32, 33, 1000, 4099, 8192])
def test_msm_vs_c_oracle(cg, srs13, n):
    h, bases = srs13
    sc = cr.random_field(12 + n, 1, max(n, 1), False)[:n]
    if n >= 8:
        sc[0] = 0
        sc[1] = cr.int_to_limbs(1)
        sc[2] = cr.int_to_limbs(bn.R - 1)
        sc[3] = cr.int_to_limbs(2**13 - 1)
        sc[4] = cr.int_to_limbs(2**13)
        sc[6] = cr.int_to_limbs(5)
        sc[7] = cr.int_to_limbs(bn.R - 5)     # P and -P cancel to infinity inside one bucket
    got = cr.g1_to_affine(cg.msm_g1(h, sc))
    assert np.array_equal(got, cr.g1_to_affine(cr.msm_g1(bases[:n], sc)))


def test_msm_all_zero_scalars_and_non_canonical(cg, srs13):
    h, bases = srs13
    assert cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h, np.zeros((100, 4), np.uint64)))) is None
    # arkworks' MSM takes any 256-bit integer; so does this one (k and k mod r give the same point)
    big = np.full((3, 4), 0xFFFFFFFFFFFFFFFF, dtype=np.uint64)
    red = cr.ints_to_array([((1 << 256) - 1) % bn.R] * 3)
    assert np.array_equal(cr.g1_to_affine(cg.msm_g1(h, big)), cr.g1_to_affine(cg.msm_g1(h, red)))


def test_msm_batch_offsets_and_montgomery_scalars(cg, srs13):
    h, bases = srs13
    scs = [cr.random_field(500 + i, 1, 1000, False) for i in range(5)]
    got = cg.msm_g1_batch(h, scs, offsets=[3] * 5)
    for i in range(5):
        assert np.array_equal(cr.g1_to_affine(got[i]), cr.g1_to_affine(cr.msm_g1(bases[3:1003], scs[i])))
    got = cg.msm_g1_batch(h, [scs[0][:10], scs[1][:700]], offsets=[0, 100])   # ragged batch
    assert np.array_equal(cr.g1_to_affine(got[0]), cr.g1_to_affine(cr.msm_g1(bases[:10], scs[0][:10])))
    assert np.array_equal(cr.g1_to_affine(got[1]), cr.g1_to_affine(cr.msm_g1(bases[100:800], scs[1][:700])))
    # device-resident Montgomery scalars (polynomial coefficients as the prover holds them)
    mont = cr.vec_to_mont(1, scs[2])
    d = cg.DevBuf.from_numpy(mont)
    out = cg.msm_g1_dev(h, d, 1000, montgomery=True).to_numpy()
    assert np.array_equal(cr.g1_to_affine(out), cr.g1_to_affine(cr.msm_g1(bases[:1000], scs[2])))


def test_msm_linearity_and_g1_sum(cg, srs13):
    h, bases = srs13
    a = cr.random_field(71, 1, 2000, False)
    b = cr.random_field(72, 1, 2000, False)
    s = cr.ints_to_array([(x + y) % bn.R for x, y in zip(cr.array_to_ints(a), cr.array_to_ints(b))])
    pa, pb, ps = cg.msm_g1(h, a), cg.msm_g1(h, b), cg.msm_g1(h, s)
    assert np.array_equal(cr.g1_to_affine(cg.g1_sum(np.stack([pa, pb]))), cr.g1_to_affine(ps))
    assert np.array_equal(cr.g1_to_affine(cg.g1_sum(np.stack([pa, pb]))), cr.g1_to_affine(cr.g1_add(pa, pb)))
    # point-range sharding: partial sums over [0,700) and [700,2000) add up to the whole
    p0 = cg.msm_g1(h, a[:700])
    p1 = cg.msm_g1(h, a[700:], offset=700)
    assert np.array_equal(cr.g1_to_affine(cg.g1_sum(np.stack([p0, p1]))), cr.g1_to_affine(pa))


@pytest.mark.parametrize("log_n", [15, 17])
def test_msm_known_tau_identity_full_size(cg, tau, log_n):
    """BASELINE config 2: 2^17 points (Aztec CRS size).  With bases [tau^i]G the MSM of the coefficients of f
    must equal [f(tau)]G - one scalar multiplication checks the whole MSM exactly."""
    n = (1 << log_n) + (2 if log_n == 15 else 0)      # 32770 = the transfer-note commit size
    h = cg.srs_generate(tau, n)
    first = cg.srs_download(h, 0, 3)
    assert [cr.affine_to_ints(p) for p in first] == [bn.g1_mul(bn.G1_GEN, pow(tau, i, bn.R)) for i in range(3)]
    coef = cr.random_field(77 + log_n, 1, n, False)
    got = cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h, coef)))
    ftau = bn.from_mont(cr.poly_eval_fr(cr.vec_to_mont(1, coef), bn.to_mont(tau, bn.R)), bn.R)
    assert got == bn.g1_mul(bn.G1_GEN, ftau)
    cg.srs_free(h)


def test_msm_skewed_scalars_in_a_large_batch(cg, tau):
    """Worst case for a bucket method: every scalar of an MSM equal, so each window sends all n entries into ONE
    bucket.  In a batch large enough for the running-sum reduction (>= 64 MSMs at c = 13) the giant buckets are cut
    into many length-sorted work items and recombined; the batch also holds ordinary random MSMs and an all-zero one.
    Checked with the known-tau identity: sum_i k tau^i G = [k (tau^n - 1) / (tau - 1)] G, and against the same MSMs
    run one at a time (the single-MSM path reduces through bit planes instead)."""
    n, batch = 6000, 64
    h = cg.srs_generate(tau, n)
    geo = (pow(tau, n, bn.R) - 1) * pow(tau - 1, bn.R - 2, bn.R) % bn.R
    rng = bn.SplitMix64(4242)
    consts = [rng.field(bn.R) for _ in range(batch)]
    consts[1], consts[2], consts[3] = 1, bn.R - 1, (1 << 13) - 1
    scs = []
    for b in range(batch):
        if b % 4 == 3:
            scs.append(cr.random_field(9000 + b, 1, n, False))                 # ordinary MSM
        elif b == 8:
            scs.append(np.zeros((n, 4), np.uint64))                            # all zero: result is infinity
        else:
            scs.append(np.tile(cr.int_to_limbs(consts[b]), (n, 1)))            # one giant bucket per window
    got = cg.msm_g1_batch(h, scs)
    for b in range(batch):
        pt = cr.affine_to_ints(cr.g1_to_affine(got[b]))
        if b % 4 == 3:
            if b in (3, 35, 63):
                assert pt == cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h, scs[b]))), b
        elif b == 8:
            assert pt is None
        else:
            assert pt == bn.g1_mul(bn.G1_GEN, consts[b] * geo % bn.R), b
    # the same skewed MSM alone (32-entry items, bit-plane reduction)
    assert cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h, scs[0]))) == bn.g1_mul(bn.G1_GEN, consts[0] * geo % bn.R)
    cg.srs_free(h)


@pytest.mark.parametrize("n,batch,offset", [(4096, 32, 0), (5000, 33, 7), (20011, 40, 100), (65539, 32, 0)])
def test_msm_wide_window_path_vs_single_path(cg, tau, n, batch, offset):
    """Batches of >= 32 MSMs over >= 4096 points take the c = 15 table and the two-level sort; the same MSMs run one
    at a time take c = 13 and the bit-plane reduction.  Odd sizes, a base offset, Montgomery-form scalars; one member
    of every batch is also checked against the known-tau identity."""
    h = cg.srs_generate(tau, n + offset)
    rng = np.random.default_rng(n + batch)
    sc = rng.integers(0, 1 << 63, size=(batch, n, 4), dtype=np.uint64) * 2 + rng.integers(0, 2, size=(batch, n, 4), dtype=np.uint64)
    sc[:, :, 3] &= np.uint64((1 << 60) - 1)                     # < 2^252: canonical
    sc[1, :17] = 0                                               # a run of zero scalars
    got = cg.msm_g1_batch(h, [sc[b] for b in range(batch)], offsets=[offset] * batch)
    for b in (0, 1, batch // 2, batch - 1):
        one = cg.msm_g1(h, sc[b], offset=offset)
        assert np.array_equal(cr.g1_to_affine(got[b]), cr.g1_to_affine(one)), b
    ks = [int(w[0]) | int(w[1]) << 64 | int(w[2]) << 128 | int(w[3]) << 192 for w in sc[0]]
    acc, x = 0, pow(tau, offset, bn.R)
    for k in ks:
        acc = (acc + k * x) % bn.R
        x = x * tau % bn.R
    assert cr.affine_to_ints(cr.g1_to_affine(got[0])) == bn.g1_mul(bn.G1_GEN, acc)
    cg.srs_free(h)


@pytest.mark.parametrize("n,batch", [(9, 1), (2000, 1), (4096, 40)])
def test_msm_special_cases_inside_one_bucket(cg, n, batch):
    """The accumulation loop takes every special case - accumulator at infinity (first entry, or after P + (-P)), a base
    at infinity, P + P - through one test on the x-difference.  Here nine bases P, -P, Q, Q, -Q, R, infinity, P, -P carry
    the same small scalar d, so all of them meet in ONE bucket of window 0 (in whatever order the tile sort leaves them):
    the result must be d (Q + R).  Run on the one-level path (c = 9 / 13) and, padded with zero scalars to 4096 points
    in a batch of 40, on the two-level path (c = 15)."""
    P_, Q_, R_ = (bn.g1_mul(bn.G1_GEN, k) for k in (1234567, 7654321, 1357911))
    pts = [P_, bn.g1_neg(P_), Q_, Q_, bn.g1_neg(Q_), R_, None, P_, bn.g1_neg(P_)]
    filler = bn.g1_mul(bn.G1_GEN, 99)
    bases = cr.points_to_array(pts + [filler] * (n - len(pts)))
    h = cg.srs_upload(bases)
    for d in (1, 5, (1 << 12) + 1):
        sc = np.zeros((n, 4), np.uint64)
        sc[:len(pts), 0] = d
        want = bn.g1_mul(bn.g1_add(Q_, R_), d)
        if batch == 1:
            got = [cg.msm_g1(h, sc)]
        else:
            got = cg.msm_g1_batch(h, [sc] * batch)
        for g in (got[0], got[-1]):
            assert cr.affine_to_ints(cr.g1_to_affine(g)) == want, d
    # everything cancels: the accumulator ends at infinity
    sc = np.zeros((n, 4), np.uint64)
    sc[[0, 1, 7, 8], 0] = 3
    out = cg.msm_g1(h, sc) if batch == 1 else cg.msm_g1_batch(h, [sc] * batch)[0]
    assert cr.affine_to_ints(cr.g1_to_affine(out)) is None
    cg.srs_free(h)


@pytest.mark.parametrize("batch", [1, 5, 40, 140])
def test_msm_special_cases_between_buckets(cg, batch):
    """Equal, opposite and empty BUCKET sums: what the tails of a launch - bucket = sum of its items, row / column sums of
    the bucket grid, the weighted sums, the running sums and their finish - have to get right after the accumulation loop.
    Small launches run these tails on quads (one point per four lanes, quad29.hpp), whose general case (P + P, P - P) is an
    out-of-line path that random data never takes.  Bases are drawn from {P, -P, Q, infinity} only and the scalars are
    small, so neighbouring buckets, whole rows and columns of the grid, and the items of one bucket hold the same point,
    opposite points or nothing.  batch 1 / 5: narrow table (4096 buckets, 64 x 64 grid); 40: wide table (128 x 128 grid);
    140: wide table, running sums (msm_reduce_segments / msm_reduce_final)."""
    p_k, q_k = 1234567, 7654321
    P_, Q_ = bn.g1_mul(bn.G1_GEN, p_k), bn.g1_mul(bn.G1_GEN, q_k)
    kinds = [(P_, p_k), (bn.g1_neg(P_), bn.R - p_k), (Q_, q_k), (None, 0)]
    rng = np.random.default_rng(77)
    n = 4096
    which = rng.integers(0, 4, n)
    which[:600] = 0                                   # 600 copies of P ...
    bases = cr.points_to_array([kinds[w][0] for w in which])
    h = cg.srs_upload(bases)
    scs, want = [], []
    for b in range(min(batch, 6)):
        ks = rng.integers(0, 1 << 14, n)
        ks[:600] = 1 if b % 2 == 0 else 3             # ... in one bucket: its work items are equal points
        if b == 1:
            ks[600:] = rng.choice([1, 2, 65, 66, 129, 130, 4097], n - 600)   # neighbours, same row, same column
        if b == 2:
            ks[:] = 0
            ks[[0, 1]] = [1, 2]                       # P in buckets 0 and 1 only: a row sum P + P, nothing else
        sc = np.zeros((n, 4), np.uint64)
        sc[:, 0] = ks
        scs.append(sc)
        want.append(sum(int(k) * kinds[w][1] for k, w in zip(ks, which)) % bn.R)
    got = [cg.msm_g1(h, scs[0])] if batch == 1 else cg.msm_g1_batch(h, [scs[i % len(scs)] for i in range(batch)])
    for i, g in enumerate(got):
        w = want[i % len(scs)]
        assert cr.affine_to_ints(cr.g1_to_affine(g)) == (bn.g1_mul(bn.G1_GEN, w) if w else None), (batch, i)
    cg.srs_free(h)


def test_msm_batch_run_in_slices(cg, tau, monkeypatch):
    """Batches whose sort tables would outgrow 32-bit counters are run in slices of the batch (msm.hip: batch_slice);
    at test sizes the slicing is forced through CAPGPU_MSM_SLICE.  Results must not depend on it - for plain batches
    and for the prover's grouped addressing (5 polynomials per proof inside one array: slices start on a group
    boundary), checked through whole proofs."""
    n, batch = 4500, 37
    h = cg.srs_generate(tau, n)
    scs = [cr.random_field(3000 + b, 1, n, False) for b in range(batch)]
    want = [cr.g1_to_affine(p) for p in cg.msm_g1_batch(h, scs)]
    for limit in ("16", "7", "1"):
        monkeypatch.setenv("CAPGPU_MSM_SLICE", limit)
        assert cg.msm_plan(h, n, batch)["slice"] == int(limit)
        got = [cr.g1_to_affine(p) for p in cg.msm_g1_batch(h, scs)]     # (the Jacobian triple depends on the reduction used)
        assert all(np.array_equal(a, b) for a, b in zip(got, want)), limit
    cg.srs_free(h)
    # grouped addressing: a batch of 9 proofs at n = 2^12 (45-wide launches of 5 per proof) sliced to 12 -> 10 + 10 + ...
    from cap_amd import bench_utils as bu
    sc = bu.synthetic_circuit(12, 3, seed=31)
    hs = cg.srs_generate(tau, sc.n + 3)
    monkeypatch.delenv("CAPGPU_MSM_SLICE")
    pk, _ = cg.plonk_preprocess(hs, sc.n, 3, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = [], [], []
    for p in range(9):
        w, pubs = sc.witness(700 + p)
        ws.append(sc.wires_mont(w)); ps.append(bu.to_mont_array(pubs)); bls.append(bu.to_mont_array(bu.blinders(800 + p)))
    plain = cg.plonk_prove_batch(pk, np.stack(ws), np.stack(ps), np.stack(bls), b"s", 9)
    monkeypatch.setenv("CAPGPU_MSM_SLICE", "12")
    sliced = cg.plonk_prove_batch(pk, np.stack(ws), np.stack(ps), np.stack(bls), b"s", 9)
    assert [bytes(cg.proof_serialize(a)) for a in plain] == [bytes(cg.proof_serialize(b)) for b in sliced]
    cg.plonk_free_key(pk)
    cg.srs_free(hs)


def test_msm_affine_seq_bases(cg):
    """BASELINE config 5's synthetic bases P_i = [a + i b]G: sum k_i P_i = [sum k_i (a + i b)] G."""
    a, b, n = 12345678901234567890, 987654321987654321, 5000
    h = cg.srs_generate_affine_seq(a, b, n)
    sc = cr.random_field(5, 1, n, False)
    ks = cr.array_to_ints(sc)
    exp = bn.g1_mul(bn.G1_GEN, sum(k * (a + i * b) for i, k in enumerate(ks)) % bn.R)
    assert cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h, sc))) == exp
    cg.srs_free(h)


def test_msm_error_paths(cg, srs13):
    h, _ = srs13
    sc = cr.random_field(1, 1, 4, False)
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1(999999, sc)
    assert e.value.code == -4
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1(h, sc, offset=(1 << 13) - 2)
    assert e.value.code == -1
    assert b"offset" in cg.load().capgpu_last_error()
