"""One process, N GPUs (capgpu_init(device_ids, n), cap_amd/csrc/context.hpp) - on the ONE GPU a test box has.

The reference is one process whose rayon threads each call prove() (src/utils/params_builder.rs:194-226), so the library
itself spreads work over the devices it was given.  With CAPGPU_ALLOW_DUPLICATE_DEVICES=1 device 0 may be bound twice:
two contexts - two streams, two sets of resident tables, two locks - that run exactly the code two GPUs would, except
that a replica on the "other device" is the same memory.  Checked here:
  * batches with host-resident witnesses are dealt over the contexts, the proofs bit-identical to the one-context run
    (the full-size 64-proof mix of BASELINE config 4 included);
  * handles are process-wide: a key / SRS made on one context works on the other;
  * an SRS above the sharding threshold is cut by point range over the contexts and every MSM form on it (host, batch,
    device-resident, sub-ranges across the cut) equals the unsharded result and the known answer;
  * coalesced single-proof calls are served by both contexts.
"""
import os
import threading

import numpy as np
import pytest

from cap_amd import bench_utils as bu
from oracle import bn254 as bn
from oracle import capref as cr
from tests import helpers as H

pytestmark = pytest.mark.gpu

A_SEQ, B_SEQ = 0x1234567890ABCDEF1234567890ABCDEF % bn.R, 0xFEDCBA0987654321FEDCBA % bn.R
# contexts a plain cg.init(0) gives in this environment: four on one bound device unless CAPGPU_CONTEXTS_PER_DEVICE says
# otherwise (the suite is also run with 1)
AMBIENT_CONTEXTS = max(int(os.environ.get("CAPGPU_CONTEXTS_PER_DEVICE", "4") or 4), 1)


@pytest.fixture(scope="module")
def cg2(cg):
    """the library re-initialised with device 0 bound twice; restored to the session's single context afterwards"""
    cg.shutdown()
    os.environ["CAPGPU_ALLOW_DUPLICATE_DEVICES"] = "1"
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "4096"
    ambient = os.environ.get("CAPGPU_CONTEXTS_PER_DEVICE")
    os.environ["CAPGPU_CONTEXTS_PER_DEVICE"] = "1"       # two "devices", one context each
    try:
        cg.init(devices=[0, 0])
        assert cg.device_count() == 2
        yield cg
    finally:
        cg.shutdown()
        del os.environ["CAPGPU_ALLOW_DUPLICATE_DEVICES"]
        del os.environ["CAPGPU_SHARD_MIN_POINTS"]
        if ambient is None:
            del os.environ["CAPGPU_CONTEXTS_PER_DEVICE"]
        else:
            os.environ["CAPGPU_CONTEXTS_PER_DEVICE"] = ambient
        cg.init(0)
        assert cg.device_count() == AMBIENT_CONTEXTS


def test_a_device_may_be_listed_once(cg):
    """(runs on the session's context: a second capgpu_init is a no-op, so the refusal is checked after a shutdown)"""
    cg.shutdown()
    try:
        with pytest.raises(cg.CapGpuError) as e:
            cg.init(devices=[0, 0])
        assert e.value.code == -1 and "twice" in str(e.value)
        with pytest.raises(cg.CapGpuError) as e:
            cg.init(devices=[0, 99])
        assert e.value.code == -1
        assert cg.device_count() == 0
    finally:
        cg.init(0)
    assert cg.device_count() == AMBIENT_CONTEXTS
    with pytest.raises(cg.CapGpuError):
        cg.set_device(AMBIENT_CONTEXTS)
    cg.set_device(0)
    assert cg.get_device() == (0, 0)
    cg.set_device(-1)


def _witnesses(sc, count, seed):
    ws, ps, bls = [], [], []
    for p in range(count):
        w, pubs = sc.witness(seed + p)
        ws.append(sc.wires_mont(w))
        ps.append(bu.to_mont_array(pubs))
        bls.append(bu.to_mont_array(bu.blinders(seed + 1000 + p)))
    return np.stack(ws), np.stack(ps), np.stack(bls)


def test_host_batches_are_dealt_over_the_contexts(cg2, tau):
    cg = cg2
    sc = bu.synthetic_circuit(10, 4, seed=21)
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = _witnesses(sc, 37, 300)
    cg.set_device(0)                                   # a bound thread keeps its batch on its context
    cg.profile_reset()
    one = cg.plonk_prove_batch(pkh, ws, ps, bls, b"deal", 37)
    cg.set_device(-1)
    cg.profile_enable(True)
    cg.profile_reset()
    two = cg.plonk_prove_batch(pkh, ws, ps, bls, b"deal", 37)     # unbound: parts of 18 and 19 proofs
    st = cg.profile_stats()
    cg.profile_enable(False)
    assert [bytes(p) for p in one] == [bytes(p) for p in two]
    assert st["k_quotient"][1] == 2, st["k_quotient"]             # one quotient launch per part: both contexts worked
    # ... and against the CPU restatement
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    for i in (0, 18, 36):
        rc, comms, evals = key.prove(ws[i], ps[i], bls[i], b"deal")
        assert rc == 0 and H.proof_points(two[i]) == H.cref_proof_points(comms, evals)
    # a batch too small to cut stays whole
    cg.profile_enable(True)
    cg.profile_reset()
    small = cg.plonk_prove_batch(pkh, ws[:9], ps[:9], bls[:9], b"deal", 9)
    assert cg.profile_stats()["k_quotient"][1] == 1
    cg.profile_enable(False)
    assert [bytes(p) for p in small] == [bytes(p) for p in one[:9]]
    # an unsatisfied witness in one part fails the call with that part's error
    bad = ws.copy()
    bad[30, 0, 5, 0] ^= 1
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_prove_batch(pkh, bad, ps, bls, b"deal", 37)
    assert e.value.code == -7
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_handles_are_process_wide(cg2, tau):
    """an SRS and a key made by a thread on context 0 serve a thread bound to context 1 (replicated on first use)"""
    cg = cg2
    sc = bu.synthetic_circuit(8, 3, seed=5)
    cg.set_device(0)
    h = cg.srs_generate(tau, 3000)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 3, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = _witnesses(sc, 2, 11)
    k = cr.random_field(9, 1, 3000, False)
    want_msm = cr.g1_to_affine(cg.msm_g1(h, k))
    want = cg.plonk_prove_batch(pkh, ws, ps, bls, None, 2)
    out = {}

    def other():
        try:
            cg.set_device(1)
            assert cg.get_device()[0] == 1
            out["msm"] = cr.g1_to_affine(cg.msm_g1(h, k))
            d = cg.DevBuf.from_numpy(k)                             # device memory of context 1
            out["msm_dev"] = cr.g1_to_affine(cg.msm_g1_dev(h, d, 3000).to_numpy())
            out["proofs"] = cg.plonk_prove_batch(pkh, ws, ps, bls, None, 2)
            out["srs"] = cg.srs_download(h, 5, 10)
            d.free()
        except Exception as ex:                                      # noqa: BLE001
            out["error"] = ex

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert "error" not in out, out.get("error")
    assert np.array_equal(out["msm"], want_msm) and np.array_equal(out["msm_dev"], want_msm)
    assert [bytes(p) for p in out["proofs"]] == [bytes(p) for p in want]
    assert np.array_equal(out["srs"], cg.srs_download(h, 5, 10))
    cg.set_device(-1)
    cg.plonk_free_key(pkh)
    with pytest.raises(cg.CapGpuError) as e:                       # gone on every context
        cg.plonk_prove_batch(pkh, ws, ps, bls, None, 2)
    assert e.value.code == -4
    cg.srs_free(h)
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1(h, k)
    assert e.value.code == -4


def test_a_large_srs_is_sharded_by_point_range(cg2, tau):
    cg = cg2
    n = 10001                                                       # above the (lowered) threshold, odd: ragged shards
    h = cg.srs_generate_affine_seq(A_SEQ, B_SEQ, n)
    assert cg.srs_shards(h) == 2 and cg.srs_size(h) == n
    assert cg.msm_plan(h, n, 1)["shards"] == 2
    sc = bu.random_canonical_scalars(91, n)
    sc[0] = 0
    sc[1] = cr.int_to_limbs(1)
    sc[5000] = cr.int_to_limbs(bn.R - 1)

    def expect(s, lo=0):
        s0, s1 = bu.weighted_scalar_sums(s, lo)
        return bn.g1_mul(bn.G1_GEN, (A_SEQ * s0 + B_SEQ * s1) % bn.R)

    def aff(j):
        return cr.affine_to_ints(cr.g1_to_affine(j))

    assert aff(cg.msm_g1(h, sc)) == expect(sc)
    # sub-ranges: inside shard 0, inside shard 1, across the cut, empty
    for off, m in ((0, 4000), (6000, 4001), (4990, 25), (3000, 5500), (777, 0)):
        assert aff(cg.msm_g1(h, sc[off:off + m], offset=off)) == expect(sc[off:off + m], off), (off, m)
    # several MSMs of unequal ranges in one call
    got = cg.msm_g1_batch(h, [sc[:6000], sc[4000:9000], sc[9000:]], offsets=[0, 4000, 9000])
    assert [aff(g) for g in got] == [expect(sc[:6000]), expect(sc[4000:9000], 4000), expect(sc[9000:], 9000)]
    # device-resident scalars (Montgomery form, three arrays, a stride): the slices travel to the shards' devices
    scs = np.stack([bu.random_canonical_scalars(92 + i, 7000) for i in range(3)])
    d = cg.DevBuf.from_numpy(cr.vec_to_mont(1, scs.reshape(-1, 4)))
    out = cg.msm_g1_dev(h, d, 6500, count=3, stride=7000, montgomery=True, offset=2000).to_numpy().reshape(3, 12)
    for i in range(3):
        assert aff(out[i]) == expect(scs[i, :6500], 2000), i
    d.free()
    # the sharded table is the unsharded one, cut
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "100000"
    h1 = cg.srs_generate_affine_seq(A_SEQ, B_SEQ, n)
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "4096"
    assert cg.srs_shards(h1) == 1
    assert np.array_equal(cg.srs_download(h, 4990, 30), cg.srs_download(h1, 4990, 30))
    assert np.array_equal(cr.g1_to_affine(cg.msm_g1(h1, sc)), cr.g1_to_affine(cg.msm_g1(h, sc)))
    # powers of tau, uploaded from host memory (the other way an SRS arrives)
    host = cg.srs_download(h1, 0, n)
    h2 = cg.srs_upload(host)
    assert cg.srs_shards(h2) == 2
    assert aff(cg.msm_g1(h2, sc)) == expect(sc)
    ht = cg.srs_generate(tau, 5000)
    assert cg.srs_shards(ht) == 2
    f = bu.random_canonical_scalars(95, 5000)
    ftau = bn.from_mont(cr.poly_eval_fr(cr.vec_to_mont(1, f), bn.to_mont(tau, bn.R)), bn.R)
    assert aff(cg.msm_g1(ht, f)) == bn.g1_mul(bn.G1_GEN, ftau)
    # a sharded SRS commits no proving key: refused, not mis-served
    circ = bu.synthetic_circuit(6, 2, seed=1)
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_preprocess(ht, circ.n, 2, circ.selectors_mont(), circ.sigma_mont())
    assert e.value.code == -1 and "sharded" in str(e.value)
    for x in (h, h1, h2, ht):
        cg.srs_free(x)


def test_coalesced_calls_use_every_context(cg2, tau):
    cg = cg2
    sc = bu.synthetic_circuit(9, 3, seed=8)
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 3, sc.selectors_mont(), sc.sigma_mont())
    T = 24
    ws, ps, bls = _witnesses(sc, T, 40)
    want = cg.plonk_prove_batch(pkh, ws, ps, bls, b"c", T)
    got, errs = [None] * T, []
    cg.plonk_set_coalescing(300, 6)                                  # small batches: both contexts get some
    b0, p0 = cg.plonk_coalescing_stats()

    def worker(i):
        try:
            got[i] = cg.plonk_prove(pkh, ws[i], ps[i], bls[i], b"c")
        except Exception as ex:                                      # noqa: BLE001
            errs.append(ex)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(T)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    cg.plonk_set_coalescing(0, 0)
    assert not errs, errs
    assert [bytes(p) for p in got] == [bytes(p) for p in want]
    b1, p1 = cg.plonk_coalescing_stats()
    assert p1 - p0 == T and 4 <= b1 - b0 < T
    cg.plonk_free_key(pkh)
    cg.srs_free(h)


def test_mixed_batch_of_64_full_size_split_over_two_contexts(cg2, tau):
    """BASELINE config 4 at full size: 32 transfer 2x3 + 19 freeze 3 (n = 2^15, one mixed-key device batch) and 13 mint
    (n = 2^14), dealt over two contexts - every proof the bytes of the one-context run."""
    cg = cg2
    mix = [("transfer_2x3", 32), ("mint", 13), ("freeze_3", 19)]
    n_max = 1 << max(bu.NOTE_SHAPES[k][0] for k, _ in mix)
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = str(1 << 20)            # (the module lowers it; a commit key is never cut)
    h = cg.srs_generate(tau, n_max + 3)
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "4096"
    assert cg.srs_shards(h) == 1
    groups = []
    for gi, (kind, count) in enumerate(mix):
        sc = bu.note_circuit(kind, seed=40 + gi)
        pkh, vk = cg.plonk_preprocess(h, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
        wit = [sc.witness(500 + 10 * gi + i) for i in range(3)]
        wires = np.stack([sc.wires_mont(wit[i % 3][0]) for i in range(count)])
        pubs = np.stack([bu.to_mont_array(wit[i % 3][1]) for i in range(count)])
        blind = np.stack([bu.to_mont_array(bu.blinders(900 + 100 * gi + i)) for i in range(count)])
        msg = None if kind == "mint" else b"txn-memo-ver-key"
        groups.append((pkh, wires, pubs, blind, msg, count, sc.n))
    same = [g for g in groups if g[6] == n_max]
    mint = [g for g in groups if g[6] != n_max][0]
    max_in = max(g[2].shape[1] for g in same)
    rows = np.concatenate([np.pad(g[2], ((0, 0), (0, max_in - g[2].shape[1]), (0, 0))) for g in same])
    handles = [g[0] for g in same for _ in range(g[5])]
    wires = np.concatenate([g[1] for g in same])
    blind = np.concatenate([g[3] for g in same])
    msgs = [g[4] for g in same for _ in range(g[5])]

    def run():
        a = cg.plonk_prove_multi(handles, wires, rows, blind, msgs)
        b = cg.plonk_prove_batch(mint[0], mint[1], mint[2], mint[3], mint[4], mint[5])
        return [bytes(p) for p in a + b]

    cg.set_device(0)
    one = run()
    cg.set_device(-1)
    cg.profile_enable(True)
    cg.profile_reset()
    two = run()
    launches = cg.profile_stats()["k_quotient"][1]
    cg.profile_enable(False)
    assert launches == 3                  # the 51-proof batch ran as two parts, the 13 mints (< 2 x 8) stayed whole
    assert one == two and len(set(one)) == 64
    for g in groups:
        cg.plonk_free_key(g[0])
    cg.srs_free(h)


def test_two_contexts_under_concurrent_mixed_load(cg2, tau):
    """Lock-order check by brute force: for a few seconds, threads make every kind of call that takes context locks in
    a different way - dealt host batches (one lock per part, from helper threads), MSMs on a sharded SRS (all locks, in
    slot order), coalesced single proofs (a leader's try-lock plus a helper's lock), bound device-pointer calls, handle
    creation / replication / release - and every result must equal the lone caller's; nothing may deadlock."""
    import time
    cg = cg2
    sc = bu.synthetic_circuit(9, 3, seed=8)
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = str(1 << 20)
    h = cg.srs_generate(tau, sc.n + 3)                               # a commit key: never sharded
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "4096"
    hs = cg.srs_generate_affine_seq(A_SEQ, B_SEQ, 9000)             # sharded over the two contexts
    assert cg.srs_shards(h) == 1 and cg.srs_shards(hs) == 2
    pkh, _vk = cg.plonk_preprocess(h, sc.n, 3, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = _witnesses(sc, 20, 700)
    k9 = bu.random_canonical_scalars(3, 9000)
    ref_batch = [bytes(p) for p in cg.plonk_prove_batch(pkh, ws, ps, bls, b"x", 20)]
    ref_msm = cr.g1_to_affine(cg.msm_g1(hs, k9)).tobytes()
    ref_small = cr.g1_to_affine(cg.msm_g1(h, k9[:sc.n])).tobytes()
    stop = time.time() + 4.0
    errors, counts = [], {}
    lock = threading.Lock()

    def worker(kind, slot):
        n_ok = 0
        try:
            if slot is not None:
                cg.set_device(slot)
            while time.time() < stop:
                if kind == "batch":
                    ok = [bytes(p) for p in cg.plonk_prove_batch(pkh, ws, ps, bls, b"x", 20)] == ref_batch
                elif kind == "sharded":
                    ok = cr.g1_to_affine(cg.msm_g1(hs, k9)).tobytes() == ref_msm
                elif kind == "single":
                    i = n_ok % 20
                    ok = bytes(cg.plonk_prove(pkh, ws[i], ps[i], bls[i], b"x")) == ref_batch[i]
                elif kind == "dev":
                    d = cg.DevBuf.from_numpy(np.ascontiguousarray(k9[:sc.n]))
                    ok = cr.g1_to_affine(cg.msm_g1_dev(h, d, sc.n).to_numpy()).tobytes() == ref_small
                    d.free()
                else:                                                 # handles come and go on both contexts
                    h2 = cg.srs_generate(tau, 700)
                    ok = cg.srs_size(h2) == 700 and cg.msm_g1(h2, k9[:700]) is not None
                    cg.srs_free(h2)
                if not ok:
                    errors.append(kind)
                n_ok += 1
        except Exception as ex:                                      # noqa: BLE001
            errors.append(f"{kind}: {ex}")
        with lock:
            counts[kind] = counts.get(kind, 0) + n_ok

    cg.plonk_set_coalescing(300, 16)
    plan = [("batch", None), ("batch", None), ("sharded", None), ("sharded", 1), ("single", None), ("single", None),
            ("single", None), ("dev", 0), ("dev", 1), ("handles", None)]
    ths = [threading.Thread(target=worker, args=a) for a in plan]
    for t in ths:
        t.start()
    for t in ths:
        t.join(timeout=120)
    cg.plonk_set_coalescing(0, 0)
    assert not any(t.is_alive() for t in ths), "deadlock"
    assert not errors, errors[:5]
    assert all(counts.get(k, 0) > 0 for k in ("batch", "sharded", "single", "dev", "handles")), counts
    cg.plonk_free_key(pkh)
    cg.srs_free(h)
    cg.srs_free(hs)


def test_scalars_resident_with_their_point_ranges(cg2, tau):
    """"GPU g holds its bases resident and receives the matching scalar slice".  capgpu_msm_g1_dev on a sharded
    SRS scatters the caller's scalars over the devices on every call; a scalar set (capgpu_msm_scalars_upload /
    _scatter_dev) places them once, and capgpu_msm_g1_resident then moves nothing but the 96-byte partials - asserted on
    the library's own byte counters."""
    cg = cg2
    n = 20011
    h = cg.srs_generate_affine_seq(A_SEQ, B_SEQ, n)
    assert cg.srs_shards(h) == 2
    scs = np.stack([bu.random_canonical_scalars(70 + i, n) for i in range(3)])

    def expect(s, lo=0):
        s0, s1 = bu.weighted_scalar_sums(s, lo)
        return bn.g1_mul(bn.G1_GEN, (A_SEQ * s0 + B_SEQ * s1) % bn.R)

    def aff(j):
        return cr.affine_to_ints(cr.g1_to_affine(j))

    cg.set_device(0)
    d = cg.DevBuf.from_numpy(scs)
    # the fallback form: every call scatters the slice of the other context again
    s0 = cg.msm_shard_stats()
    cg.msm_g1_dev(h, d, n, count=3)
    s1 = cg.msm_shard_stats()
    cg.msm_g1_dev(h, d, n, count=3)
    s2 = cg.msm_shard_stats()
    per_call = s1["scalar_bytes"] - s0["scalar_bytes"]
    assert per_call == 3 * 32 * (n // 2) and s2["scalar_bytes"] - s1["scalar_bytes"] == per_call
    # resident: scattered once ...
    hs = cg.msm_scalars_scatter_dev(h, d, n, count=3)
    s3 = cg.msm_shard_stats()
    assert s3["scalar_bytes"] - s2["scalar_bytes"] == per_call
    # ... and not again, call after call
    for _ in range(2):
        out = cg.msm_g1_resident(h, hs, count=3).to_numpy().reshape(3, 12)
        assert [aff(o) for o in out] == [expect(scs[i]) for i in range(3)]
    s4 = cg.msm_shard_stats()
    assert s4["scalar_bytes"] == s3["scalar_bytes"], "resident scalar slices were copied again"
    assert s4["partial_bytes"] - s3["partial_bytes"] == 2 * 3 * 96          # one partial per MSM from the other context
    assert s4["sharded_calls"] - s3["sharded_calls"] == 2
    # from host memory, a sub-range across the cut, Montgomery form
    sub = scs[1, 4000:17000]
    hu = cg.msm_scalars_upload(h, cr.vec_to_mont(1, sub).reshape(-1, 4), offset=4000)
    assert aff(cg.msm_g1_resident(h, hu, montgomery=True).to_numpy()) == expect(sub, 4000)
    # the same API on an unsharded SRS: one slice on the caller's context
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = str(1 << 20)
    h1 = cg.srs_generate(tau, 3000)
    os.environ["CAPGPU_SHARD_MIN_POINTS"] = "4096"
    f = bu.random_canonical_scalars(77, 3000)
    h1s = cg.msm_scalars_upload(h1, f)
    ftau = bn.from_mont(cr.poly_eval_fr(cr.vec_to_mont(1, f), bn.to_mont(tau, bn.R)), bn.R)
    assert aff(cg.msm_g1_resident(h1, h1s).to_numpy()) == bn.g1_mul(bn.G1_GEN, ftau)
    # a set belongs to its SRS; a freed set is gone
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1_resident(h1, hs, count=3)
    assert e.value.code == -1
    for x in (hs, hu, h1s):
        cg.msm_scalars_free(x)
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1_resident(h, hs, count=3)
    assert e.value.code == -4
    d.free()
    cg.set_device(-1)
    cg.srs_free(h)
    cg.srs_free(h1)


def test_peer_info_and_callers_hip_device(cg2):
    """capgpu_init settles the memory path of every bound device pair (here: one device bound twice = 2); and a call that
    the library places on a context must leave the CALLER's HIP current device alone (an integration next to torch)."""
    import ctypes
    cg = cg2
    assert cg.device_peer_info(0, 1) == 2 and cg.device_peer_info(0, 0) == 2
    with pytest.raises(cg.CapGpuError):
        cg.device_peer_info(0, 2)
    hip = ctypes.CDLL("libamdhip64.so")
    dev = ctypes.c_int(-1)
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0
    before = dev.value
    x = cr.random_field(6, 1, 1 << 10, True)
    cg.set_device(-1)
    cg.ntt_fr(x, 10, False, True)                                   # unbound host-buffer call: dealt to a context
    assert hip.hipGetDevice(ctypes.byref(dev)) == 0 and dev.value == before


@pytest.fixture()
def cg_replicas(cg):
    """device 0 bound twice with CAPGPU_FORCE_REPLICATE=1: the second context COPIES the SRS and key tables instead of
    sharing them, so that clone_srs_to_current / clone_key_to_current - and proofs made from the copies - run on the one
    GPU of a test box"""
    cg.shutdown()
    env = {"CAPGPU_ALLOW_DUPLICATE_DEVICES": "1", "CAPGPU_FORCE_REPLICATE": "1", "CAPGPU_CONTEXTS_PER_DEVICE": "1"}
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        cg.init(devices=[0, 0])
        yield cg
    finally:
        cg.shutdown()
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        cg.init(0)


def test_replicated_tables_give_the_same_results(cg_replicas, tau):
    cg = cg_replicas
    sc = bu.synthetic_circuit(10, 4, seed=33)
    cg.set_device(0)
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = _witnesses(sc, 5, 90)
    want = [bytes(p) for p in cg.plonk_prove_batch(pkh, ws, ps, bls, b"r", 5)]
    k = cr.random_field(12, 1, sc.n, False)
    want_msm = cr.g1_to_affine(cg.msm_g1(h, k))
    r0 = cg.msm_shard_stats()["replications"]
    out = {}

    def other():
        try:
            cg.set_device(1)
            out["proofs"] = [bytes(p) for p in cg.plonk_prove_batch(pkh, ws, ps, bls, b"r", 5)]
            out["msm"] = cr.g1_to_affine(cg.msm_g1(h, k))
            out["srs"] = cg.srs_download(h, 0, 64)
            out["key"] = cg.plonk_key_serialize(pkh, cg.g2_generator(), cg.g2_mul(cg.g2_generator(), tau))
        except Exception as ex:                                      # noqa: BLE001
            out["error"] = ex

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert "error" not in out, out.get("error")
    assert cg.msm_shard_stats()["replications"] - r0 == 2           # the SRS tables and the key, once each
    assert out["proofs"] == want and np.array_equal(out["msm"], want_msm)
    assert np.array_equal(out["srs"], cg.srs_download(h, 0, 64))
    assert out["key"] == cg.plonk_key_serialize(pkh, cg.g2_generator(), cg.g2_mul(cg.g2_generator(), tau))
    # the replica stays valid for later calls and is released with the handle
    cg.set_device(-1)
    assert [bytes(p) for p in cg.plonk_prove_batch(pkh, np.concatenate([ws] * 4), np.concatenate([ps] * 4),
                                                   np.concatenate([bls] * 4), b"r", 20)] == want * 4
    cg.plonk_free_key(pkh)
    cg.srs_free(h)
