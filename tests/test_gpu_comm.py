"""The multi-GPU exchange step inside the C ABI (cap_amd/csrc/comm.hip) on the one GPU a test box has: a
communicator of world size 1 goes through the same code - RCCL loaded, ncclCommInitRank, the all-gather of the 96-byte
partials on the library stream, the on-device sum - as a world of 8; the N > 1 arithmetic is covered by the 8-range test
in test_gpu_configs.py and the gloo test in test_dist_cpu.py."""
import numpy as np
import pytest

from cap_amd import bench_utils as bu
from oracle import bn254 as bn
from oracle import capref as cr
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture()
def comm1(cg):
    assert cg.comm_info() == (0, 0)
    cg.comm_init(0, 1, cg.comm_unique_id())
    yield cg
    cg.plonk_shard_msm(False)
    cg.comm_destroy()
    assert cg.comm_info() == (0, 0)


def test_sharded_msm_calls_need_a_communicator(cg, tau):
    h = cg.srs_generate(tau, 64)
    with pytest.raises(cg.CapGpuError) as e:
        cg.msm_g1_sharded(h, cr.random_field(1, 1, 64, False))
    assert e.value.code == -6
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_shard_msm(True)
    assert e.value.code == -6
    with pytest.raises(cg.CapGpuError) as e:
        cg.comm_init(3, 2, bytes(128))
    assert e.value.code == -1
    cg.srs_free(h)


def test_sharded_msm_world_1(comm1, tau):
    cg = comm1
    assert cg.comm_info() == (0, 1)
    with pytest.raises(cg.CapGpuError):                       # a second communicator is refused
        cg.comm_init(0, 1, cg.comm_unique_id())
    n = 5000
    h = cg.srs_generate(tau, n)
    sc = cr.random_field(31, 1, n, False)
    want = cr.g1_to_affine(cg.msm_g1(h, sc))
    assert np.array_equal(cr.g1_to_affine(cg.msm_g1_sharded(h, sc)), want)
    # three MSMs in one launch, one all-gather of 3 x 96 bytes; a base offset; Montgomery-form scalars
    scs = np.stack([cr.random_field(40 + i, 1, 1000, False) for i in range(3)])
    d = cg.DevBuf.from_numpy(cr.vec_to_mont(1, scs.reshape(-1, 4)))
    out = cg.msm_g1_sharded_dev(h, d, 1000, count=3, montgomery=True, offset=17).to_numpy().reshape(3, 12)
    for i in range(3):
        assert np.array_equal(cr.g1_to_affine(out[i]), cr.g1_to_affine(cg.msm_g1(h, scs[i], offset=17)))
    assert cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1_sharded(h, np.zeros((0, 4), np.uint64)))) is None
    cg.srs_free(h)


def test_prover_with_sharded_commitment_msms(comm1, tau):
    """BASELINE config 4, mode A: the prover's MSMs go through the point-range split + exchange; the proofs are the
    same bytes as without it (and as the oracle's)."""
    cg = comm1
    sc = bu.synthetic_circuit(10, 4, seed=12)
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = [], [], []
    for p in range(3):
        w, pubs = sc.witness(60 + p)
        ws.append(sc.wires_mont(w)); ps.append(bu.to_mont_array(pubs)); bls.append(bu.to_mont_array(bu.blinders(70 + p)))
    plain = cg.plonk_prove_batch(pkh, np.stack(ws), np.stack(ps), np.stack(bls), b"m", 3)
    cg.plonk_shard_msm(True)
    pk2, vk2 = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())   # preprocess shards too
    assert bytes(vk2) == bytes(vk)
    sharded = cg.plonk_prove_batch(pkh, np.stack(ws), np.stack(ps), np.stack(bls), b"m", 3)
    cg.plonk_shard_msm(False)
    for a, b in zip(plain, sharded):
        assert H.proof_points(a) == H.proof_points(b)
    key = cr.PlonkKey(cg.srs_download(h, 0, sc.n + 3), sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    rc, comms, evals = key.prove(ws[0], ps[0], bls[0], b"m")
    assert rc == 0 and H.proof_points(sharded[0]) == H.cref_proof_points(comms, evals)
    for k in (pkh, pk2):
        cg.plonk_free_key(k)
    cg.srs_free(h)


# ---- the N > 1 data path on one GPU: loopback communicator ---------------------------------------------------------
# A real world of k ranks needs k GPUs (RCCL refuses two ranks on one device).  The loopback communicator makes this
# process play the k ranks one after the other through the same payload layout, gather buffer, status words and
# g1_sum_ranks kernel; only the all-gather itself becomes a copy into the rank's slot.
@pytest.fixture(params=[2, 3, 8])
def loop(cg, request):
    assert cg.comm_info() == (0, 0)
    cg.comm_init_loopback(request.param)
    yield cg, request.param
    cg.plonk_shard_msm(False)
    cg.comm_destroy()
    assert cg.comm_info() == (0, 0)


def _split(n, world, rank):
    base, rem = divmod(n, world)
    return rank * base + min(rank, rem), base + (1 if rank < rem else 0)


def test_sharded_msm_loopback_world(loop, tau):
    cg, k = loop
    assert cg.comm_info() == (0, k)
    n, count, off = 5003, 3, 17                                      # ragged: n is not divisible by k
    h = cg.srs_generate(tau, n + off)
    scs = np.stack([cr.random_field(50 + i, 1, n, False) for i in range(count)])
    want = [cr.g1_to_affine(cg.msm_g1(h, scs[i], offset=off)) for i in range(count)]
    d_out = cg.DevBuf(96 * count)
    for rank in range(k):
        lo, ln = _split(n, k, rank)
        cg.comm_loopback_set_rank(rank)
        assert cg.comm_info() == (rank, k)
        # this rank's slice of every MSM, as consecutive arrays of ln scalars (Montgomery form)
        d = cg.DevBuf.from_numpy(cr.vec_to_mont(1, np.ascontiguousarray(scs[:, lo:lo + ln]).reshape(-1, 4)))
        cg.msm_g1_sharded_dev(h, d, ln, count=count, montgomery=True, offset=off + lo, d_out=d_out)
        d.free()
    out = d_out.to_numpy().reshape(count, 12)
    for i in range(count):
        assert np.array_equal(cr.g1_to_affine(out[i]), want[i]), i
    # an empty rank (more ranks than points) contributes infinity
    tiny = cr.random_field(77, 1, k - 1, False)
    d_out1 = cg.DevBuf(96)
    for rank in range(k):
        lo, ln = _split(k - 1, k, rank)
        cg.comm_loopback_set_rank(rank)
        d = cg.DevBuf.from_numpy(np.ascontiguousarray(tiny[lo:lo + ln]) if ln else np.zeros((1, 4), np.uint64))
        cg.msm_g1_sharded_dev(h, d, ln, offset=lo, d_out=d_out1)
        d.free()
    assert np.array_equal(cr.g1_to_affine(d_out1.to_numpy()), cr.g1_to_affine(cg.msm_g1(h, tiny)))
    cg.srs_free(h)


def test_sum_over_more_ranks_than_lanes(cg, tau):
    """g1_sum_ranks with 130 ranks: the wavefront's lanes stride over the ranks (two and a bit rounds), two MSMs per
    exchange, most ranks holding two or three points"""
    k, n, count = 130, 301, 2
    cg.comm_init_loopback(k)
    try:
        h = cg.srs_generate(tau, n)
        scs = np.stack([cr.random_field(90 + i, 1, n, False) for i in range(count)])
        d_out = cg.DevBuf(96 * count)
        for rank in range(k):
            lo, ln = _split(n, k, rank)
            cg.comm_loopback_set_rank(rank)
            d = cg.DevBuf.from_numpy(np.ascontiguousarray(scs[:, lo:lo + ln]).reshape(-1, 4))
            cg.msm_g1_sharded_dev(h, d, ln, count=count, offset=lo, d_out=d_out)
            d.free()
        out = d_out.to_numpy().reshape(count, 12)
        for i in range(count):
            assert np.array_equal(cr.g1_to_affine(out[i]), cr.g1_to_affine(cg.msm_g1(h, scs[i]))), i
        cg.srs_free(h)
    finally:
        cg.comm_destroy()


def test_a_failed_rank_fails_every_rank(loop, tau):
    """comm.hip: the status word behind every rank's partials.  Rank 1 fails locally (its range lies beyond the SRS);
    it still deposits its payload, and the rank that completes the exchange reports the peer's failure."""
    cg, k = loop
    n = 1000
    h = cg.srs_generate(tau, n)
    sc = cr.random_field(3, 1, n, False)
    d_out = cg.DevBuf(96)
    codes = []
    for rank in range(k):
        lo, ln = _split(n, k, rank)
        cg.comm_loopback_set_rank(rank)
        d = cg.DevBuf.from_numpy(np.ascontiguousarray(sc[lo:lo + ln]))
        try:
            cg.msm_g1_sharded_dev(h, d, ln, offset=lo + (5 * n if rank == 1 else 0), d_out=d_out)
            codes.append(0)
        except cg.CapGpuError as e:
            codes.append(e.code)
            if rank == k - 1 and k > 2:
                assert "rank 1" in str(e)
        d.free()
    assert codes[1] == -1                                            # the failing rank sees its own error
    assert codes[-1] != 0                                            # and the exchange did not pass silently
    if k > 2:
        assert codes[-1] == -9 and codes[0] == 0
    cg.srs_free(h)


def test_prover_mode_a_in_a_loopback_world(loop, tau):
    """capgpu_plonk_shard_msm with k ranks: every commitment MSM of preprocess and of the five rounds is cut into k
    point ranges (rank > 0 range arithmetic, count = 5P partials per exchange) and reassembled; same bytes as plain."""
    cg, k = loop
    log_n = 12 if k == 8 else 9
    sc = bu.synthetic_circuit(log_n, 4, seed=12)
    h = cg.srs_generate(tau, sc.n + 3)
    pkh, vk = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    ws, ps, bls = [], [], []
    for p in range(3):
        w, pubs = sc.witness(60 + p)
        ws.append(sc.wires_mont(w)); ps.append(bu.to_mont_array(pubs)); bls.append(bu.to_mont_array(bu.blinders(70 + p)))
    plain = cg.plonk_prove_batch(pkh, np.stack(ws), np.stack(ps), np.stack(bls), b"m", 3)
    cg.plonk_shard_msm(True)
    pk2, vk2 = cg.plonk_preprocess(h, sc.n, 4, sc.selectors_mont(), sc.sigma_mont())
    assert bytes(vk2) == bytes(vk)
    sharded = cg.plonk_prove_batch(pk2, np.stack(ws), np.stack(ps), np.stack(bls), b"m", 3)
    cg.plonk_shard_msm(False)
    assert [bytes(p) for p in plain] == [bytes(p) for p in sharded]
    for key in (pkh, pk2):
        cg.plonk_free_key(key)
    cg.srs_free(h)


def test_comm_init_gives_up_on_a_missing_rank(cg):
    """A world of two of which only this rank shows up: capgpu_comm_init returns CAPGPU_ERR_COMM after the deadline
    instead of blocking for ever, and the library keeps working."""
    import os
    import time
    os.environ["CAPGPU_COMM_TIMEOUT_MS"] = "1500"
    try:
        t0 = time.time()
        with pytest.raises(cg.CapGpuError) as e:
            cg.comm_init(0, 2, cg.comm_unique_id())
        took = time.time() - t0
    finally:
        del os.environ["CAPGPU_COMM_TIMEOUT_MS"]
    assert e.value.code in (-9, -3), e.value                         # deadline (or RCCL's own bootstrap failure)
    assert took < 60
    assert cg.comm_info() == (0, 0)
    x = cr.random_field(6, 1, 1 << 8, True)
    assert np.array_equal(cg.ntt_fr(x, 8).reshape(-1), cr.ntt_fr(x, 8, False, False).reshape(-1))
