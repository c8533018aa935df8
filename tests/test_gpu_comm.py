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
