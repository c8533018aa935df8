"""Small batches replay their kernel schedule as hipGraphs (cap_amd/csrc/plonk.hip: ProveGraphSet): the launches between
two transcript steps are captured on the second call with the same key, batch size and buffers, and replayed from then
on.  The proofs must be the bytes of the direct-launch path - whatever the witness in the (same) buffers, after an
unsatisfied witness, for host- and device-resident wires, batches and mixed keys - and those are pinned to the C oracle."""
import os

import numpy as np
import pytest

from cap_amd import bench_utils as bu
from oracle import capref as cr
from tests import helpers as H

pytestmark = pytest.mark.gpu


def pubs_arr(pubs):
    return bu.to_mont_array(pubs) if pubs else np.zeros((0, 4), np.uint64)


def instance(sc, seed):
    w, pubs = sc.witness(seed)
    return sc.wires_mont(w), pubs_arr(pubs), bu.to_mont_array(bu.blinders(seed + 500))


def graphs_in_this_process(cg):
    """hipGraph replay needs a HIP runtime at least as new as the library's build (7.2): see tests/conftest.py"""
    return cg.runtime_info()[0] >= 70200000 or os.environ.get("CAPGPU_GRAPH_FORCE") == "1"


def test_graph_replay_in_a_process_on_the_build_runtime(cg):
    """This file once more in a child process that loads the library BEFORE torch (CAPGPU_TEST_LIBRARY_FIRST=1): there the
    library runs on /opt/rocm's HIP runtime, captures and replays, and every test of this file must pass with the counts
    checked.  (Nothing to do when this process is already such a process.)"""
    import subprocess
    import sys
    if graphs_in_this_process(cg):
        return
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, CAPGPU_TEST_LIBRARY_FIRST="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-m", "gpu", "-x", "--timeout=300",
                        "-p", "no:cacheprovider"], cwd=root, env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-1500:] + r.stderr[-500:]
    assert " passed" in r.stdout and "skipped" not in r.stdout.splitlines()[-1], r.stdout[-300:]


def direct(cg, fn):
    """the same call with graphs switched off (CAPGPU_GRAPH_MAX_BATCH is read per call)"""
    old = os.environ.get("CAPGPU_GRAPH_MAX_BATCH")
    os.environ["CAPGPU_GRAPH_MAX_BATCH"] = "0"
    try:
        return fn()
    finally:
        if old is None:
            del os.environ["CAPGPU_GRAPH_MAX_BATCH"]
        else:
            os.environ["CAPGPU_GRAPH_MAX_BATCH"] = old


@pytest.mark.parametrize("log_n,nin", [(6, 2), (11, 5), (15, 27)])
def test_replayed_proofs_are_the_direct_proofs(cg, tau, log_n, nin):
    sc = bu.synthetic_circuit(log_n, nin, seed=60 + log_n)
    n = sc.n
    h = cg.srs_generate(tau, n + 3)
    pk, vk = cg.plonk_preprocess(h, n, nin, sc.selectors_mont(), sc.sigma_mont())
    key = cr.PlonkKey(cg.srs_download(h, 0, n + 3), n, nin, sc.selectors_mont(), sc.sigma_mont())
    insts = [instance(sc, 10 + i) for i in range(3)]
    d = cg.DevBuf.from_numpy(insts[0][0][None])
    cap0, rep0 = cg.plonk_graph_stats()
    got = []
    for rnd in range(6):                       # same buffer, the witness in it changes from call to call
        wm, pm, bm = insts[rnd % 3]
        d.upload(wm)
        got.append(bytes(cg.plonk_prove_batch_dev(pk, d, pm[None], bm[None], b"g", 1)[0]))
    cap1, rep1 = cg.plonk_graph_stats()
    if not graphs_in_this_process(cg):
        # torch was imported before the library (tests/conftest.py): the process runs on torch's older HIP runtime, where
        # graph replay is off (plonk.hip: graph_runtime_ok) and every call launches directly - the parity assertions below
        # still hold; the capture / replay counts are checked by test_graph_replay_in_a_process_on_the_build_runtime
        assert (cap1 - cap0, rep1 - rep0) == (0, 0)
    else:
        # (eight segments; six when round 1 runs its transforms on the side stream and segment 1 - the coset transforms
        # behind the commitments - has nothing left to do: CAPGPU_R1_OVERLAP_MAX)
        assert cap1 - cap0 in (6, 7, 8), "the second (or third) call captures the segments once"
        assert rep1 - rep0 >= 6 * 2, "later calls replay them"
    for i in range(3):
        assert got[i] == got[i + 3]
        wm, pm, bm = insts[i]
        assert got[i] == bytes(direct(cg, lambda: cg.plonk_prove_batch_dev(pk, d_of(cg, wm), pm[None], bm[None], b"g", 1))[0])
    rc, comms, evals = key.prove(*insts[0], b"g")
    pr0 = cg.plonk_prove_batch_dev(pk, d_of(cg, insts[0][0]), insts[0][1][None], insts[0][2][None], b"g", 1)[0]
    assert rc == 0 and H.proof_points(pr0) == H.cref_proof_points(comms, evals) and bytes(pr0) == got[0]
    # an unsatisfied witness under replay fails with the prover's error, and the next good call is good
    bad = insts[1][0].copy()
    bad[4, n // 2, 0] ^= 1
    d.upload(bad)
    with pytest.raises(cg.CapGpuError) as e:
        cg.plonk_prove_batch_dev(pk, d, insts[1][1][None], insts[1][2][None], b"g", 1)
    assert e.value.code == -7
    wm, pm, bm = insts[2]
    d.upload(wm)
    assert bytes(cg.plonk_prove_batch_dev(pk, d, pm[None], bm[None], b"g", 1)[0]) == got[2]
    # host-resident wires (capgpu_plonk_prove: the staging buffer is the library's), another message per call
    for rnd in range(4):
        wm, pm, bm = insts[rnd % 3]
        a = bytes(cg.plonk_prove(pk, wm, pm, bm, b"m%d" % rnd))
        assert a == bytes(direct(cg, lambda: cg.plonk_prove(pk, wm, pm, bm, b"m%d" % rnd)))
    # coefficient-form input has its own signature
    wc = np.stack([cr.ntt_fr(c, log_n, True, False).reshape(-1, 4) for c in insts[0][0]])
    for _ in range(3):
        assert bytes(cg.plonk_prove(pk, wc, insts[0][1], insts[0][2], b"g", input_form="coeffs")) == got[0]
    d.free()
    for x in _keep:
        x.free()
    _keep.clear()
    cg.plonk_free_key(pk)
    cg.srs_free(h)


_keep = []


def d_of(cg, wm):
    b = cg.DevBuf.from_numpy(np.ascontiguousarray(wm)[None])
    _keep.append(b)
    return b


def test_small_batches_and_mixed_keys_under_replay(cg, tau):
    log_n = 9
    n = 1 << log_n
    srs = cg.srs_generate(tau, n + 3)
    circuits = [bu.synthetic_circuit(log_n, ni, seed=seed) for ni, seed in ((3, 31), (9, 32))]
    keys = [cg.plonk_preprocess(srs, n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())[0] for sc in circuits]
    order = [0, 1, 1, 0, 1]
    wires, rows, blinds, msgs = [], [], [], []
    for i, k in enumerate(order):
        wm, pm, bm = instance(circuits[k], 800 + i)
        row = np.zeros((9, 4), np.uint64)
        row[:len(pm)] = pm
        wires.append(wm); rows.append(row); blinds.append(bm); msgs.append(b"n%d" % i)
    handles = [keys[k] for k in order]
    d = cg.DevBuf.from_numpy(np.stack(wires))
    want = [bytes(p) for p in direct(cg, lambda: cg.plonk_prove_multi(handles, d, np.stack(rows), np.stack(blinds), msgs))]
    for _ in range(4):
        assert [bytes(p) for p in cg.plonk_prove_multi(handles, d, np.stack(rows), np.stack(blinds), msgs)] == want
    # a different mix with the same first key reuses the captured segments: per-proof key data travels through device
    # memory, not through kernel arguments
    order2 = [0, 0, 1, 1, 0]
    w2, r2, b2 = [], [], []
    for i, k in enumerate(order2):
        wm, pm, bm = instance(circuits[k], 900 + i)
        row = np.zeros((9, 4), np.uint64)
        row[:len(pm)] = pm
        w2.append(wm); r2.append(row); b2.append(bm)
    h2 = [keys[k] for k in order2]
    d.upload(np.stack(w2))
    a = [bytes(p) for p in cg.plonk_prove_multi(h2, d, np.stack(r2), np.stack(b2), msgs)]
    b = [bytes(p) for p in direct(cg, lambda: cg.plonk_prove_multi(h2, d, np.stack(r2), np.stack(b2), msgs))]
    assert a == b and a != want
    # a freed key's graphs are never replayed for a new key (unique key ids), even at the same addresses
    pm0 = rows[0][:3]
    single = bytes(cg.plonk_prove(keys[0], wires[0], pm0, blinds[0], b"x"))
    for _ in range(3):
        assert bytes(cg.plonk_prove(keys[0], wires[0], pm0, blinds[0], b"x")) == single
    cg.plonk_free_key(keys[0])
    k0 = cg.plonk_preprocess(srs, n, circuits[1].num_inputs, circuits[1].selectors_mont(), circuits[1].sigma_mont())[0]
    wm, pm, bm = instance(circuits[1], 77)
    x = bytes(cg.plonk_prove(k0, wm, pm, bm, b"x"))
    assert x == bytes(direct(cg, lambda: cg.plonk_prove(k0, wm, pm, bm, b"x")))
    d.free()
    for k in (keys[1], k0):
        cg.plonk_free_key(k)
    cg.srs_free(srs)
