"""Footprint control and diagnostics of the library (round-5 VERDICT item 6, ADVICE): capgpu_trim gives the scratch of idle
contexts back and the next proofs are the same bytes; capgpu_set_memory_limit turns a batch that does not fit into
CAPGPU_ERR_OOM without harming the next smaller one; the HIP-event timer has one owner per context; the staging slots of
coalesced callers are scratch; the phase trace records what it says."""
import ctypes
import threading

import numpy as np
import pytest

from cap_amd import bench_utils as bu

pytestmark = pytest.mark.gpu


def pubs_arr(pubs):
    return bu.to_mont_array(pubs) if pubs else np.zeros((0, 4), np.uint64)


@pytest.fixture(scope="module")
def setup(cg, tau):
    log_n, nin, P = 12, 5, 24
    sc = bu.synthetic_circuit(log_n, nin, seed=91)
    h = cg.srs_generate(tau, sc.n + 3)
    pk, _vk = cg.plonk_preprocess(h, sc.n, nin, sc.selectors_mont(), sc.sigma_mont())
    wires, pubs = sc.witnesses_mont([40 + i for i in range(P)])
    blind = np.stack([bu.to_mont_array(bu.blinders(300 + i)) for i in range(P)])
    return {"pk": pk, "wires": wires, "pubs": pubs, "blind": blind, "P": P, "n": sc.n, "nin": nin, "srs": h}


def prove(cg, s, count):
    return [bytes(p) for p in cg.plonk_prove_batch(s["pk"], s["wires"][:count], s["pubs"][:count], s["blind"][:count], b"mem", count)]


def test_trim_gives_scratch_back_and_proofs_stay_the_same(cg, setup):
    s = setup
    cg.set_device(-1)
    want = prove(cg, s, s["P"])
    scratch0, _ = cg.scratch_info()
    free0, total = cg.mem_info()
    assert scratch0 > 0
    released, busy = cg.trim()
    scratch1, _ = cg.scratch_info()
    free1, _ = cg.mem_info()
    assert busy == 0 and released >= scratch0 and scratch1 == 0
    assert free1 - free0 >= released * 0.9          # the device really has it back (the allocator may keep a little)
    assert cg.trim() == (0, 0)                      # idempotent
    assert prove(cg, s, s["P"]) == want             # the next call allocates again: same bytes
    assert cg.scratch_info()[0] > 0


def test_memory_limit_refuses_what_does_not_fit_and_recovers(cg, setup):
    s = setup
    cg.set_device(-1)
    want4 = prove(cg, s, 4)
    cg.trim()
    prove(cg, s, 4)
    small, _ = cg.scratch_info()
    try:
        cg.set_memory_limit(int(small * 1.5))       # room for batches of 4 (on two contexts' worth at most), not for 24
        assert cg.scratch_info()[1] == int(small * 1.5)
        with pytest.raises(cg.CapGpuError) as e:
            prove(cg, s, s["P"])
        assert e.value.code == -5 and "capgpu_set_memory_limit" in str(e.value)
        assert cg.scratch_info()[0] <= int(small * 1.5)
        assert prove(cg, s, 4) == want4              # the smaller batch still fits (idle contexts were trimmed for it)
    finally:
        cg.set_memory_limit(0)
    assert len(prove(cg, s, s["P"])) == s["P"]
    # a cap below what is already held trims the idle contexts at once
    held, _ = cg.scratch_info()
    try:
        cg.set_memory_limit(1 << 20)
        assert cg.scratch_info()[0] < held
    finally:
        cg.set_memory_limit(0)


def test_timer_has_one_owner_per_context(cg, setup):
    cg.set_device(0)
    cg.timer_begin()
    cg.timer_begin()                                  # the owner may restart its measurement
    res = {}

    def other():
        cg.set_device(0)
        try:
            cg.timer_begin()
            res["begin"] = "ok"
        except cg.CapGpuError as e:
            res["begin"] = e.code
        try:
            cg.timer_end()
            res["end"] = "ok"
        except cg.CapGpuError as e:
            res["end"] = e.code

    t = threading.Thread(target=other)
    t.start()
    t.join()
    assert res == {"begin": -1, "end": -1}            # another thread is refused while it is open
    d = cg.DevBuf.from_numpy(bu.random_canonical_scalars(3, 1 << 12))
    cg.ntt_fr_dev(d, 12)
    assert cg.timer_end() > 0
    with pytest.raises(cg.CapGpuError):
        cg.timer_end()                                # closed: nothing to end
    t = threading.Thread(target=other)                # ... and now the other thread can measure
    t.start()
    t.join()
    assert res == {"begin": "ok", "end": "ok"}
    d.free()


def test_coalesced_callers_stage_their_witnesses_and_the_trace_says_so(cg, setup, tmp_path):
    s = setup
    cg.set_device(-1)
    want = prove(cg, s, s["P"])
    L = cg.load()
    u64p = ctypes.POINTER(ctypes.c_uint64)
    msg = (ctypes.c_uint8 * 3).from_buffer_copy(b"mem")
    T = s["P"]
    proofs = [cg.Proof() for _ in range(T)]
    rcs = [None] * T
    import os
    os.environ["CAPGPU_COALESCE_PRESTAGE"] = "1"      # (off by default: same throughput, lower caller latency, 5 n 32 B per caller)
    cg.plonk_set_coalescing(2000, 256)
    cg.trace_enable(True)
    try:
        def call(i):
            rcs[i] = L.capgpu_plonk_prove_ex(ctypes.c_uint64(s["pk"]), s["wires"][i].ctypes.data_as(u64p),
                                             s["pubs"][i].ctypes.data_as(u64p), ctypes.c_size_t(s["nin"]), msg,
                                             ctypes.c_size_t(3), s["blind"][i].ctypes.data_as(u64p), ctypes.c_int(0),
                                             ctypes.byref(proofs[i]))
        for _ in range(2):
            th = [threading.Thread(target=call, args=(i,)) for i in range(T)]
            for t in th:
                t.start()
            for t in th:
                t.join()
            assert rcs == [0] * T
            assert [bytes(p) for p in proofs] == want      # staged or not, gathered in any order: the same proofs
    finally:
        cg.trace_enable(False)
        cg.plonk_set_coalescing(0)
        os.environ.pop("CAPGPU_COALESCE_PRESTAGE", None)
    path = str(tmp_path / "trace.txt")
    n_ev = cg.trace_dump(path)
    tags = [ln.split()[2] for ln in open(path)]
    assert n_ev == len(tags) > 0
    assert tags.count("co_submit") == 2 * T and tags.count("co_return") == 2 * T
    assert tags.count("pb_begin") == tags.count("pb_end") >= 2
    staged = [int(ln.split()[3]) for ln in open(path) if ln.split()[2] == "co_prestaged"]
    assert len(staged) == 2 * T and all(staged), "one bound device: every caller gets a staging slot"
    assert "co_gather_staged" in tags
    # the slots are scratch: trim frees them
    held, _ = cg.scratch_info()
    cg.trim()
    assert cg.scratch_info()[0] == 0 and held >= T * 5 * s["n"] * 32


def test_host_batch_parts_in_any_split_make_the_same_proofs(cg, setup):
    """the dealer's uneven cut and the copy turns of a host batch's parts (plonk.hip: deal, H2dTurn) change the schedule,
    never the proofs: a batch of 64 host witnesses against the same witnesses proved one context at a time"""
    sc = bu.synthetic_circuit(10, 3, seed=17)
    from oracle import bn254 as bn
    tau = bn.SplitMix64(0xCA9).field(bn.R)
    h = cg.srs_generate(tau, sc.n + 3)
    pk, _ = cg.plonk_preprocess(h, sc.n, 3, sc.selectors_mont(), sc.sigma_mont())
    P = 64
    wires, pubs = sc.witnesses_mont([7 + i for i in range(P)])
    blind = np.stack([bu.to_mont_array(bu.blinders(900 + i)) for i in range(P)])
    cg.set_device(-1)
    dealt = [bytes(p) for p in cg.plonk_prove_batch(pk, wires, pubs, blind, b"deal", P)]
    cg.set_device(0)
    whole = [bytes(p) for p in cg.plonk_prove_batch(pk, wires, pubs, blind, b"deal", P)]
    assert dealt == whole
    cg.plonk_free_key(pk)
    cg.srs_free(h)
