"""Host-side workload synthesis (cap_amd/bench_utils.py) produces satisfiable TurboPlonk instances with the
note shapes the reference pins (src/utils/mod.rs:136-193).  (`-m "not gpu"`)"""
import numpy as np
import pytest

from cap_amd import bench_utils as bu
from cap_amd import parallel as par
from oracle import bn254 as bn
from oracle import plonk as pl


def test_note_shapes_match_reference_pins():
    # Transfer(2,2,depth 10) -> domain 32768; Mint(26) / Freeze(2,5) -> 16384 (src/utils/mod.rs:149-177)
    assert bu.NOTE_SHAPES["transfer_2x2"] == (15, 27)
    assert bu.NOTE_SHAPES["mint"][0] == 14 and bu.NOTE_SHAPES["freeze_2"][0] == 14
    assert bu.K == pl.K and bu.R == bn.R


@pytest.mark.parametrize("log_n,nin", [(3, 1), (5, 3), (7, 0), (9, 27)])
def test_synthetic_circuit_is_satisfiable(log_n, nin):
    sc = bu.synthetic_circuit(log_n, nin, seed=log_n)
    for wseed in (1, 2):
        w, pubs = sc.witness(wseed)
        assert len(pubs) == nin
        pl.check_circuit_satisfiability(pl.Circuit(n=sc.n, num_inputs=nin, selectors=sc.selectors, sigma=sc.sigma,
                                                   wires=w, pub_inputs=pubs))
    # every selector column is exercised and the permutation is non-trivial
    assert all(any(col) for col in sc.selectors)
    om = bn.root_of_unity(log_n)
    ident = sum(1 for i in range(5) for j in range(sc.n) if sc.sigma[i][j] == bu.K[i] * pow(om, j, bn.R) % bn.R)
    assert ident < 5 * sc.n // 2


def test_mont_array_roundtrip():
    vals = [0, 1, bn.R - 1, 12345678901234567890123]
    a = bu.to_mont_array(vals)
    assert a.shape == (4, 4) and a.dtype == np.uint64
    assert bu.from_mont_array(a) == vals
    assert [bn.from_mont(int(r[0]) | int(r[1]) << 64 | int(r[2]) << 128 | int(r[3]) << 192, bn.R) for r in a] == vals


def test_shard_ranges_partition():
    for n in (0, 1, 7, 1 << 17, (1 << 24) + 3):
        for world in (1, 2, 3, 8):
            spans = [par.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    assert sorted(sum((par.shard_proofs(64, r, 8) for r in range(8)), [])) == list(range(64))


def test_param_sizes_pinned_by_the_reference():
    """src/utils/mod.rs:136-193 (test_compute_srs_size, BN254 values) and the shapes the benchmark synthesises."""
    import pytest
    from cap_amd import proof
    assert proof.compute_universal_param_size("Transfer", 3, 5, 26) == 65538
    assert proof.compute_universal_param_size("Transfer", 2, 2, 10) == 32770
    assert proof.compute_universal_param_size("Mint", 0, 0, 26) == 16386
    assert proof.compute_universal_param_size("Freeze", 2, 0, 5) == 16386
    assert proof.compute_universal_param_size("Freeze", 5, 0, 26) == 65538
    # the arithmetic itself: domain = next power of two of the gate count, + 2 for the blinding
    assert proof.compute_universal_param_size("Transfer", 2, 6, 10) == 32770          # 30 740 gates (transfer.rs:602)
    assert proof.universal_param_size_for_gates(30740) == 32770 and proof.eval_domain_size(32768) == 32768
    assert proof.eval_domain_size(32769) == 65536 and proof.eval_domain_size(1) == 1
    with pytest.raises(proof.TxnApiError):
        proof.compute_universal_param_size("Transfer", 7, 7, 3)
    # the synthetic circuits of the benchmark have exactly these domains: size - 2 = 2^log_n
    assert (1 << bu.NOTE_SHAPES["transfer_2x2"][0]) + 2 == proof.compute_universal_param_size("transfer", 2, 2, 10)
    assert (1 << bu.NOTE_SHAPES["transfer_2x2_d26"][0]) + 2 == proof.compute_universal_param_size("transfer", 3, 5, 26)
    assert (1 << bu.NOTE_SHAPES["mint"][0]) + 2 == proof.compute_universal_param_size("mint", 0, 0, 26)
    assert (1 << bu.NOTE_SHAPES["freeze_2"][0]) + 2 == proof.compute_universal_param_size("freeze", 2, 0, 5)


def test_weighted_scalar_sums_exact():
    """the full-size MSM known answer (sum k_i, sum (lo + i) k_i) against Python integers, across a block boundary"""
    sc = bu.random_canonical_scalars(3, (1 << 18) + 77)
    ks = [int(w[0]) | int(w[1]) << 64 | int(w[2]) << 128 | int(w[3]) << 192 for w in sc]
    assert all(k < bu.R for k in ks[:1000])
    s0, s1 = bu.weighted_scalar_sums(sc, 12345)
    assert s0 == sum(ks)
    assert s1 == sum((12345 + i) * k for i, k in enumerate(ks))


def test_transfer_public_input_counts():
    """27 is pinned by the reference for 2-in/2-out (src/proof/transfer.rs:443-458 + the viewing memo layout)."""
    assert bu.transfer_num_public_inputs(2, 2) == 27
    assert bu.NOTE_SHAPES["transfer_2x2"] == (15, 27) and bu.NOTE_SHAPES["transfer_2x3"] == (15, 32)


@pytest.mark.parametrize("log_n,nin", [(5, 3), (9, 27)])
def test_c_witness_generator_is_the_python_one(log_n, nin):
    """cap_amd/csrc/witgen.c (what bench.py uses for its 256 distinct witnesses) against SyntheticCircuit.witness()"""
    sc = bu.synthetic_circuit(log_n, nin, seed=log_n)
    seeds = [1, 2, 77, 1 << 40]
    wm, pm = sc.witnesses_mont(seeds, threads=3, verify=True)
    assert wm.shape == (4, 5, sc.n, 4) and pm.shape == (4, nin, 4)
    for k, sd in enumerate(seeds):
        w, pubs = sc.witness(sd)
        assert np.array_equal(wm[k], sc.wires_mont(w)) and np.array_equal(pm[k], bu.to_mont_array(pubs))
    assert not np.array_equal(wm[0], wm[1])


def test_cap_like_circuit_is_satisfiable_and_cap_shaped():
    """The CAP-shaped model circuit (round-4 VERDICT item 1b): satisfiable, lands on the reference's pinned domain for
    2-in/2-out at depth 10 (src/utils/mod.rs:149-153), and its wire columns hold what a note's do - mostly zeros,
    booleans and full-width values."""
    sc = bu.cap_like_circuit("transfer_2x2", seed=7)
    assert sc.n == 1 << 15 and sc.num_inputs == 27
    assert sc.n // 2 < sc.gate_rows <= sc.n                       # the gate count pads to 2^15, not 2^14
    wm, pm = sc.witnesses_mont([5, 6], verify=True)               # verify: every constraint row is checked in C
    w, pubs = sc.witness(5)                                       # ... and the Python definition agrees
    assert np.array_equal(wm[0], sc.wires_mont(w)) and np.array_equal(pm[0], bu.to_mont_array(pubs))
    pl.check_circuit_satisfiability(pl.Circuit(n=sc.n, num_inputs=27, selectors=sc.selectors, sigma=sc.sigma,
                                               wires=w, pub_inputs=pubs))
    cls = bu.value_classes(wm[0])
    assert cls["cells"] == 5 * sc.n and abs(sum(cls[k] for k in ("zero", "one", "below_2^64", "full_width")) - 1) < 1e-9
    assert 0.3 < cls["zero"] < 0.7 and 0.02 < cls["one"] < 0.15 and 0.25 < cls["full_width"] < 0.6
    uni = bu.synthetic_circuit(9, 27, seed=9)
    assert bu.value_classes(uni.witnesses_mont([1])[0])["full_width"] > 0.85
    assert sc.gadget_rows["padding"] == sc.n - sc.gate_rows and sc.gadget_rows["merkle path"] == 2 * 10 * 156


def test_skewed_synthetic_circuits_are_satisfiable():
    """synthetic_circuit(skew=...): most free variables booleans / 64-bit values, more padding - the witness
    distributions the fuzzers of the commitments-from-evaluations draw (tools/gpu_fuzz_prover.py)"""
    for log_n, fill, skew in ((6, 0.3, 0.97), (8, 0.6, 0.5)):
        sc = bu.synthetic_circuit(log_n, 3, seed=5, fill=fill, skew=skew)
        w, pubs = sc.witness(9)
        pl.check_circuit_satisfiability(pl.Circuit(n=sc.n, num_inputs=3, selectors=sc.selectors, sigma=sc.sigma,
                                                   wires=w, pub_inputs=pubs))
        wm, _ = sc.witnesses_mont([9], verify=True)
        assert np.array_equal(wm[0], sc.wires_mont(w))
        cls = bu.value_classes(wm[0])
        assert cls["zero"] > 0.3 and cls["full_width"] < 0.6
    # the counter of an MSM's work agrees with a direct count of signed base-2^c digits
    vals = [0, 1, 2, (1 << 15) - 1, 1 << 14, (1 << 14) + 1, bu.R - 1, 12345678901234567890123456789]
    ent, bkt = bu.msm_work(bu.to_mont_array(vals), 15)

    def digits(k, c=15):
        out, carry = [], 0
        for wdw in range((254 + c - 1) // c):
            d = ((k >> (wdw * c)) & ((1 << c) - 1)) + carry
            carry = 0
            if d > (1 << (c - 1)):
                d, carry = (1 << c) - d, 1
            out.append(d)
        return out
    ds = [d for k in vals for d in digits(k) if d]
    assert ent == len(ds) and bkt == len(set(ds))
