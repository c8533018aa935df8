"""On-disk parameter formats: oracle/params.py against the committed fixture, the host-only
verifying-key codec of the C ABI against the oracle (no GPU), and - on the GPU - bulk G1 (de)compression, the
UniversalSrs blob and the proving-key blob against the oracle's bytes, bit for bit.

The reference's own tests for this module are store -> load round trips that assert the note-shape fields
(src/parameters.rs:603-688); `test_store_and_load_*` below are their counterparts."""
import ctypes

import numpy as np
import pytest

from cap_amd import bench_utils as bu
from cap_amd import lib as cglib
from cap_amd import parameters as prm
from cap_amd import proof as capproof
from oracle import bn254 as bn
from oracle import capref as cr
from oracle import pairing as pr2
from oracle import params as pm
from oracle import plonk as pl
from tests import helpers as H

G = H.load_golden("params.json")


def g2_words(q) -> np.ndarray:
    """oracle G2 point -> 16 Montgomery words (x.c0, x.c1, y.c0, y.c1)."""
    vals = [q[0][0], q[0][1], q[1][0], q[1][1]]
    return cr.ints_to_array([bn.to_mont(v, bn.P) for v in vals]).reshape(16)


def g1_words(p) -> np.ndarray:
    if p is bn.INF:
        return np.zeros(8, dtype=np.uint64)
    return cr.ints_to_array([bn.to_mont(p[0], bn.P), bn.to_mont(p[1], bn.P)]).reshape(8)


def words_g1(w):
    return cr.affine_to_ints(np.asarray(w, dtype=np.uint64).reshape(8))


def jac_ints(jac):
    return cr.affine_to_ints(cr.g1_to_affine(jac))


def oracle_key():
    tau = bn.SplitMix64(G["tau_seed"]).field(bn.R)
    sc = bu.synthetic_circuit(G["log_n"], G["num_inputs"], seed=G["circuit_seed"])
    return tau, sc


# ---- CPU: the oracle against the fixture, and the host-only part of the ABI -------------------------------------
def test_oracle_g1_g2_codec_matches_fixture():
    for hexed, point in G["g1_ok"]:
        p = H.unhex_pt(point)
        assert bn.g1_serialize_compressed(p).hex() == hexed
        assert pm.g1_deserialize_compressed(bytes.fromhex(hexed)) == p
    for name, hexed in G["g1_bad"].items():
        with pytest.raises(pm.SerializationError):
            pm.g1_deserialize_compressed(bytes.fromhex(hexed))
    for hexed, coords in G["g2"]:
        q = ((int(coords[0][0], 16), int(coords[0][1], 16)), (int(coords[1][0], 16), int(coords[1][1], 16)))
        assert pm.g2_serialize_compressed(q).hex() == hexed
        assert pm.g2_deserialize_compressed(bytes.fromhex(hexed)) == q
    # a point of the twist outside the prime-order subgroup must be rejected (ark-ec checks the subgroup)
    for x0 in range(1, 200):
        rhs = pr2.f2_add(pr2.f2_mul(pr2.f2_mul((x0, 0), (x0, 0)), (x0, 0)), pm._g2_b())
        y = pm.f2_sqrt(rhs)
        if y is not None and not pm.g2_in_subgroup(((x0, 0), y)):
            with pytest.raises(pm.SerializationError):
                pm.g2_deserialize_compressed(pm.g2_serialize_compressed(((x0, 0), y)))
            break
    else:
        pytest.fail("no small-x twist point outside the subgroup found")


def test_oracle_blobs_match_fixture_and_round_trip():
    tau, sc = oracle_key()
    srs = pm.deserialize_universal_params(bytes.fromhex(G["srs"]))
    assert srs["consumed"] == len(G["srs"]) // 2 and len(srs["powers_of_g"]) == sc.n + 3
    x = 1
    for p in srs["powers_of_g"][:4]:
        assert p == bn.g1_mul(bn.G1_GEN, x)
        x = x * tau % bn.R
    assert srs["h"] == pr2.G2_GEN and srs["beta_h"] == pr2.g2_mul(pr2.G2_GEN, tau)
    key = pm.deserialize_proving_key(bytes.fromhex(G["proving_key"]))
    assert key["consumed"] == len(G["proving_key"]) // 2
    assert key["vk"]["domain_size"] == sc.n and key["vk"]["num_inputs"] == G["num_inputs"] and key["vk"]["k"] == pl.K
    assert key["powers_of_g"] == srs["powers_of_g"]
    # the polynomials interpolate the circuit's columns
    ln = G["log_n"]
    for col, poly in zip(sc.selectors, key["selectors"]):
        assert bn.ntt(poly + [0] * (sc.n - len(poly)), ln) == [v % bn.R for v in col]
    for col, poly in zip(sc.sigma, key["sigmas"]):
        assert bn.ntt(poly + [0] * (sc.n - len(poly)), ln) == [v % bn.R for v in col]
    # and the commitments inside the vk are the commitments of those polynomials
    assert key["vk"]["selector_comms"][3] == pl.commit(key["selectors"][3], tau)
    assert key["vk"]["sigma_comms"][4] == pl.commit(key["sigmas"][4], tau)
    # re-serialising gives the same bytes
    vkb = pm.serialize_verifying_key(sc.n, G["num_inputs"], key["vk"]["sigma_comms"], key["vk"]["selector_comms"],
                                     key["vk"]["k"], key["vk"]["g"], key["vk"]["gamma_g"], key["vk"]["h"],
                                     key["vk"]["beta_h"])
    assert vkb.hex() == G["vk"]
    assert pm.serialize_proving_key(key["sigmas"], key["selectors"], key["powers_of_g"], vkb).hex() == G["proving_key"]


def test_abi_vk_codec_matches_oracle_without_gpu():
    """capgpu_plonk_vk_{de,}serialize run on the host: exact bytes of the oracle, and every field recovered."""
    blob = bytes.fromhex(G["vk"])
    vk, g, gg, h, bh, used = cglib.plonk_vk_deserialize(blob + b"\x07trailer")
    assert used == len(blob)
    o = pm.read_verifying_key(pm.Reader(blob))
    assert vk.domain_size == o["domain_size"] and vk.num_inputs == o["num_inputs"]
    assert [words_g1(np.ctypeslib.as_array(vk.sigma_comms[i])) for i in range(5)] == o["sigma_comms"]
    assert [words_g1(np.ctypeslib.as_array(vk.selector_comms[i])) for i in range(13)] == o["selector_comms"]
    assert H.fr_to_ints(np.ctypeslib.as_array(vk.k).reshape(5, 4)) == o["k"]
    assert words_g1(g) == o["g"] and not gg.any()
    assert np.array_equal(h, g2_words(o["h"])) and np.array_equal(bh, g2_words(o["beta_h"]))
    assert cglib.plonk_vk_serialize(vk, g, h, bh) == blob
    # a hiding generator, when given, is written in place of infinity
    p7 = bn.g1_mul(bn.G1_GEN, 7)
    with_gamma = cglib.plonk_vk_serialize(vk, g, h, bh, gamma_g=g1_words(p7))
    assert pm.read_verifying_key(pm.Reader(with_gamma))["gamma_g"] == p7


@pytest.mark.parametrize("mutate", ["truncate", "sigma_count", "bad_point", "bad_scalar", "g2_flags", "merged", "g2_off_subgroup"])
def test_abi_vk_rejects_malformed_blobs(mutate):
    blob = bytearray(bytes.fromhex(G["vk"]))
    if mutate == "truncate":
        blob = blob[:-40]
    elif mutate == "sigma_count":
        blob[16] = 4
    elif mutate == "bad_point":
        blob[24:56] = bytes.fromhex(G["g1_bad"]["x_not_on_curve"])
    elif mutate == "bad_scalar":
        off = 16 + 8 + 5 * 32 + 8 + 13 * 32 + 8
        blob[off:off + 32] = bn.R.to_bytes(32, "little")
    elif mutate == "g2_flags":
        blob[-3] |= 0xC0
    elif mutate == "merged":
        blob[-2] = 1
    elif mutate == "g2_off_subgroup":
        for x0 in range(1, 200):
            rhs = pr2.f2_add(pr2.f2_mul(pr2.f2_mul((x0, 0), (x0, 0)), (x0, 0)), pm._g2_b())
            y = pm.f2_sqrt(rhs)
            if y is not None and not pm.g2_in_subgroup(((x0, 0), y)):
                blob[-130:-66] = pm.g2_serialize_compressed(((x0, 0), y))      # open_key.h
                break
    with pytest.raises(cglib.CapGpuError) as e:
        cglib.plonk_vk_deserialize(bytes(blob))
    assert e.value.code == cglib.CAPGPU_ERR_SERIALIZATION
    with pytest.raises(capproof.TxnApiError, match="DeserializationError"):
        prm.deserialize_verifying_key(bytes(blob))


def test_abi_vk_parser_survives_random_corruption():
    """Byte flips, truncations and length-prefix edits never crash the host-side parser: it returns a key or the
    serialization error (what ark-serialize's InvalidData becomes), nothing else."""
    blob = bytes.fromhex(G["vk"])
    rng = np.random.default_rng(2024)
    outcomes = {"ok": 0, "rejected": 0}
    for trial in range(120):
        b = bytearray(blob)
        kind = trial % 4
        if kind == 0:
            for pos in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
                b[pos] ^= 1 << int(rng.integers(0, 8))
        elif kind == 1:
            b = b[:int(rng.integers(0, len(b)))]
        elif kind == 2:
            off = [0, 8, 16, 16 + 8 + 5 * 32, 16 + 8 + 5 * 32 + 8 + 13 * 32][int(rng.integers(0, 5))]
            b[off:off + 8] = int(rng.integers(0, 1 << 62)).to_bytes(8, "little")
        else:
            b += bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))   # trailing bytes are fine
        try:
            vk, *_rest, used = cglib.plonk_vk_deserialize(bytes(b))
            assert used <= len(b)
            outcomes["ok"] += 1
        except cglib.CapGpuError as e:
            assert e.code == cglib.CAPGPU_ERR_SERIALIZATION
            outcomes["rejected"] += 1
    assert outcomes["rejected"] > 40 and outcomes["ok"] >= 30      # appended bytes / harmless flips parse, the rest not


def test_default_paths_follow_the_reference(monkeypatch, tmp_path):
    monkeypatch.setenv("CAP_UNIV_PARAM_DIR", str(tmp_path))
    d = tmp_path / "data"
    assert prm.default_path("universal_srs", "bin") == d / "universal_srs.bin"
    assert prm.default_transfer_proving_key_path(2, 5, 10) == d / "transfer_prover_2_input_5_output_10_depth.bin"
    assert prm.default_transfer_verifying_key_path(2, 5, 10) == d / "transfer_verifier_2_input_5_output_10_depth.bin"
    assert prm.default_mint_proving_key_path(10) == d / "mint_prover_1_input_2_output_10_depth.bin"
    assert prm.default_mint_verifying_key_path(10) == d / "mint_verifier_1_input_2_output_10_depth.bin"
    assert prm.default_freeze_proving_key_path(2, 10) == d / "freeze_prover_2_input_2_output_10_depth.bin"
    assert prm.default_freeze_verifying_key_path(2, 10) == d / "freeze_verifier_2_input_2_output_10_depth.bin"
    with pytest.raises(capproof.TxnApiError, match="IoError"):
        prm.load_bytes(d / "missing.bin")
    # verifying keys round-trip through files without a GPU, trailer included
    vk, used = prm.deserialize_verifying_key(bytes.fromhex(G["vk"]))
    d.mkdir()
    blob = bytes.fromhex(G["vk"])
    prm.store_bytes(blob + prm._transfer_trailer(2, 5, 10), prm.default_transfer_verifying_key_path(2, 5, 10))
    k = prm.load_transfer_verifying_key(2, 5, 10)
    assert (k.n_inputs, k.n_outputs, k.tree_depth) == (2, 5, 10) and k.verifying_key.n == 32
    prm.store_bytes(blob + prm._freeze_trailer(10, 3), prm.default_freeze_verifying_key_path(3, 10))
    f = prm.load_freeze_verifying_key(3, 10)
    assert (f.tree_depth, f.num_input) == (10, 3)
    prm.store_bytes(blob, prm.default_mint_verifying_key_path(10))
    with pytest.raises(capproof.TxnApiError, match="DeserializationError"):
        prm.load_mint_verifying_key(10)      # trailer missing


# ---- GPU ------------------------------------------------------------------------------------------------------
@pytest.mark.gpu
def test_g1_codec_vs_oracle(cg):
    rng = bn.SplitMix64(77)
    ks = [rng.field(bn.R) for _ in range(300)]
    pts_arr = cr.g1_fixed_base_batch(cr.ints_to_array(ks))
    pts = [cr.affine_to_ints(p) for p in pts_arr]
    pts[10] = bn.INF
    pts[11] = bn.g1_neg(pts[12])
    arr = np.stack([g1_words(p) for p in pts])
    blob = b"".join(bn.g1_serialize_compressed(p) for p in pts)
    assert cg.g1_compress(arr) == blob
    back = cg.g1_decompress(blob)
    assert np.array_equal(back, arr)
    for hexed, point in G["g1_ok"]:
        assert words_g1(cg.g1_decompress(bytes.fromhex(hexed))[0]) == H.unhex_pt(point)
    for name, hexed in G["g1_bad"].items():
        with pytest.raises(cg.CapGpuError) as e:
            cg.g1_decompress(blob[:64] + bytes.fromhex(hexed) + blob[64:96])
        assert e.value.code == cg.CAPGPU_ERR_SERIALIZATION and "point 2" in str(e.value), name
    assert cg.g1_decompress(b"").shape == (0, 8) and cg.g1_compress(np.zeros((0, 8), np.uint64)) == b""


@pytest.mark.gpu
def test_srs_blob_vs_oracle(cg, tau):
    blob = bytes.fromhex(G["srs"])
    n = 32 + 3
    h = cg.srs_generate(tau, n)
    g2h = cg.g2_generator()
    assert cg.srs_serialize(h, g2h, cg.g2_mul(g2h, tau)) == blob            # device SRS -> the oracle's bytes
    h2, hh, bh, used = cg.srs_deserialize(blob + b"xx")
    assert used == len(blob) and cg.srs_size(h2) == n
    assert np.array_equal(cg.srs_download(h2, 0, n), cg.srs_download(h, 0, n))
    assert np.array_equal(hh, g2h) and np.array_equal(bh, cg.g2_mul(g2h, tau))
    # trimmed load keeps a prefix; the MSM over it agrees with the generated SRS
    h3, _, _, _ = cg.srs_deserialize(blob, max_degree=19)
    assert cg.srs_size(h3) == 20
    sc = H.seeded_fr(5, 20, mont=False)
    assert jac_ints(cg.msm_g1(h3, sc)) == jac_ints(cg.msm_g1(h, sc))
    # hiding powers and negative powers of h are validated and skipped
    o = pm.deserialize_universal_params(blob)
    rich = pm.serialize_universal_params(o["powers_of_g"], {0: o["powers_of_g"][3], 7: o["powers_of_g"][1]}, o["h"],
                                         o["beta_h"], {1: pr2.g2_neg(o["beta_h"])})
    h4, hh4, bh4, used = cg.srs_deserialize(rich)
    assert used == len(rich) and np.array_equal(cg.srs_download(h4, 0, n), cg.srs_download(h, 0, n))
    # ... and kept with the handle: load -> store gives the file back byte for byte (jf-plonk's `trim` reads
    # powers_of_gamma_g, so a re-stored file must still hold them - round-1 ADVICE)
    assert cg.srs_serialize(h4, hh4, bh4) == rich
    for hd in (h, h2, h3, h4):
        cg.srs_free(hd)
    # malformed blobs
    bad = bytearray(blob); bad[8 + 32 * 4: 8 + 32 * 5] = bytes.fromhex(G["g1_bad"]["x_not_on_curve"])
    for b in (bytes(bad), blob[:100], blob[:-9], (1 << 60).to_bytes(8, "little") + blob[8:]):
        with pytest.raises(cg.CapGpuError) as e:
            cg.srs_deserialize(b)
        assert e.value.code == cg.CAPGPU_ERR_SERIALIZATION
    off = 8 + 32 * n + 8          # h sits after the (empty) gamma map
    notg2 = bytearray(blob); notg2[off:off + 64] = (1).to_bytes(32, "little") + bytes(32)
    with pytest.raises(cg.CapGpuError):
        cg.srs_deserialize(bytes(notg2))


@pytest.mark.gpu
def test_proving_key_blob_vs_oracle(cg, tau):
    """preprocess on the device -> bytes equal the oracle's; bytes -> key that proves exactly like the original."""
    g = H.load_golden("proof_log5.json")
    sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
    srs = capproof.universal_setup(sc.n + 2, tau)
    pk, vk, _ = capproof.preprocess(srs, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
    blob = prm.serialize_proving_key(pk)
    assert blob.hex() == G["proving_key"]
    assert prm.serialize_verifying_key(vk, srs).hex() == G["vk"]
    pk2, used = prm.deserialize_proving_key(blob + b"\x01\x02")
    assert used == len(blob) and (pk2.n, pk2.num_inputs) == (sc.n, sc.num_inputs)
    assert cg.srs_size(pk2.srs.handle) == sc.n + 3
    w, pubs = sc.witness(g["witness_seed"])
    bl = bu.to_mont_array(bu.blinders(g["blinder_seed"]))
    pa = bu.to_mont_array(pubs)
    p1 = capproof.prove(pk, sc.wires_mont(w), pa, bl, g["ext_msg"].encode())
    p2 = capproof.prove(pk2, sc.wires_mont(w), pa, bl, g["ext_msg"].encode())
    assert H.proof_points(p1) == H.proof_points(p2)
    exp = [H.unhex_pt(p) for p in g["wires_poly_comms"]]
    assert H.proof_points(p2)[0][:5] == exp
    vk2, _ = prm.deserialize_verifying_key(bytes.fromhex(G["vk"]))
    capproof.verify(vk2, pa, p2, g["ext_msg"].encode())          # the loaded verifying key accepts it
    assert prm.serialize_proving_key(pk2) == blob                # and the loaded key serialises to the same bytes
    # a key whose commit key carries hiding powers (CommitKey::powers_of_gamma_g): kept and written back verbatim
    o = pm.deserialize_proving_key(blob)
    rich = pm.serialize_proving_key(o["sigmas"], o["selectors"], o["powers_of_g"], bytes.fromhex(G["vk"]),
                                    gamma_powers=[o["powers_of_g"][2], o["powers_of_g"][5]])
    pk3, used3 = prm.deserialize_proving_key(rich)
    assert used3 == len(rich) and prm.serialize_proving_key(pk3) == rich
    p3 = capproof.prove(pk3, sc.wires_mont(w), pa, bl, g["ext_msg"].encode())
    assert H.proof_points(p3) == H.proof_points(p1)
    # malformed key blobs
    nine = bytearray(blob); nine[0] = 4
    longpoly = bytearray(blob); longpoly[8:16] = (sc.n + 1).to_bytes(8, "little")
    badcoef = bytearray(blob); badcoef[16:48] = bn.R.to_bytes(32, "little")
    for b in (bytes(nine), bytes(longpoly), bytes(badcoef), blob[:5000], blob[:-1]):
        with pytest.raises(capproof.TxnApiError, match="DeserializationError"):
            prm.deserialize_proving_key(b)
    for k in (pk, pk2):
        cg.plonk_free_key(k.handle)
        cg.srs_free(k.srs.handle)


@pytest.mark.gpu
def test_hiding_powers_of_a_loaded_srs_in_key_blobs(cg, tau):
    """CommitKey::powers_of_gamma_g of a ProvingKey blob is a Vec<G1> for degrees 0 .. n + 2 (what jf-plonk's trim
    emits).  A key preprocessed under a loaded UniversalSrs takes them from that SRS's BTreeMap by DEGREE: all of them in
    order, or - when the map lacks one - none; a sparse map must not become a shorter, mislabelled vector (round-2
    ADVICE).  A trimmed load (max_degree) also trims the map, so that a re-stored file is consistent."""
    g = H.load_golden("proof_log5.json")
    sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
    n_ck = sc.n + 3
    o = pm.deserialize_universal_params(bytes.fromhex(G["srs"]))
    pts = o["powers_of_g"]
    assert len(pts) == n_ck
    gamma_full = {d: pts[(5 * d + 1) % n_ck] for d in range(n_ck + 1)}          # degrees 0 .. max_degree + 1
    sparse = {0: pts[3], 7: pts[1]}
    key_blob = {}
    for name, gm in (("full", gamma_full), ("sparse", sparse)):
        blob = pm.serialize_universal_params(pts, gm, o["h"], o["beta_h"], {})
        h, hh, bh, _ = cg.srs_deserialize(blob)
        assert cg.srs_serialize(h, hh, bh) == blob
        pk, _vk = cg.plonk_preprocess(h, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
        key_blob[name] = cg.plonk_key_serialize(pk, hh, bh)
        cg.plonk_free_key(pk)
        cg.srs_free(h)
    plain = pm.deserialize_proving_key(bytes.fromhex(G["proving_key"]))
    v = pm.read_verifying_key(pm.Reader(bytes.fromhex(G["vk"])))

    def vk_bytes(gamma_g):                                       # the open key's gamma_g = degree 0 of the hiding powers
        return pm.serialize_verifying_key(v["domain_size"], v["num_inputs"], v["sigma_comms"], v["selector_comms"], v["k"],
                                          v["g"], gamma_g, v["h"], v["beta_h"])

    want_full = pm.serialize_proving_key(plain["sigmas"], plain["selectors"], plain["powers_of_g"], vk_bytes(gamma_full[0]),
                                         gamma_powers=[gamma_full[d] for d in range(n_ck)])
    assert key_blob["full"] == want_full                        # degrees 0 .. n + 2, in order
    # a map with holes: no hiding powers in the commit key - not two of them - but the open key's gamma_g is degree 0
    assert key_blob["sparse"] == pm.serialize_proving_key(plain["sigmas"], plain["selectors"], plain["powers_of_g"],
                                                          vk_bytes(sparse[0]))
    # trimmed load: powers_of_g cut to max_degree + 1, the map to the degrees a setup of that size holds
    blob = pm.serialize_universal_params(pts, gamma_full, o["h"], o["beta_h"], {})
    h, hh, bh, _ = cg.srs_deserialize(blob, max_degree=19)
    back = pm.deserialize_universal_params(cg.srs_serialize(h, hh, bh))
    assert len(back["powers_of_g"]) == 20 and sorted(back["powers_of_gamma_g"]) == list(range(21))
    assert all(back["powers_of_gamma_g"][d] == gamma_full[d] for d in range(21))
    cg.srs_free(h)


@pytest.mark.gpu
def test_generated_srs_with_hiding_powers(cg, tau):
    """universal_setup(max_degree, tau, gamma): powers_of_gamma_g = { i: [gamma tau^i] G, i <= max_degree + 1 } like
    KZG10::setup's (src/proof/mod.rs:59-69); stored with the SRS and - by degree - with every key preprocessed under it."""
    import oracle.bn254 as bn
    gamma = 0x1D2C3B4A5968778695A4B3C2D1E0F11223344556677889900AABBCCDDEEFF01 % bn.R
    max_degree = 34
    srs = capproof.universal_setup(max_degree, tau, gamma)
    plain = capproof.universal_setup(max_degree, tau)
    o = pm.deserialize_universal_params(prm.serialize_universal_parameter(srs))
    assert o["powers_of_g"] == pm.deserialize_universal_params(prm.serialize_universal_parameter(plain))["powers_of_g"]
    assert sorted(o["powers_of_gamma_g"]) == list(range(max_degree + 2))
    for d in (0, 1, 2, 17, max_degree, max_degree + 1):
        assert o["powers_of_gamma_g"][d] == bn.g1_mul(bn.G1_GEN, gamma * pow(tau, d, bn.R) % bn.R)
    # the oracle's own setup gives the same file
    want = pm.serialize_universal_params(o["powers_of_g"],
                                         {d: bn.g1_mul(bn.G1_GEN, gamma * pow(tau, d, bn.R) % bn.R)
                                          for d in range(max_degree + 2)}, o["h"], o["beta_h"], {})
    assert prm.serialize_universal_parameter(srs) == want
    # a key preprocessed under it: commit key = degrees 0 .. n + 2 of both vectors, open key's gamma_g = degree 0
    g = H.load_golden("proof_log5.json")
    sc = bu.synthetic_circuit(g["log_n"], g["num_inputs"], seed=g["circuit_seed"])
    pk, vk, _ = capproof.preprocess(srs, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
    k = pm.deserialize_proving_key(prm.serialize_proving_key(pk))
    assert k["powers_of_gamma_g"] == [o["powers_of_gamma_g"][d] for d in range(sc.n + 3)]
    assert k["vk"]["gamma_g"] == o["powers_of_gamma_g"][0]
    # ... and proofs are the same as under the plain SRS: commitments stay non-hiding
    w, pubs = sc.witness(g["witness_seed"])
    bl = bu.to_mont_array(bu.blinders(g["blinder_seed"]))
    pa = bu.to_mont_array(pubs)
    p = capproof.prove(pk, sc.wires_mont(w), pa, bl, g["ext_msg"].encode())
    assert H.proof_points(p)[0][:5] == [H.unhex_pt(q) for q in g["wires_poly_comms"]]
    capproof.verify(vk, pa, p, g["ext_msg"].encode())


@pytest.mark.gpu
def test_load_srs_mirror(cg, tau):
    """capproof.load_srs (src/proof/mod.rs:74-109) on this SRS's own ark-serialize bytes: the whole file is loaded."""
    import hashlib
    blob = bytes.fromhex(G["srs"])
    srs = capproof.load_srs(20, blob, hashlib.sha256(blob).digest())
    assert srs.max_degree == 34 and cg.srs_size(srs.handle) == 35
    cg.srs_free(srs.handle)
    # src/proof/mod.rs:121-141: the staging API of the default (bn254) feature is load_srs with its rng ignored
    srs2 = capproof.universal_setup_for_staging(20, object(), blob, hashlib.sha256(blob).digest())
    assert srs2.max_degree == 34
    cg.srs_free(srs2.handle)


def test_load_srs_guards_run_before_any_device_work():
    """the degree bound and the integrity assert of src/proof/mod.rs:83-102 need no GPU"""
    import hashlib
    blob = bytes.fromhex(G["srs"])
    with pytest.raises(capproof.TxnApiError, match="only supports 2\\^17"):
        capproof.load_srs((1 << 17) + 1, blob, hashlib.sha256(blob).digest())
    with pytest.raises(AssertionError, match="Mismatched sha256sum digest"):
        capproof.load_srs(1 << 17, blob)                         # the reference's pinned digest: not the Aztec bytes
    bad = bytearray(blob)
    bad[100] ^= 1
    with pytest.raises(AssertionError, match="Mismatched sha256sum digest"):
        capproof.load_srs(100, bytes(bad), hashlib.sha256(blob).digest())
    assert capproof.AZTEC_CRS_SHA256.hex() == "6b81e75fb9c14fd0e58fb2b29e48978cdad5511503685a61f1391dc4a4fc7cbf"
    # universal_setup_for_staging (src/proof/mod.rs:121-141) is the same function: same guards, rng untouched
    with pytest.raises(capproof.TxnApiError, match="only supports 2\\^17"):
        capproof.universal_setup_for_staging((1 << 17) + 1, None, blob, hashlib.sha256(blob).digest())
    with pytest.raises(AssertionError, match="Mismatched sha256sum digest"):
        capproof.universal_setup_for_staging(1 << 17, None, blob)


@pytest.mark.gpu
def test_blob_parsers_survive_random_corruption(cg):
    """Same for the device-backed parsers (UniversalSrs, ProvingKey): error or success, never a crash or a HIP fault,
    and every handle a successful parse returns is usable and can be freed."""
    rng = np.random.default_rng(7)
    for name, parse in (("srs", lambda b: cg.srs_deserialize(b)[:1]),
                        ("proving_key", lambda b: cg.plonk_key_deserialize(b)[:2])):
        blob = bytes.fromhex(G[name])
        ok = bad = 0
        for trial in range(60):
            b = bytearray(blob)
            kind = trial % 3
            if kind == 0:
                for pos in rng.integers(0, len(b), size=int(rng.integers(1, 4))):
                    b[pos] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                b = b[:int(rng.integers(0, len(b)))]
            else:
                pos = int(rng.integers(0, len(b) - 8))
                b[pos:pos + 8] = int(rng.integers(0, 1 << 62)).to_bytes(8, "little")
            try:
                handles = parse(bytes(b))
                ok += 1
                if name == "srs":
                    assert cg.srs_size(handles[0]) > 0
                    cg.srs_free(handles[0])
                else:
                    cg.plonk_free_key(handles[1])
                    cg.srs_free(handles[0])
            except cg.CapGpuError as e:
                assert e.code in (cg.CAPGPU_ERR_SERIALIZATION, -1), (name, trial, e)
                bad += 1
        assert bad > 20, (name, ok, bad)
    # the library is still healthy afterwards
    h = cg.srs_deserialize(bytes.fromhex(G["srs"]))[0]
    assert cg.srs_size(h) == 35
    cg.srs_free(h)


@pytest.mark.gpu
def test_store_and_load_round_trips(cg, tau, tmp_path, monkeypatch):
    """Counterparts of src/parameters.rs:603-688: store, load, check the note-shape fields."""
    monkeypatch.setenv("CAP_UNIV_PARAM_DIR", str(tmp_path))
    (tmp_path / "data").mkdir()
    sc = bu.synthetic_circuit(6, 4, seed=9)
    srs = capproof.universal_setup(sc.n + 2, tau)
    prm.store_universal_parameter_for_demo(srs)
    srs2 = prm.load_universal_parameter()
    assert srs2.max_degree == srs.max_degree
    assert np.array_equal(cg.srs_download(srs2.handle, 0, sc.n + 3), cg.srs_download(srs.handle, 0, sc.n + 3))
    pk, vk, _ = capproof.preprocess(srs2, sc.n, sc.num_inputs, sc.selectors_mont(), sc.sigma_mont())
    prm.store_transfer_proving_key(prm.TransferProvingKey(pk, 2, 5, 10))
    prm.store_transfer_verifying_key(prm.TransferVerifyingKey(vk, 2, 5, 10), srs2)
    tk = prm.load_transfer_proving_key(2, 5, 10)
    tv = prm.load_transfer_verifying_key(2, 5, 10)
    assert (tk.n_inputs, tk.n_outputs, tk.tree_depth) == (2, 5, 10) == (tv.n_inputs, tv.n_outputs, tv.tree_depth)
    prm.store_mint_proving_key(prm.MintProvingKey(pk, 10))
    prm.store_mint_verifying_key(prm.MintVerifyingKey(vk, 10), srs2)
    assert prm.load_mint_proving_key(10).tree_depth == 10 == prm.load_mint_verifying_key(10).tree_depth
    prm.store_freeze_proving_key(prm.FreezeProvingKey(pk, 10, 2))
    prm.store_freeze_verifying_key(prm.FreezeVerifyingKey(vk, 10, 2), srs2)
    fk, fv = prm.load_freeze_proving_key(2, 10), prm.load_freeze_verifying_key(2, 10)
    assert (fk.num_input, fk.tree_depth) == (2, 10) == (fv.num_input, fv.tree_depth)
    # a proof made with the loaded transfer key verifies under the loaded verifying key
    w, pubs = sc.witness(3)
    pa = bu.to_mont_array(pubs)
    p = capproof.prove(tk.proving_key, sc.wires_mont(w), pa, bu.to_mont_array(bu.blinders(4)), b"vk")
    capproof.verify(tv.verifying_key, pa, p, b"vk")
    with pytest.raises(capproof.TxnApiError):
        capproof.verify(tv.verifying_key, pa, p, b"other")


@pytest.mark.gpu
def test_full_size_srs_round_trip(cg, tau):
    """2^17 + 3 powers (the size of the reference's Aztec CRS, src/proof/mod.rs:79-93): compress -> decompress is the
    identity, and the loaded SRS satisfies the known-tau MSM identity  MSM(P, coeffs f) = [f(tau)] G."""
    n = (1 << 17) + 3
    h = cg.srs_generate(tau, n)
    g2h = cg.g2_generator()
    blob = cg.srs_serialize(h, g2h, cg.g2_mul(g2h, tau))
    assert len(blob) == 8 + 32 * n + 8 + 128 + 8
    h2, _, _, used = cg.srs_deserialize(blob)
    assert used == len(blob)
    for off in (0, 70001, n - 64):
        assert np.array_equal(cg.srs_download(h2, off, 64), cg.srs_download(h, off, 64))
    coeffs = H.seeded_fr(17, n, mont=False)
    f_tau, x = 0, 1
    for c in cr.array_to_ints(coeffs):
        f_tau = (f_tau + c * x) % bn.R
        x = x * tau % bn.R
    assert jac_ints(cg.msm_g1(h2, coeffs)) == bn.g1_mul(bn.G1_GEN, f_tau)
    cg.srs_free(h)
    cg.srs_free(h2)
