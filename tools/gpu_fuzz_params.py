"""Mutation fuzz of the on-disk parameter loaders (UniversalSrs and ProvingKey blobs): bit flips, truncations and
overwritten length fields must end in CAPGPU_ERR_SERIALIZATION / INVALID_ARG or in a usable object - never in a crash,
a hang or a leak of handles.  python tools/gpu_fuzz_params.py [rounds] [seed]
(also run, bounded, by tests/test_gpu_fuzz.py)"""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu  # noqa: E402

def run(rounds=300, seed=1, log=print):
    """returns {'srs': [loaded, refused], 'key': [loaded, refused]}; raises on anything but an honest outcome"""
    rng = random.Random(seed)
    cg.init(0)
    tau = bu.SplitMix64(5).field()
    n = 64
    srs = cg.srs_generate(tau, n + 3)
    h2 = cg.g2_generator()
    bh = cg.g2_mul(h2, tau)
    sc = bu.synthetic_circuit(6, 3, seed=4)
    pk, vk = cg.plonk_preprocess(srs, n, 3, sc.selectors_mont(), sc.sigma_mont())
    blobs = {"srs": cg.srs_serialize(srs, h2, bh), "key": cg.plonk_key_serialize(pk, h2, bh)}
    w, pubs = sc.witness(1)
    wm, pm, bl = sc.wires_mont(w), bu.to_mont_array(pubs), bu.to_mont_array(bu.blinders(1))
    ref = bytes(cg.plonk_prove_batch(pk, wm[None], pm[None], bl[None], None, 1)[0])
    stats = {"srs": [0, 0], "key": [0, 0]}
    for r in range(rounds):
        kind = rng.choice(["srs", "key"])
        b = bytearray(blobs[kind])
        mode = rng.random()
        if mode < 0.45:
            for _ in range(rng.choice([1, 1, 3])):
                b[rng.randrange(len(b))] ^= 1 << rng.randrange(8)
        elif mode < 0.7:
            b = b[:rng.randrange(len(b))]
        elif mode < 0.9:                       # a length prefix somewhere near the start of a vector
            i = rng.choice([0, 8, 16]) if rng.random() < 0.5 else rng.randrange(0, len(b) - 8, 8)
            b[i:i + 8] = rng.choice([0, 1, n + 2, n + 3, n + 4, 1 << 20, (1 << 64) - 1, rng.getrandbits(64)]).to_bytes(8, "little")
        else:
            b += bytes(rng.getrandbits(8) for _ in range(rng.randrange(1, 40)))      # trailing bytes are the caller's
        try:
            if kind == "srs":
                hdl, _, _, used = cg.srs_deserialize(bytes(b))
                cg.srs_free(hdl)
            else:
                s2, p2, vk2, _, _, used = cg.plonk_key_deserialize(bytes(b))
                # a key that loads must prove: either the same proof (the mutation hit nothing the prover reads) or an honest error
                try:
                    got = bytes(cg.plonk_prove_batch(p2, wm[None], pm[None], bl[None], None, 1)[0])
                except cg.CapGpuError:
                    got = None
                cg.plonk_free_key(p2)
                cg.srs_free(s2)
            stats[kind][0] += 1
        except cg.CapGpuError as e:
            assert e.code in (cg.CAPGPU_ERR_SERIALIZATION, -1, -5), (kind, e)
            stats[kind][1] += 1
    # the library still works afterwards
    assert bytes(cg.plonk_prove_batch(pk, wm[None], pm[None], bl[None], None, 1)[0]) == ref
    log(f"no crash; loaded / refused: {stats}")
    cg.plonk_free_key(pk)
    cg.srs_free(srs)
    return stats


if __name__ == "__main__":
    run(int(sys.argv[1]) if len(sys.argv) > 1 else 300, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
