"""Generates tests/golden/*.json from the Python big-int oracle (oracle/bn254.py, oracle/plonk.py).

Run here (CPU container):  python tests/golden/make_golden.py
The reference (Rust) cannot be built or imported in this environment and holds no vectors for this path
(SURVEY.md §8c), so these fixtures pin the *oracle's* outputs on seeded inputs; inputs are re-derived from
the seeds (SplitMix64 -> 4 words LE -> mod r) and only expected outputs are stored.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from cap_amd import bench_utils as bu  # noqa: E402  (workload synthesis only)
from oracle import bn254 as bn  # noqa: E402
from oracle import pairing as pr2  # noqa: E402
from oracle import params as pm  # noqa: E402
from oracle import plonk as pl  # noqa: E402


def hx(v):
    return "%064x" % v


def pt(p):
    return None if p is None else [hx(p[0]), hx(p[1])]


def msm_vectors():
    out = []
    for n in (1, 2, 3, 31, 32, 33, 100):
        ks = bn.SplitMix64(1000 + n)
        bases = [bn.g1_mul(bn.G1_GEN, ks.field(bn.R)) for _ in range(n)]
        sc = bn.SplitMix64(2000 + n)
        scalars = [sc.field(bn.R) for _ in range(n)]
        out.append({"n": n, "base_seed": 1000 + n, "scalar_seed": 2000 + n, "edge": False,
                    "result": pt(bn.msm_naive(bases, scalars))})
    # edge vector: zero / one / r-1 / window-boundary scalars, duplicate bases, P and -P, infinity base
    n = 40
    ks = bn.SplitMix64(1040)
    bases = [bn.g1_mul(bn.G1_GEN, ks.field(bn.R)) for _ in range(n)]
    bases[7] = bases[6]
    bases[5] = None
    sc = bn.SplitMix64(2040)
    scalars = [sc.field(bn.R) for _ in range(n)]
    scalars[0:5] = [0, 1, bn.R - 1, 2**13 - 1, 2**13]
    scalars[6], scalars[7] = 5, bn.R - 5
    scalars[8] = (1 << 254) - 1 - bn.R  # all-ones low windows after reduction is irrelevant: still < r
    scalars[9] = int("1" * 253, 2) % bn.R
    out.append({"n": n, "base_seed": 1040, "scalar_seed": 2040, "edge": True,
                "scalars": [hx(s) for s in scalars], "result": pt(bn.msm_naive(bases, scalars))})
    return out


def ntt_vectors():
    out = []
    for log_n in range(0, 9):
        rng = bn.SplitMix64(3000 + log_n)
        a = [rng.field(bn.R) for _ in range(1 << log_n)]
        out.append({"log_n": log_n, "seed": 3000 + log_n,
                    "ntt": [hx(v) for v in bn.ntt(a, log_n)], "intt": [hx(v) for v in bn.intt(a, log_n)],
                    "coset_ntt": [hx(v) for v in bn.coset_ntt(a, log_n)],
                    "coset_intt": [hx(v) for v in bn.coset_intt(a, log_n)]})
    return out


def proof_vector():
    tau = bn.SplitMix64(0xCA9).field(bn.R)
    sc = bu.synthetic_circuit(5, 3, seed=2)
    w, pubs = sc.witness(100)
    bl = bu.blinders(200)
    c = pl.Circuit(n=sc.n, num_inputs=3, selectors=sc.selectors, sigma=sc.sigma)
    pk = pl.preprocess(c, tau)
    pr = pl.prove(pk, w, pubs, bl, ext_msg=b"memo-key")
    assert pl.verify(sc.n, 3, pk.selector_comms, pk.sigma_comms, pubs, pr, tau, ext_msg=b"memo-key")
    return {"log_n": 5, "num_inputs": 3, "circuit_seed": 2, "witness_seed": 100, "blinder_seed": 200,
            "tau_seed": 0xCA9, "ext_msg": "memo-key",
            "selector_comms": [pt(p) for p in pk.selector_comms], "sigma_comms": [pt(p) for p in pk.sigma_comms],
            "wires_poly_comms": [pt(p) for p in pr.wires_poly_comms], "prod_perm_poly_comm": pt(pr.prod_perm_poly_comm),
            "split_quot_poly_comms": [pt(p) for p in pr.split_quot_poly_comms], "opening_proof": pt(pr.opening_proof),
            "shifted_opening_proof": pt(pr.shifted_opening_proof), "wires_evals": [hx(v) for v in pr.wires_evals],
            "wire_sigma_evals": [hx(v) for v in pr.wire_sigma_evals], "perm_next_eval": hx(pr.perm_next_eval)}


def params_vector():
    """On-disk formats (oracle/params.py) for the log-5 circuit of proof_log5.json under the same tau."""
    tau = bn.SplitMix64(0xCA9).field(bn.R)
    sc = bu.synthetic_circuit(5, 3, seed=2)
    c = pl.Circuit(n=sc.n, num_inputs=3, selectors=sc.selectors, sigma=sc.sigma)
    pk = pl.preprocess(c, tau)
    powers, x = [], 1
    for _ in range(sc.n + 3):
        powers.append(bn.g1_mul(bn.G1_GEN, x))
        x = x * tau % bn.R
    h, beta_h = pr2.G2_GEN, pr2.g2_mul(pr2.G2_GEN, tau)
    srs = pm.serialize_universal_params(powers, {}, h, beta_h, {})
    vk = pm.serialize_verifying_key(sc.n, 3, pk.sigma_comms, pk.selector_comms, pl.K, powers[0], bn.INF, h, beta_h)
    key = pm.serialize_proving_key(pk.sigma_polys, pk.selector_polys, powers, vk)
    # compressed G1 edge cases: infinity, both roots of one x, and encodings ark-serialize rejects
    p5 = bn.g1_mul(bn.G1_GEN, 5)
    x_off = next(v for v in range(1, 50) if pm.fq_sqrt(v ** 3 + 3) is None)
    bad = {"x_not_on_curve": x_off.to_bytes(32, "little").hex(),
           "x_not_canonical": (bn.P + 1).to_bytes(32, "little").hex(),
           "both_flags": (bytes(31) + b"\xc0").hex(),
           "infinity_with_x": (b"\x01" + bytes(30) + b"\x40").hex()}
    return {"log_n": 5, "num_inputs": 3, "circuit_seed": 2, "tau_seed": 0xCA9, "srs": srs.hex(), "vk": vk.hex(),
            "proving_key": key.hex(),
            "g1_ok": [[bn.g1_serialize_compressed(q).hex(), pt(q)] for q in (bn.INF, p5, bn.g1_neg(p5), bn.G1_GEN)],
            "g1_bad": bad,
            "g2": [[pm.g2_serialize_compressed(q).hex(), [[hx(q[0][0]), hx(q[0][1])], [hx(q[1][0]), hx(q[1][1])]]]
                   for q in (h, beta_h, pr2.g2_neg(beta_h))]}


if __name__ == "__main__":
    for name, fn in (("msm.json", msm_vectors), ("ntt.json", ntt_vectors), ("proof_log5.json", proof_vector),
                     ("params.json", params_vector)):
        with open(os.path.join(HERE, name), "w") as f:
            json.dump(fn(), f, indent=0)
        print("wrote", name)
