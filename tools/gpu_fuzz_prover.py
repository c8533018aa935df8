"""Differential fuzz of the device prover against the C restatement: random domain sizes, public-input counts,
batch sizes, messages, witness value distributions (uniform / mostly booleans and small values, varying padding),
single- and mixed-key batches, both input forms of the ABI (tables of values / polynomials in
coefficient form, for the keys and for the wires independently), and every batch proved three times (small batches are
captured as hipGraphs on the second call and replayed on the third: all three must be the same bytes).  python tools/gpu_fuzz_prover.py [rounds] [seed] [big]
(also run, bounded, by tests/test_gpu_fuzz.py)"""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu  # noqa: E402
from oracle import capref as cr  # noqa: E402  (checker)
from tests import helpers as H  # noqa: E402

# "big": shapes that cross the plan thresholds of the prover's MSM launches (narrow / wide tables at 32 MSMs per launch,
# log-depth / running-sum reduction at 64 of them) instead of random ones
BIG = [(12, 6), (12, 7), (12, 12), (12, 13), (12, 33), (13, 7), (13, 14), (11, 64), (14, 3)]


def run(rounds=30, seed=1, big=False, max_checked=None, log=print):
    """returns the number of rounds with a mismatch.  max_checked: proofs per round compared with the C restatement
    (None: all of them); the others are still compared across the two entry points."""
    rng = random.Random(seed)
    cg.init(0)
    tau = bu.SplitMix64(77).field()
    bad = 0
    if big:
        rounds = len(BIG)
    for r in range(rounds):
        log_n = BIG[r][0] if big else rng.choice([4, 4, 5, 5, 6, 7, 8, 9, 10, 11, 12, 13])
        n = 1 << log_n
        srs = cg.srs_generate(tau, n + 3)
        srs_host = cg.srs_download(srs, 0, n + 3)
        # (a C-side key costs 18 CPU MSMs: the threshold shapes keep to one or two keys)
        nkeys = (1 if log_n >= 13 else rng.choice([1, 2])) if big else rng.choice([1, 1, 2, 3])
        circuits, keys, ckeys = [], [], []

        def coeffs(cols):     # columns of values -> coefficient form, by the CPU restatement's inverse NTT
            flat = np.ascontiguousarray(cols, dtype=np.uint64).reshape(-1, n, 4)
            return np.stack([cr.ntt_fr(c, log_n, True, False).reshape(-1, 4) for c in flat]).reshape(np.shape(cols))

        key_coeffs, wire_coeffs = rng.random() < 0.5, rng.random() < 0.5
        for k in range(nkeys):
            ni = rng.randint(0, min(30, n // 2 - 1))
            # (witness value classes too: uniform, or most free variables booleans / 64-bit values, and more or less padding -
            # the scalars of round 1's commitments from evaluations: heavy buckets, sparse digits, zero columns)
            sc = bu.synthetic_circuit(log_n, ni, seed=rng.randint(1, 10 ** 6), fill=rng.choice([0.94, 0.94, 0.6, 0.3]),
                                      skew=rng.choice([0.0, 0.0, 0.5, 0.97]))
            circuits.append(sc)
            if key_coeffs:
                keys.append(cg.plonk_preprocess(srs, n, ni, coeffs(sc.selectors_mont()), coeffs(sc.sigma_mont()),
                                                input_form="coeffs")[0])
            else:
                keys.append(cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())[0])
            ckeys.append(cr.PlonkKey(srs_host, n, ni, sc.selectors_mont(), sc.sigma_mont()))
        P = BIG[r][1] if big else (rng.choice([1, 2, 3, 5, 8, 13, 40]) if log_n <= 10 else rng.randint(1, 4))
        order = [rng.randrange(nkeys) for _ in range(P)]
        max_in = max(circuits[k].num_inputs for k in order)      # the row length follows the keys actually in the batch
        checked = set(range(P)) if max_checked is None or P <= max_checked else \
            {0, P - 1} | set(rng.sample(range(P), max_checked - 2))
        wires, rows, blinds, msgs, exp = [], [], [], [], {}
        for i, k in enumerate(order):
            sc = circuits[k]
            w, pubs = sc.witness(rng.randint(1, 10 ** 6))
            bl = bu.to_mont_array(bu.blinders(rng.randint(1, 10 ** 6)))
            msg = bytes(rng.getrandbits(8) for _ in range(rng.choice([0, 1, 31, 32, 33, 100])))
            row = np.zeros((max_in, 4), np.uint64)
            pm = bu.to_mont_array(pubs) if pubs else np.zeros((0, 4), np.uint64)
            row[:len(pubs)] = pm
            wires.append(sc.wires_mont(w)); rows.append(row); blinds.append(bl); msgs.append(msg)
            if i in checked:
                rc, comms, evals = ckeys[k].prove(sc.wires_mont(w), pm, bl, msg or None)
                assert rc == 0
                exp[i] = H.cref_proof_points(comms, evals)
        w_in = coeffs(np.stack(wires)) if wire_coeffs else np.stack(wires)
        form = "coeffs" if wire_coeffs else "evals"
        got = cg.plonk_prove_multi([keys[k] for k in order], w_in, np.stack(rows), np.stack(blinds), msgs, input_form=form)
        ok = all(H.proof_points(got[i]) == exp[i] for i in checked)
        for _ in range(2):    # the second call captures the schedule of a small batch, the third replays it
            again = cg.plonk_prove_multi([keys[k] for k in order], w_in, np.stack(rows), np.stack(blinds), msgs,
                                         input_form=form)
            ok = ok and [bytes(p) for p in again] == [bytes(p) for p in got]
        if nkeys == 1:
            pm = np.stack(rows)[:, :circuits[0].num_inputs]
            for i in range(P):        # the single-key entry point too (per-proof message)
                one = cg.plonk_prove_batch(keys[0], wires[i][None], pm[i][None], blinds[i][None], msgs[i] or None, 1)[0]
                ok = ok and bytes(one) == bytes(got[i])
        log(f"round {r}: log_n={log_n} keys={nkeys} P={P} inputs={[c.num_inputs for c in circuits]} "
            f"key_form={'coeffs' if key_coeffs else 'evals'} wire_form={form} {'ok' if ok else 'MISMATCH'}")
        bad += 0 if ok else 1
        for k in keys:
            cg.plonk_free_key(k)
        cg.srs_free(srs)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 30, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
              len(sys.argv) > 3 and sys.argv[3] == "big")
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
