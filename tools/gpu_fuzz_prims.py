"""Differential fuzz of the MSM and NTT entry points against the C restatement: random sizes (not powers of two),
offsets, batches of unequal lengths, skewed / tiny / extreme scalars, duplicate and opposite bases, infinity among the
bases; NTT sizes 2^0 .. 2^14 in all four modes.  python tools/gpu_fuzz_prims.py [rounds] [seed]
(also run, bounded, by tests/test_gpu_fuzz.py)"""
import random
import sys

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu  # noqa: E402
from oracle import capref as cr  # noqa: E402  (checker)
from oracle import bn254 as bn  # noqa: E402

def run(rounds=100, seed=1, log=print):
    """returns the number of rounds with a mismatch"""
    rng = random.Random(seed)
    cg.init(0)
    bad = 0

    def scalars(n, kind):
        out = []
        for _ in range(n):
            if kind == "uniform":
                v = rng.randrange(bn.R)
            elif kind == "small":
                v = rng.randrange(1 << rng.choice([1, 3, 13, 15, 16, 30]))
            elif kind == "extreme":
                v = rng.choice([0, 1, 2, bn.R - 1, bn.R - 2, (1 << 13) - 1, 1 << 13, (1 << 15) - 1, 1 << 15, (1 << 253) - 1,
                                int("1" * 13 * 19, 2) % bn.R, bn.R // 2])
            else:  # skewed: most scalars share one value
                v = 0x1234567 if rng.random() < 0.9 else rng.randrange(bn.R)
            out.append(v)
        return bu.to_canonical_array(out)

    for r in range(rounds):
        n_srs = rng.choice([1, 2, 3, 17, 31, 32, 33, 100, 1000, 1025, 2047, 4096, 4100, 9000])
        a, b = rng.randrange(1, bn.R), rng.randrange(bn.R)
        h = cg.srs_generate_affine_seq(a, b, n_srs)
        bases = cg.srs_download(h, 0, n_srs).copy()
        if n_srs >= 3 and rng.random() < 0.5:          # duplicates, opposites and infinity among the bases: re-upload
            bases = bases.reshape(n_srs, 8)
            i, j, k = rng.sample(range(n_srs), 3)
            bases[j] = bases[i]                          # a duplicate: P + P in one bucket when the digits agree
            if n_srs >= 5:
                m = rng.choice([t for t in range(n_srs) if t not in (i, j, k)])
                y = int.from_bytes(bases[i, 4:8].tobytes(), "little")
                bases[m, 0:4] = bases[i, 0:4]
                bases[m, 4:8] = np.frombuffer(((bn.P - y) % bn.P).to_bytes(32, "little"), dtype=np.uint64)   # -P
            bases[k] = 0                                 # the point at infinity
            cg.srs_free(h)
            h = cg.srs_upload(bases)
        cnt = rng.choice([1, 1, 2, 5, 40])
        scs, offs = [], []
        for _ in range(cnt):
            off = rng.randrange(n_srs)
            n = rng.randint(1, n_srs - off)
            offs.append(off)
            scs.append(scalars(n, rng.choice(["uniform", "uniform", "small", "extreme", "skewed"])))
        got = cg.msm_g1_batch(h, scs, offs) if cnt > 1 or rng.random() < 0.5 else cg.msm_g1(h, scs[0], offs[0])[None]
        ok = True
        flat = np.asarray(bases).reshape(n_srs, 8)
        for i in range(cnt):
            exp = cr.g1_to_affine(cr.msm_g1(flat[offs[i]:offs[i] + len(scs[i])], scs[i]))
            ok = ok and np.array_equal(cr.g1_to_affine(got[i]), exp)
        # NTT
        log_n = rng.randint(0, 14)
        inv, coset = rng.random() < 0.5, rng.random() < 0.5
        data = bu.to_mont_array([rng.randrange(bn.R) if rng.random() < 0.9 else 0 for _ in range(1 << log_n)])
        ok_ntt = np.array_equal(cg.ntt_fr(data.copy(), log_n, inverse=inv, coset=coset), cr.ntt_fr(data.copy(), log_n, inv, coset))
        log(f"round {r}: srs={n_srs} msms={cnt} ntt=2^{log_n}{'i' if inv else 'f'}{'c' if coset else ''} "
              f"{'ok' if ok and ok_ntt else 'MISMATCH msm=%s ntt=%s' % (ok, ok_ntt)}")
        bad += 0 if (ok and ok_ntt) else 1
        cg.srs_free(h)
    return bad


if __name__ == "__main__":
    bad = run(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    print("mismatches:", bad)
    sys.exit(1 if bad else 0)
