"""Ad-hoc GPU parity check for MSM / NTT against the C oracle (development aid; the real tests are in tests/)."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg  # noqa: E402
from oracle import bn254 as bn  # noqa: E402
from oracle import capref as cr  # noqa: E402

cg.init(0)
print(cg.load().capgpu_version().decode())
ok = True
# ---- NTT
for log_n in [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 15, 16, 18, 19, 21]:
    a = cr.random_field(100 + log_n, 1, 1 << log_n, True)
    for inv in (False, True):
        for coset in (False, True):
            if log_n >= 18 and (inv != coset):
                continue
            t = time.time(); got = cg.ntt_fr(a, log_n, inv, coset); tg = time.time() - t
            t = time.time(); exp = cr.ntt_fr(a, log_n, inv, coset); tc = time.time() - t
            good = np.array_equal(got.reshape(-1), exp.reshape(-1))
            ok &= good
            print(f"ntt log_n={log_n} inv={inv} coset={coset}: {'OK' if good else 'MISMATCH'} "
                  f"gpu {tg*1e3:.1f} ms cpu {tc*1e3:.1f} ms", flush=True)
# ---- MSM
nmax = 1 << 13
ks = cr.random_field(11, 1, nmax, False)
bases = cr.g1_fixed_base_batch(ks)
bases[5] = 0           # a point at infinity among the bases
bases[7] = bases[6]    # duplicate base
h = cg.srs_upload(bases)
dl = cg.srs_download(h, 0, 16)
print("srs roundtrip", np.array_equal(dl, bases[:16]))
for n in [0, 1, 2, 3, 31, 32, 33, 100, 1000, 4099, nmax]:
    sc = cr.random_field(12 + n, 1, max(n, 1), False)[:n]
    if n >= 8:
        sc[0] = 0; sc[1] = cr.int_to_limbs(1); sc[2] = cr.int_to_limbs(bn.R - 1); sc[3] = cr.int_to_limbs(2**13 - 1)
        sc[4] = cr.int_to_limbs(2**13); sc[6] = cr.int_to_limbs(5); sc[7] = cr.int_to_limbs(bn.R - 5)
    t = time.time(); got = cg.msm_g1(h, sc); tg = time.time() - t
    t = time.time(); exp = cr.msm_g1(bases[:n], sc); tc = time.time() - t
    ga = cr.g1_to_affine(got); ea = cr.g1_to_affine(exp)
    good = np.array_equal(ga, ea)
    ok &= good
    print(f"msm n={n}: {'OK' if good else 'MISMATCH'} gpu {tg*1e3:.1f} ms cpu {tc*1e3:.1f} ms", flush=True)
# batch + offset
scs = [cr.random_field(500 + i, 1, 1000, False) for i in range(5)]
got = cg.msm_g1_batch(h, scs, offsets=[3] * 5)
for i in range(5):
    good = np.array_equal(cr.g1_to_affine(got[i]), cr.g1_to_affine(cr.msm_g1(bases[3:1003], scs[i])))
    ok &= good
    print("batch", i, "OK" if good else "MISMATCH")
# generated SRS: known-tau identity
tau = bn.SplitMix64(0xCA9).field(bn.R)
n = 1 << 12
h2 = cg.srs_generate(tau, n)
pts = cg.srs_download(h2, 0, 4)
exp_pts = [bn.g1_mul(bn.G1_GEN, pow(tau, i, bn.R)) for i in range(4)]
good = [cr.affine_to_ints(p) for p in pts] == exp_pts
ok &= good
print("srs_generate first points", "OK" if good else "MISMATCH")
coef = cr.random_field(77, 1, n, False); ci = cr.array_to_ints(coef)
got = cr.affine_to_ints(cr.g1_to_affine(cg.msm_g1(h2, coef)))
ftau = bn.poly_eval(ci, tau)
good = got == bn.g1_mul(bn.G1_GEN, ftau)
ok &= good
print("known-tau identity", "OK" if good else "MISMATCH")
# timing with resident data
cg.profile_enable(True)
for (n, cnt) in [(1 << 12, 1), (1 << 12, 5)]:
    sc = np.concatenate([cr.random_field(900 + i, 1, n, False) for i in range(cnt)])
    d_sc = cg.DevBuf.from_numpy(sc)
    cg.profile_reset()
    for _ in range(3):
        d_out = cg.msm_g1_dev(h2, d_sc, n, count=cnt)
    cg.sync()
    print("msm resident", n, cnt, {k: (round(v[0] / v[1] * 1e3, 1), v[1]) for k, v in cg.profile_stats().items()})
for log_n in (15, 18):
    a = cr.random_field(5, 1, 1 << log_n, True)
    d = cg.DevBuf.from_numpy(a)
    cg.profile_reset()
    for _ in range(4):
        cg.ntt_fr_dev(d, log_n)
    cg.sync()
    print("ntt resident", log_n, {k: (round(v[0] / v[1] * 1e3, 1), v[1]) for k, v in cg.profile_stats().items()})
print("ALL OK" if ok else "FAILURES")
