"""The device prover evaluates the quotient identity on a coset of 6n = 3 * 2^(log n + 1) points, not on jf-plonk's 8n
(cap_amd/csrc/ntt.hpp).  This CPU test pins the mathematics behind that with Python integers: the 6n-th root of unity
built as omega_3 * omega_2n^c (3c = 1 mod 2n) has the two properties the kernels rely on, and the quotient polynomial
interpolated from the 6n coset is the oracle's (which uses 8n like the reference), coefficient for coefficient."""
from cap_amd import bench_utils as bu
from oracle import bn254 as bn
from oracle import plonk as pl

R = bn.R


def root_6n(log_n: int):
    m_small = 1 << (log_n + 1)                       # M = 2n
    w3 = pow(bn.FR_GENERATOR, (R - 1) // 3, R)
    assert w3 != 1 and pow(w3, 3, R) == 1
    c = (m_small + 1) // 3 if m_small % 3 == 2 else (2 * m_small + 1) // 3
    assert 3 * c % m_small == 1
    return w3 * pow(bn.root_of_unity(log_n + 1), c, R) % R, m_small


def test_root_of_unity_of_order_6n():
    for log_n in (3, 4, 9, 15, 16):
        w, m_small = root_6n(log_n)
        big = 3 * m_small
        assert pow(w, big, R) == 1 and pow(w, big // 2, R) != 1 and pow(w, big // 3, R) != 1   # order exactly 6n
        assert pow(w, 3, R) == bn.root_of_unity(log_n + 1)      # the three sub-transforms are ordinary 2n-point ones
        assert pow(w, 6, R) == bn.root_of_unity(log_n)          # six points further = the next row of the circuit
        if log_n <= 4:
            # the kernels keep the 6n points as three cosets of the 2n-point domain, block a = 5 w^a <omega_2n>
            g, wm = bn.FR_GENERATOR, bn.root_of_unity(log_n + 1)
            blocks = {g * pow(w, a, R) % R * pow(wm, k, R) % R for a in range(3) for k in range(m_small)}
            assert blocks == {g * pow(w, i, R) % R for i in range(big)}


def test_quotient_from_the_6n_coset_equals_the_oracles():
    log_n, nin = 3, 2
    n = 1 << log_n
    sc = bu.synthetic_circuit(log_n, nin, seed=5)
    c = pl.Circuit(n=n, num_inputs=nin, selectors=sc.selectors, sigma=sc.sigma)
    tau = 987654321
    pk = pl.preprocess(c, tau)
    w, pubs = sc.witness(11)
    tr = {}
    pl.prove(pk, w, pubs, bu.blinders(12), ext_msg=b"x", trace=tr)
    t_ref = tr["t_poly"]
    beta, gamma, alpha = tr["beta"], tr["gamma"], tr["alpha"]
    wN, _ = root_6n(log_n)
    big = 6 * n
    g = bn.FR_GENERATOR
    pts = [g * pow(wN, i, R) % R for i in range(big)]
    ev = lambda poly: [bn.poly_eval(poly, x) for x in pts]
    sel_c = [ev(p) for p in pk.selector_polys]
    sig_c = [ev(p) for p in pk.sigma_polys]
    w_c = [ev(p) for p in tr["wire_polys"]]
    z_c = ev(tr["z_poly"])
    pi_c = ev(tr["pi_poly"])
    quot = []
    for i, x in enumerate(pts):
        wv = [w_c[j][i] for j in range(5)]
        t_circ = pl.gate_eval([sel_c[s][i] for s in range(13)], wv, pi_c[i])
        a, b = z_c[i], z_c[(i + 6) % big]                      # z(omega x) sits six points further
        for j in range(5):
            a = a * ((wv[j] + beta * pl.K[j] * x + gamma) % R) % R
            b = b * ((wv[j] + beta * sig_c[j][i] + gamma) % R) % R
        l1 = alpha * alpha % R * (z_c[i] - 1) % R * pow(n * (x - 1) % R, R - 2, R) % R
        zh_inv = pow((pow(x, n, R) - 1) % R, R - 2, R)
        quot.append(((t_circ + alpha * (a - b)) % R * zh_inv + l1) % R)
    # interpolate: t_k = g^-k / N * sum_i quot_i wN^(-ik)
    inv_big = pow(big, R - 2, R)
    winv, ginv = pow(wN, R - 2, R), pow(g, R - 2, R)
    t6 = []
    for k in range(big):
        acc = 0
        for i in range(big):
            acc = (acc + quot[i] * pow(winv, i * k % big, R)) % R
        t6.append(acc * inv_big % R * pow(ginv, k, R) % R)
    deg = 5 * (n + 1) + 2
    assert t6[:deg + 1] == t_ref[:deg + 1] and t6[deg] != 0
    assert all(v == 0 for v in t6[deg + 1:]) and all(v == 0 for v in t_ref[deg + 1:])
    assert deg + 1 <= big                                     # 5n + 8 coefficients fit 6n points from n = 8 on
