"""BN254 optimal-ate pairing with Python integers (test infrastructure only; see oracle/bn254.py).

Restates the published construction used by ark-bn254 0.3.0 / the EVM precompile (EIP-197):
Fq2 = Fq[u]/(u^2+1), twist y^2 = x^3 + 3/(9+u), Fq12 = Fq[w]/(w^12 - 18 w^6 + 82) (so u = w^6 - 9),
twist map (x, y) -> (x w^2, y w^3), Miller loop over 6x+2 with x = 4965661367192848881, two Frobenius
line steps, final exponentiation (p^12 - 1)/r.  Deliberately naive (dense Fq12, affine points); it is the
cross-check for the product's host-side verifier (cap_amd/csrc/pairing.hpp) and the source of [tau]H.
Public known answers: the G2 generator below is EIP-197's; bilinearity is checked in tests/test_verify.py.
"""
from __future__ import annotations

from .bn254 import P, R, inv_mod

ATE_LOOP_COUNT = 29793968203157093288      # 6x + 2
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


# ---- Fq2 = Fq[u]/(u^2 + 1), elements (c0, c1) ---------------------------------------------------------
def f2_add(a, b):
    return ((a[0] + b[0]) % P, (a[1] + b[1]) % P)


def f2_sub(a, b):
    return ((a[0] - b[0]) % P, (a[1] - b[1]) % P)


def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % P, (a[0] * b[1] + a[1] * b[0]) % P)


def f2_scalar(a, k):
    return (a[0] * k % P, a[1] * k % P)


def f2_inv(a):
    d = inv_mod(a[0] * a[0] + a[1] * a[1], P)
    return (a[0] * d % P, (-a[1]) * d % P)


def f2_pow(a, e):
    r = (1, 0)
    while e:
        if e & 1:
            r = f2_mul(r, a)
        a = f2_mul(a, a)
        e >>= 1
    return r


def f2_conj(a):
    return (a[0], (-a[1]) % P)


XI = (9, 1)
B2 = f2_mul((3, 0), f2_inv(XI))            # twist coefficient 3 / (9 + u)


# ---- G2 on the twist, affine (None = infinity) ---------------------------------------------------------
def g2_is_on_curve(q):
    if q is None:
        return True
    x, y = q
    return f2_sub(f2_mul(y, y), f2_add(f2_mul(f2_mul(x, x), x), B2)) == (0, 0)


def g2_add(a, b):
    if a is None:
        return b
    if b is None:
        return a
    (x1, y1), (x2, y2) = a, b
    if x1 == x2:
        if f2_add(y1, y2) == (0, 0):
            return None
        m = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    y3 = f2_sub(f2_mul(m, f2_sub(x1, x3)), y1)
    return (x3, y3)


def g2_mul(q, k):
    k %= R
    acc = None
    while k:
        if k & 1:
            acc = g2_add(acc, q)
        q = g2_add(q, q)
        k >>= 1
    return acc


def g2_neg(q):
    return None if q is None else (q[0], ((-q[1][0]) % P, (-q[1][1]) % P))


# ---- Fq12 = Fq[w]/(w^12 - 18 w^6 + 82): lists of 12 coefficients ----------------------------------------
F12_ONE = [1] + [0] * 11


def f12_mul(a, b):
    t = [0] * 23
    for i, x in enumerate(a):
        if x:
            for j, y in enumerate(b):
                t[i + j] += x * y
    for k in range(22, 11, -1):          # w^k = 18 w^(k-6) - 82 w^(k-12)
        c = t[k]
        if c:
            t[k - 6] += 18 * c
            t[k - 12] -= 82 * c
    return [v % P for v in t[:12]]


def f12_pow(a, e):
    r = list(F12_ONE)
    while e:
        if e & 1:
            r = f12_mul(r, a)
        a = f12_mul(a, a)
        e >>= 1
    return r


def f12_from_f2(c, pos):
    """embed (c0 + c1 u) * w^pos, with u = w^6 - 9"""
    out = [0] * 12
    out[pos] = (c[0] - 9 * c[1]) % P
    out[pos + 6] = c[1] % P
    return out


def _line(m2, r_pt, p_pt):
    """line through the twisted image of r_pt with twist-slope m2, evaluated at P = (xp, yp) in G1:
    l = -yp + (m2 xp) w + (yr - m2 xr) w^3   (SURVEY-style derivation in cap_amd/csrc/pairing.hpp)"""
    xp, yp = p_pt
    xr, yr = r_pt
    l = [0] * 12
    l[0] = (-yp) % P
    a = f12_from_f2(f2_scalar(m2, xp), 1)
    b = f12_from_f2(f2_sub(yr, f2_mul(m2, xr)), 3)
    return [(l[i] + a[i] + b[i]) % P for i in range(12)]


def _step(r_pt, q_pt, p_pt):
    """returns (line value, r + q) for affine twist points (r == q -> tangent)"""
    (x1, y1), (x2, y2) = r_pt, q_pt
    if r_pt == q_pt:
        m = f2_mul(f2_scalar(f2_mul(x1, x1), 3), f2_inv(f2_scalar(y1, 2)))
    else:
        m = f2_mul(f2_sub(y2, y1), f2_inv(f2_sub(x2, x1)))
    x3 = f2_sub(f2_sub(f2_mul(m, m), x1), x2)
    y3 = f2_sub(f2_mul(m, f2_sub(x1, x3)), y1)
    return _line(m, r_pt, p_pt), (x3, y3)


GAMMA_X1 = f2_pow(XI, (P - 1) // 3)
GAMMA_Y1 = f2_pow(XI, (P - 1) // 2)
GAMMA_X2 = f2_pow(XI, (P * P - 1) // 3)
GAMMA_Y2 = f2_pow(XI, (P * P - 1) // 2)


def miller_loop(q_pt, p_pt):
    """f_{6x+2,Q}(P) with the two Frobenius steps; q_pt in G2 (twist, affine), p_pt in G1 (affine)"""
    if q_pt is None or p_pt is None:
        return list(F12_ONE)
    f = list(F12_ONE)
    r_pt = q_pt
    for i in range(ATE_LOOP_COUNT.bit_length() - 2, -1, -1):
        l, r_pt = _step(r_pt, r_pt, p_pt)
        f = f12_mul(f12_mul(f, f), l)
        if (ATE_LOOP_COUNT >> i) & 1:
            l, r_pt = _step(r_pt, q_pt, p_pt)
            f = f12_mul(f, l)
    q1 = (f2_mul(f2_conj(q_pt[0]), GAMMA_X1), f2_mul(f2_conj(q_pt[1]), GAMMA_Y1))
    nq2 = (f2_mul(q_pt[0], GAMMA_X2), f2_sub((0, 0), f2_mul(q_pt[1], GAMMA_Y2)))
    l, r_pt = _step(r_pt, q1, p_pt)
    f = f12_mul(f, l)
    l, _ = _step(r_pt, nq2, p_pt)
    f = f12_mul(f, l)
    return f


FINAL_EXP = (P ** 12 - 1) // R


def final_exponentiation(f):
    return f12_pow(f, FINAL_EXP)


def pairing(q_pt, p_pt):
    return final_exponentiation(miller_loop(q_pt, p_pt))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 ; pairs = [(P_i in G1, Q_i in G2), ...]"""
    f = list(F12_ONE)
    for p_pt, q_pt in pairs:
        f = f12_mul(f, miller_loop(q_pt, p_pt))
    return final_exponentiation(f) == F12_ONE
