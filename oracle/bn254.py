"""BN254 big-int oracle: Fq/Fr arithmetic, G1, naive + arkworks-style MSM, NTT.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported, linked or
executed by the product path (``cap_amd/``); only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may use it,
and there only as the checker.

PARITY UNPINNED.  The reference (`/root/reference`, Rust) holds no golden
vectors / known-answer tests for MSM, NTT, transcript or proof bytes
(SURVEY.md §8c); its arithmetic lives in third-party crates that are absent
from the tree (pinned in /root/reference/Cargo.lock):
  ark-ec 0.3.0 (Cargo.lock:103-105)  VariableBaseMSM::multi_scalar_mul
  ark-poly 0.3.0 (Cargo.lock:194-196) Radix2EvaluationDomain
  ark-ff 0.3.0 (Cargo.lock:153-155)  Fp256 Montgomery, R = 2^256
  ark-bn254 0.3.0 (Cargo.lock:81-83) curve constants
This file restates their *published* mathematics with Python integers, anchored
on the reference's call sites (src/proof/transfer.rs:181-186,
src/proof/mint.rs:113, src/proof/freeze.rs:151, src/proof/mod.rs:67) and on
public known answers that do exist outside the reference (EIP-196 G1 doubling
vector, Keccak-256 digests) - see tests/test_oracle.py.

MSM and NTT results are mathematically canonical (an affine G1 point and an
NTT output vector have exactly one value), so a big-int restatement pins the
bit pattern an arkworks build would produce for the same inputs.
"""
from __future__ import annotations

# ---------------------------------------------------------------------------
# constants (ark-bn254 0.3.0: fields/fq.rs, fields/fr.rs, curves/g1.rs)
# ---------------------------------------------------------------------------
P = 0x30644E72E131A029B85045B68181585D97816A916871CA8D3C208C16D87CFD47  # Fq
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001  # Fr
B_COEFF = 3            # y^2 = x^3 + 3
G1_GEN = (1, 2)
TWO_ADICITY = 28
FR_GENERATOR = 5       # Fr::multiplicative_generator(); coset shift of ark-poly
MONT_R = 1 << 256      # ark-ff Fp256: 4 x u64 limbs, R = 2^256
# omega_28 = 5^((r-1)/2^28)
ROOT_OF_UNITY_28 = pow(FR_GENERATOR, (R - 1) >> TWO_ADICITY, R)

INF = None             # point at infinity (affine oracle representation)


def inv_mod(a: int, m: int) -> int:
    return pow(a % m, m - 2, m)


def to_mont(a: int, m: int) -> int:
    return (a * MONT_R) % m


def from_mont(a: int, m: int) -> int:
    return (a * inv_mod(MONT_R, m)) % m


def limbs_le(a: int, n: int = 4, bits: int = 64):
    mask = (1 << bits) - 1
    return [(a >> (bits * i)) & mask for i in range(n)]


def from_limbs_le(l, bits: int = 64) -> int:
    return sum(int(v) << (bits * i) for i, v in enumerate(l))


# ---------------------------------------------------------------------------
# portable PRNG shared by oracle, C restatement and GPU harness
# ---------------------------------------------------------------------------
class SplitMix64:
    M = (1 << 64) - 1

    def __init__(self, seed: int):
        self.s = seed & self.M

    def next(self) -> int:
        self.s = (self.s + 0x9E3779B97F4A7C15) & self.M
        z = self.s
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & self.M
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & self.M
        return z ^ (z >> 31)

    def field(self, m: int) -> int:
        """4 words little-endian -> integer mod m (same rule in capref.c)."""
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % m


# ---------------------------------------------------------------------------
# G1 affine arithmetic with field inversions (independent of the projective
# formulas used by the C restatement and the HIP kernels)
# ---------------------------------------------------------------------------
def is_on_curve(pt) -> bool:
    if pt is INF:
        return True
    x, y = pt
    return (y * y - x * x * x - B_COEFF) % P == 0


def g1_neg(pt):
    if pt is INF:
        return INF
    return (pt[0], (-pt[1]) % P)


def g1_add(a, b):
    if a is INF:
        return b
    if b is INF:
        return a
    x1, y1 = a
    x2, y2 = b
    if x1 == x2:
        if (y1 + y2) % P == 0:
            return INF
        lam = (3 * x1 * x1) * inv_mod(2 * y1, P) % P
    else:
        lam = (y2 - y1) * inv_mod(x2 - x1, P) % P
    x3 = (lam * lam - x1 - x2) % P
    y3 = (lam * (x1 - x3) - y1) % P
    return (x3, y3)


def g1_mul(pt, k: int):
    k %= R
    acc = INF
    add = pt
    while k:
        if k & 1:
            acc = g1_add(acc, add)
        add = g1_add(add, add)
        k >>= 1
    return acc


# Jacobian helpers (only so big MSMs in the oracle are not dominated by modular
# inversions; checked against the affine law in tests/test_oracle.py)
def _jac_double(p):
    X, Y, Z = p
    if Z == 0:
        return p
    A = X * X % P
    Bq = Y * Y % P
    C = Bq * Bq % P
    D = 2 * ((X + Bq) * (X + Bq) - A - C) % P
    E = 3 * A % P
    F = E * E % P
    X3 = (F - 2 * D) % P
    Y3 = (E * (D - X3) - 8 * C) % P
    Z3 = 2 * Y * Z % P
    return (X3, Y3, Z3)


def _jac_add_affine(p, q):
    if q is INF:
        return p
    X1, Y1, Z1 = p
    x2, y2 = q
    if Z1 == 0:
        return (x2, y2, 1)
    Z1Z1 = Z1 * Z1 % P
    U2 = x2 * Z1Z1 % P
    S2 = y2 * Z1 * Z1Z1 % P
    if U2 == X1:
        if S2 == Y1:
            return _jac_double(p)
        return (1, 1, 0)
    H = (U2 - X1) % P
    HH = H * H % P
    I = 4 * HH % P
    J = H * I % P
    r = 2 * (S2 - Y1) % P
    V = X1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * Y1 * J) % P
    Z3 = ((Z1 + H) * (Z1 + H) - Z1Z1 - HH) % P
    return (X3, Y3, Z3)


def _jac_add(p, q):
    X1, Y1, Z1 = p
    X2, Y2, Z2 = q
    if Z1 == 0:
        return q
    if Z2 == 0:
        return p
    Z1Z1 = Z1 * Z1 % P
    Z2Z2 = Z2 * Z2 % P
    U1 = X1 * Z2Z2 % P
    U2 = X2 * Z1Z1 % P
    S1 = Y1 * Z2 * Z2Z2 % P
    S2 = Y2 * Z1 * Z1Z1 % P
    if U1 == U2:
        if S1 == S2:
            return _jac_double(p)
        return (1, 1, 0)
    H = (U2 - U1) % P
    I = 4 * H * H % P
    J = H * I % P
    r = 2 * (S2 - S1) % P
    V = U1 * I % P
    X3 = (r * r - J - 2 * V) % P
    Y3 = (r * (V - X3) - 2 * S1 * J) % P
    Z3 = ((Z1 + Z2) * (Z1 + Z2) - Z1Z1 - Z2Z2) * H % P
    return (X3, Y3, Z3)


def jac_to_affine(p):
    X, Y, Z = p
    if Z % P == 0:
        return INF
    zi = inv_mod(Z, P)
    zi2 = zi * zi % P
    return (X * zi2 % P, Y * zi2 * zi % P)


def msm_naive(bases, scalars):
    """sum_i scalars[i] * bases[i] by the definition (double-and-add each)."""
    acc = INF
    for b, k in zip(bases, scalars):
        acc = g1_add(acc, g1_mul(b, k))
    return acc


def ark_window_size(n: int) -> int:
    """ark-ec 0.3.0 msm/variable_base.rs: c = 3 if n < 32 else ln_without_floats(n) + 2,
    ln_without_floats(a) = ceil_log2(a) * 69 / 100  (SURVEY §3.2)."""
    if n < 32:
        return 3
    return ((n - 1).bit_length() * 69) // 100 + 2


def msm_pippenger(bases, scalars, c: int | None = None):
    """arkworks-style Pippenger, returns an affine point.
    scalars are canonical integers (not Montgomery)."""
    n = min(len(bases), len(scalars))
    if c is None:
        c = ark_window_size(n)
    num_bits = 254
    window_sums = []
    for w_start in range(0, num_bits, c):
        res = (1, 1, 0)
        buckets = [(1, 1, 0)] * ((1 << c) - 1)
        for i in range(n):
            k = scalars[i]
            if k == 0 or bases[i] is INF:
                continue
            if k == 1:
                if w_start == 0:
                    res = _jac_add_affine(res, bases[i])
            else:
                d = (k >> w_start) & ((1 << c) - 1)
                if d:
                    buckets[d - 1] = _jac_add_affine(buckets[d - 1], bases[i])
        running = (1, 1, 0)
        for b in reversed(buckets):
            running = _jac_add(running, b)
            res = _jac_add(res, running)
        window_sums.append(res)
    total = (1, 1, 0)
    for ws in reversed(window_sums[1:]):
        total = _jac_add(total, ws)
        for _ in range(c):
            total = _jac_double(total)
    total = _jac_add(total, window_sums[0])
    return jac_to_affine(total)


# ---------------------------------------------------------------------------
# NTT over Fr (ark-poly 0.3.0 Radix2EvaluationDomain semantics)
# ---------------------------------------------------------------------------
def root_of_unity(log_n: int) -> int:
    assert 0 <= log_n <= TWO_ADICITY
    return pow(ROOT_OF_UNITY_28, 1 << (TWO_ADICITY - log_n), R)


def dft_naive(a, omega):
    """O(n^2) definition: out[j] = sum_i a[i] * omega^(i*j)."""
    n = len(a)
    out = []
    for j in range(n):
        wj = pow(omega, j, R)
        acc = 0
        x = 1
        for i in range(n):
            acc = (acc + a[i] * x) % R
            x = x * wj % R
        out.append(acc)
    return out


def _ntt_rec(a, omega):
    n = len(a)
    if n == 1:
        return list(a)
    w2 = omega * omega % R
    ev = _ntt_rec(a[0::2], w2)
    od = _ntt_rec(a[1::2], w2)
    out = [0] * n
    x = 1
    h = n // 2
    for k in range(h):
        t = x * od[k] % R
        out[k] = (ev[k] + t) % R
        out[k + h] = (ev[k] - t) % R
        x = x * omega % R
    return out


def ntt(a, log_n: int | None = None):
    """fft: natural order in / out; input shorter than n is zero padded."""
    if log_n is None:
        log_n = (len(a) - 1).bit_length() if len(a) > 1 else 0
    n = 1 << log_n
    a = [x % R for x in a] + [0] * (n - len(a))
    return _ntt_rec(a, root_of_unity(log_n))


def intt(a, log_n: int | None = None):
    if log_n is None:
        log_n = (len(a) - 1).bit_length() if len(a) > 1 else 0
    n = 1 << log_n
    a = [x % R for x in a] + [0] * (n - len(a))
    out = _ntt_rec(a, inv_mod(root_of_unity(log_n), R))
    ninv = inv_mod(n, R)
    return [x * ninv % R for x in out]


def coset_ntt(a, log_n: int | None = None, g: int = FR_GENERATOR):
    if log_n is None:
        log_n = (len(a) - 1).bit_length() if len(a) > 1 else 0
    n = 1 << log_n
    a = [x % R for x in a] + [0] * (n - len(a))
    x = 1
    sc = []
    for v in a:
        sc.append(v * x % R)
        x = x * g % R
    return ntt(sc, log_n)


def coset_intt(a, log_n: int | None = None, g: int = FR_GENERATOR):
    out = intt(a, log_n)
    gi = inv_mod(g, R)
    x = 1
    res = []
    for v in out:
        res.append(v * x % R)
        x = x * gi % R
    return res


def poly_eval(coeffs, x: int) -> int:
    acc = 0
    for c in reversed(coeffs):
        acc = (acc * x + c) % R
    return acc


# ---------------------------------------------------------------------------
# ark-serialize 0.3 compressed G1
# ---------------------------------------------------------------------------
def g1_serialize_compressed(pt) -> bytes:
    if pt is INF:
        b = bytearray(32)
        b[31] |= 0x40
        return bytes(b)
    x, y = pt
    b = bytearray(x.to_bytes(32, "little"))
    if y > (P - y):          # SWFlags::PositiveY  => y is the larger root
        b[31] |= 0x80
    return bytes(b)


def fr_to_bytes_le(v: int) -> bytes:
    return (v % R).to_bytes(32, "little")


# ---------------------------------------------------------------------------
# Keccak-256 (original Keccak padding 0x01, as sha3 0.10.1 `Keccak256`, the hash
# behind jf-plonk's SolidityTranscript; hashlib here only has SHA3-256)
# ---------------------------------------------------------------------------
_KECCAK_RC = [
    0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000,
    0x000000000000808B, 0x0000000080000001, 0x8000000080008081, 0x8000000000008009,
    0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
    0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003,
    0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
    0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008,
]
_KECCAK_ROT = [
    [0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61],
    [28, 55, 25, 21, 56], [27, 20, 39, 8, 14],
]
_M64 = (1 << 64) - 1


def _rol(v, r):
    r %= 64
    return ((v << r) | (v >> (64 - r))) & _M64 if r else v


def _keccak_f(st):
    for rnd in range(24):
        C = [st[x][0] ^ st[x][1] ^ st[x][2] ^ st[x][3] ^ st[x][4] for x in range(5)]
        D = [C[(x - 1) % 5] ^ _rol(C[(x + 1) % 5], 1) for x in range(5)]
        st = [[st[x][y] ^ D[x] for y in range(5)] for x in range(5)]
        Bm = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                Bm[y][(2 * x + 3 * y) % 5] = _rol(st[x][y], _KECCAK_ROT[x][y])
        st = [[Bm[x][y] ^ ((~Bm[(x + 1) % 5][y]) & Bm[(x + 2) % 5][y]) for y in range(5)]
              for x in range(5)]
        st[0][0] ^= _KECCAK_RC[rnd]
    return st


def keccak256(data: bytes) -> bytes:
    rate = 136
    msg = bytearray(data)
    msg.append(0x01)
    while len(msg) % rate:
        msg.append(0x00)
    msg[-1] |= 0x80
    st = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        blk = msg[off:off + rate]
        for i in range(rate // 8):
            x, y = i % 5, i // 5
            st[x][y] ^= int.from_bytes(blk[8 * i:8 * i + 8], "little")
        st = _keccak_f(st)
    out = b""
    for i in range(4):
        x, y = i % 5, i // 5
        out += st[x][y].to_bytes(8, "little")
    return out
