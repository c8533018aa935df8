"""TEST INFRASTRUCTURE ONLY - CPU restatement of the reference's on-disk parameter formats.

Nothing under cap_amd/ may import this module; only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg use the oracle.

PARITY UNPINNED: the byte layouts below are ark-serialize 0.3 `CanonicalSerialize` (compressed) of the types the
reference stores with `store_data` (/root/reference/src/parameters.rs:560-567) and reads with `load_data`
(:570-577) / `load_srs` (/root/reference/src/proof/mod.rs:74-109).  Those types live in crates that are not in
/root/reference (ark-poly-commit @ cafc05e, jf-plonk @ bcd92b2, ark-serialize 0.3.0), and the reference ships no
parameter file (the Aztec CRS of src/proof/mod.rs:90-93 is absent), so the field order is restated from the crates'
published definitions and cannot be checked against a real blob here.

  UniversalSrs<E>        = kzg10::UniversalParams<E>:
                           powers_of_g: Vec<G1>, powers_of_gamma_g: BTreeMap<usize, G1>, h: G2, beta_h: G2,
                           neg_powers_of_h: BTreeMap<usize, G2>              (prepared_* are not written)
  ProvingKey<E>          = sigmas: Vec<DensePolynomial<Fr>>, selectors: Vec<DensePolynomial<Fr>>,
                           commit_key: Powers { powers_of_g: [G1], powers_of_gamma_g: [G1] },
                           vk: VerifyingKey<E>, plookup_pk: Option<..> (None)
  VerifyingKey<E>        = domain_size: usize, num_inputs: usize, sigma_comms: Vec<G1>, selector_comms: Vec<G1>,
                           k: Vec<Fr>, open_key: { g: G1, gamma_g: G1, h: G2, beta_h: G2 }, is_merged: bool,
                           plookup_vk: Option<..> (None)
  Transfer/Mint/FreezeProvingKey = ProvingKey ++ the note-shape trailer
                           (src/proof/transfer.rs:59-64: n_inputs usize, n_outputs usize, tree_depth u8;
                            src/proof/mint.rs:45-53: tree_depth u8; src/proof/freeze.rs:44-53: num_input usize,
                            tree_depth u8)
Encodings: usize/u64 = 8 bytes LE; Vec/BTreeMap = u64 length then the items (map: key then value); Fr/Fq = 32 bytes LE
canonical; G1 compressed = x with flags in the top two bits of the last byte (0x80: y is the larger root, 0x40:
infinity); G2 compressed = x.c0 then x.c1 with the same flags in the very last byte, "larger" in the Fq2 ordering
(c1 compared first, then c0); bool = 1 byte; Option = 1 byte tag.
"""
from __future__ import annotations

import struct

from . import bn254 as bn
from . import pairing as pr

P = bn.P
R = bn.R


class SerializationError(ValueError):
    """ark_serialize::SerializationError::InvalidData -> TxnApiError::DeserializationError (src/errors.rs:81-85)."""


# ---- primitives ------------------------------------------------------------------------------------------------
def u64(v: int) -> bytes:
    return struct.pack("<Q", v)


class Reader:
    def __init__(self, data: bytes, pos: int = 0):
        self.data = data
        self.pos = pos

    def take(self, n: int) -> bytes:
        if self.pos + n > len(self.data):
            raise SerializationError("unexpected end of input")
        b = self.data[self.pos:self.pos + n]
        self.pos += n
        return b

    def u64(self) -> int:
        return struct.unpack("<Q", self.take(8))[0]


def fq_sqrt(a: int):
    """p = 3 mod 4: the candidate root a^((p+1)/4); None if a is a non-residue."""
    a %= P
    r = pow(a, (P + 1) // 4, P)
    return r if r * r % P == a else None


def g1_deserialize_compressed(b: bytes):
    if len(b) != 32:
        raise SerializationError("G1: need 32 bytes")
    flags = b[31] & 0xC0
    x = int.from_bytes(b[:31] + bytes([b[31] & 0x3F]), "little")
    if flags == 0xC0:
        raise SerializationError("G1: both flags set")
    if flags == 0x40:
        if x != 0:
            raise SerializationError("G1: infinity with non-zero x")
        return bn.INF
    if x >= P:
        raise SerializationError("G1: x not canonical")
    y = fq_sqrt(x * x * x + 3)
    if y is None:
        raise SerializationError("G1: x is not on the curve")
    larger = max(y, P - y)
    y = larger if flags == 0x80 else P - larger
    return (x, y)          # cofactor 1: on the curve is in the group


def f2_gt(a, b) -> bool:
    """ark-ff QuadExtField ordering: c1 first, then c0."""
    return (a[1], a[0]) > (b[1], b[0])


def f2_sqrt(a):
    """Square root in Fq[u]/(u^2+1) (complex method); None if a is a non-residue."""
    a0, a1 = a[0] % P, a[1] % P
    if a1 == 0:
        r = fq_sqrt(a0)
        if r is not None:
            return (r, 0)
        r = fq_sqrt(-a0 % P)          # sqrt(a0) = u * sqrt(-a0)
        return (0, r)
    s = fq_sqrt((a0 * a0 + a1 * a1) % P)
    if s is None:
        return None
    inv2 = pow(2, P - 2, P)
    t = (a0 + s) * inv2 % P
    x0 = fq_sqrt(t)
    if x0 is None:
        t = (a0 - s) * inv2 % P
        x0 = fq_sqrt(t)
        if x0 is None:
            return None
    x1 = a1 * pow(2 * x0, P - 2, P) % P
    r = (x0, x1)
    return r if pr.f2_mul(r, r) == (a0, a1) else None


G2_B = None


def _g2_b():
    global G2_B
    if G2_B is None:
        G2_B = pr.f2_scalar(pr.f2_inv((9, 1)), 3)
    return G2_B


def g2_in_subgroup(q) -> bool:
    """[r] q = infinity (pairing.g2_mul reduces its scalar mod r, so the ladder is spelt out here)."""
    acc, k = None, R
    while k:
        if k & 1:
            acc = pr.g2_add(acc, q)
        q = pr.g2_add(q, q)
        k >>= 1
    return acc is None


def g2_serialize_compressed(q) -> bytes:
    if q is None:
        b = bytearray(64)
        b[63] |= 0x40
        return bytes(b)
    (x, y) = q
    b = bytearray(x[0].to_bytes(32, "little") + x[1].to_bytes(32, "little"))
    ny = ((-y[0]) % P, (-y[1]) % P)
    if f2_gt(y, ny):
        b[63] |= 0x80
    return bytes(b)


def g2_deserialize_compressed(b: bytes):
    if len(b) != 64:
        raise SerializationError("G2: need 64 bytes")
    flags = b[63] & 0xC0
    c0 = int.from_bytes(b[:32], "little")
    c1 = int.from_bytes(b[32:63] + bytes([b[63] & 0x3F]), "little")
    if flags == 0xC0:
        raise SerializationError("G2: both flags set")
    if flags == 0x40:
        if c0 or c1:
            raise SerializationError("G2: infinity with non-zero x")
        return None
    if c0 >= P or c1 >= P:
        raise SerializationError("G2: x not canonical")
    x = (c0, c1)
    rhs = pr.f2_add(pr.f2_mul(pr.f2_mul(x, x), x), _g2_b())
    y = f2_sqrt(rhs)
    if y is None:
        raise SerializationError("G2: x is not on the curve")
    ny = ((-y[0]) % P, (-y[1]) % P)
    big, small = (y, ny) if f2_gt(y, ny) else (ny, y)
    q = (x, big if flags == 0x80 else small)
    if not g2_in_subgroup(q):
        raise SerializationError("G2: point not in the prime-order subgroup")
    return q


def fr_vec(v) -> bytes:
    return u64(len(v)) + b"".join(bn.fr_to_bytes_le(x) for x in v)


def read_fr(rd: Reader) -> int:
    v = int.from_bytes(rd.take(32), "little")
    if v >= R:
        raise SerializationError("Fr not canonical")
    return v


def read_fr_vec(rd: Reader):
    return [read_fr(rd) for _ in range(rd.u64())]


def g1_vec(v) -> bytes:
    return u64(len(v)) + b"".join(bn.g1_serialize_compressed(p) for p in v)


def read_g1_vec(rd: Reader):
    return [g1_deserialize_compressed(rd.take(32)) for _ in range(rd.u64())]


def dense_poly(coeffs) -> bytes:
    """DensePolynomial keeps no trailing zero coefficients."""
    c = list(coeffs)
    while c and c[-1] % R == 0:
        c.pop()
    return fr_vec(c)


# ---- UniversalSrs --------------------------------------------------------------------------------------------
def serialize_universal_params(powers_of_g, powers_of_gamma_g: dict, h, beta_h, neg_powers_of_h: dict) -> bytes:
    out = g1_vec(powers_of_g)
    out += u64(len(powers_of_gamma_g))
    for k in sorted(powers_of_gamma_g):
        out += u64(k) + bn.g1_serialize_compressed(powers_of_gamma_g[k])
    out += g2_serialize_compressed(h) + g2_serialize_compressed(beta_h)
    out += u64(len(neg_powers_of_h))
    for k in sorted(neg_powers_of_h):
        out += u64(k) + g2_serialize_compressed(neg_powers_of_h[k])
    return out


def deserialize_universal_params(data: bytes):
    rd = Reader(data)
    powers = read_g1_vec(rd)
    gamma = {}
    for _ in range(rd.u64()):
        k = rd.u64()
        gamma[k] = g1_deserialize_compressed(rd.take(32))
    h = g2_deserialize_compressed(rd.take(64))
    beta_h = g2_deserialize_compressed(rd.take(64))
    neg = {}
    for _ in range(rd.u64()):
        k = rd.u64()
        neg[k] = g2_deserialize_compressed(rd.take(64))
    return dict(powers_of_g=powers, powers_of_gamma_g=gamma, h=h, beta_h=beta_h, neg_powers_of_h=neg, consumed=rd.pos)


# ---- VerifyingKey / ProvingKey ----------------------------------------------------------------------------------
def serialize_verifying_key(n: int, num_inputs: int, sigma_comms, selector_comms, k, g, gamma_g, h, beta_h) -> bytes:
    out = u64(n) + u64(num_inputs) + g1_vec(sigma_comms) + g1_vec(selector_comms) + fr_vec(k)
    out += bn.g1_serialize_compressed(g) + bn.g1_serialize_compressed(gamma_g)
    out += g2_serialize_compressed(h) + g2_serialize_compressed(beta_h)
    out += b"\x00"      # is_merged = false
    out += b"\x00"      # plookup_vk = None
    return out


def read_verifying_key(rd: Reader):
    vk = dict(domain_size=rd.u64(), num_inputs=rd.u64())
    vk["sigma_comms"] = read_g1_vec(rd)
    vk["selector_comms"] = read_g1_vec(rd)
    vk["k"] = read_fr_vec(rd)
    vk["g"] = g1_deserialize_compressed(rd.take(32))
    vk["gamma_g"] = g1_deserialize_compressed(rd.take(32))
    vk["h"] = g2_deserialize_compressed(rd.take(64))
    vk["beta_h"] = g2_deserialize_compressed(rd.take(64))
    if rd.take(1) != b"\x00":
        raise SerializationError("merged verifying keys are not supported")
    if rd.take(1) != b"\x00":
        raise SerializationError("plookup verifying keys are not supported")
    return vk


def serialize_proving_key(sigma_polys, selector_polys, commit_powers, vk_bytes: bytes, gamma_powers=()) -> bytes:
    out = u64(len(sigma_polys)) + b"".join(dense_poly(p) for p in sigma_polys)
    out += u64(len(selector_polys)) + b"".join(dense_poly(p) for p in selector_polys)
    out += g1_vec(commit_powers) + g1_vec(list(gamma_powers))
    out += vk_bytes
    out += b"\x00"      # plookup_pk = None
    return out


def deserialize_proving_key(data: bytes):
    rd = Reader(data)
    sigmas = [read_fr_vec(rd) for _ in range(rd.u64())]
    selectors = [read_fr_vec(rd) for _ in range(rd.u64())]
    powers = read_g1_vec(rd)
    gamma = read_g1_vec(rd)
    vk = read_verifying_key(rd)
    if rd.take(1) != b"\x00":
        raise SerializationError("plookup proving keys are not supported")
    return dict(sigmas=sigmas, selectors=selectors, powers_of_g=powers, powers_of_gamma_g=gamma, vk=vk,
                consumed=rd.pos)
