"""TurboPlonk prover + verifier restatement with Python integers (small domains).

TEST INFRASTRUCTURE ONLY (see oracle/bn254.py).  PARITY UNPINNED: this follows
the algorithm of `jf_plonk::proof_system::PlonkKzgSnark::{preprocess, prove,
verify}` at git bcd92b2 (/root/reference/Cargo.lock:992-994) as recalled in
SURVEY.md Appendix A - the crate's source is not in /root/reference, and the
reference's own tests (src/proof/transfer.rs:599-760, mint.rs:344-471,
freeze.rs:429-534) only round-trip prove -> verify.  What *is* checkable here is
checked: the proof verifies under the standard PLONK verifier equations (KZG
pairing checks done in G1 with the known trapdoor tau of the synthetic SRS), and
corrupted public inputs / proofs / keys are rejected like the reference's tests
demand.  Call sites restated: src/proof/transfer.rs:133 (preprocess), :181-186
(prove, with SolidityTranscript and ext_msg), :202-207 (verify).

Conventions: all field values are canonical Python ints mod R.  Commitments are
affine points (x, y) or None.  KZG commit(f) = [f(tau)] G, which equals
MSM(powers_of_tau_G, coeffs(f)) (bn254.msm_* are tested separately).
"""
from __future__ import annotations

from dataclasses import dataclass, field

from . import bn254 as bn
from .bn254 import R, inv_mod

NUM_WIRES = 5
NUM_SELECTORS = 13   # q_lc x4, q_mul x2, q_hash x4, q_o, q_c, q_ecc  
Q_LC, Q_MUL, Q_HASH, Q_O, Q_C, Q_ECC = 0, 4, 6, 10, 11, 12

# coset representatives k_i for BN254 (also hard-coded in CAPE's verifier contract)
K = [
    1,
    0x2F8DD1F1A7583C42C4E12A44E110404C73CA6C94813F85835DA4FB7BB1301D4A,
    0x1EE678A0470A75A6EAA8FE837060498BA828A3703B311D0F77F010424AFEB025,
    0x2042A587A90C187B0A087C03E29C968B950B1DB26D5C82D666905A6895790C0A,
    0x2E2B91456103698ADF57B799969DEA1C8F739DA5D8D40DD3EB9222DB7C81E881,
]


@dataclass
class Circuit:
    n: int
    num_inputs: int
    selectors: list          # 13 x n
    sigma: list              # 5 x n : sigma_i(omega^j) as field elements
    wires: list = field(default_factory=list)       # 5 x n
    pub_inputs: list = field(default_factory=list)  # num_inputs


@dataclass
class ProvingKeyOracle:
    n: int
    num_inputs: int
    selector_polys: list     # coefficient form
    sigma_polys: list
    sigma_evals: list
    selector_comms: list
    sigma_comms: list
    tau: int


@dataclass
class Proof:
    wires_poly_comms: list
    prod_perm_poly_comm: object
    split_quot_poly_comms: list
    opening_proof: object
    shifted_opening_proof: object
    wires_evals: list
    wire_sigma_evals: list
    perm_next_eval: int


class PlonkError(Exception):
    pass


# ---------------------------------------------------------------------------------------------
# SolidityTranscript (Keccak-256, labels ignored, state = 64 bytes)
# ---------------------------------------------------------------------------------------------
class SolidityTranscript:
    def __init__(self):
        self.state = bytes(64)
        self.buf = b""

    def append_message(self, msg: bytes):
        self.buf += msg

    def append_fr(self, v: int):
        self.buf += bn.fr_to_bytes_le(v)

    def append_commitment(self, pt):
        self.buf += bn.g1_serialize_compressed(pt)

    def append_vk_and_pub_input(self, n, num_inputs, selector_comms, sigma_comms, pub_inputs):
        self.buf += (254).to_bytes(8, "little")
        self.buf += int(n).to_bytes(8, "little")
        self.buf += int(num_inputs).to_bytes(8, "little")
        for k in K:
            self.append_fr(k)
        for c in selector_comms:
            self.append_commitment(c)
        for c in sigma_comms:
            self.append_commitment(c)
        for p in pub_inputs:
            self.append_fr(p)

    def get_and_append_challenge(self) -> int:
        h0 = bn.keccak256(self.state + self.buf + b"\x00")
        h1 = bn.keccak256(self.state + self.buf + b"\x01")
        self.state = h0 + h1
        return int.from_bytes(self.state[:48], "little") % R


# ---------------------------------------------------------------------------------------------
# helpers
# ---------------------------------------------------------------------------------------------
def log2(n: int) -> int:
    assert n & (n - 1) == 0 and n > 0
    return n.bit_length() - 1


def commit(coeffs, tau: int):
    return bn.g1_mul(bn.G1_GEN, bn.poly_eval(coeffs, tau))


def poly_add_scaled(acc, poly, s):
    if len(acc) < len(poly):
        acc.extend([0] * (len(poly) - len(acc)))
    for i, c in enumerate(poly):
        acc[i] = (acc[i] + s * c) % R


def divide_by_linear(f, a):
    """quotient of f(X) / (X - a), remainder dropped."""
    d = len(f) - 1
    q = [0] * d
    carry = 0
    for k in range(d, 0, -1):
        carry = (f[k] + a * carry) % R
        q[k - 1] = carry
    return q


def mask(coeffs, n, blind):
    """coeffs + (b0 + b1 X + ...) * (X^n - 1)   (jf-plonk mask_polynomial)"""
    out = list(coeffs) + [0] * (n + len(blind) - len(coeffs))
    for i, b in enumerate(blind):
        out[i] = (out[i] - b) % R
        out[n + i] = (out[n + i] + b) % R
    return out


def gate_eval(q, w, pi):
    """constraint of spec eq. (1) at one point; q = 13 selector values, w = 5 wire values"""
    w5 = [pow(x, 5, R) for x in w[:4]]
    return (q[Q_C] + pi
            + q[Q_LC] * w[0] + q[Q_LC + 1] * w[1] + q[Q_LC + 2] * w[2] + q[Q_LC + 3] * w[3]
            + q[Q_MUL] * w[0] * w[1] + q[Q_MUL + 1] * w[2] * w[3]
            + q[Q_HASH] * w5[0] + q[Q_HASH + 1] * w5[1] + q[Q_HASH + 2] * w5[2] + q[Q_HASH + 3] * w5[3]
            + q[Q_ECC] * w[0] * w[1] * w[2] * w[3] * w[4]
            - q[Q_O] * w[4]) % R


def check_circuit_satisfiability(c: Circuit):
    """gates + copy constraints (the reference runs this before proving: src/proof/transfer.rs:169-177)"""
    n = c.n
    omega = bn.root_of_unity(log2(n))
    for j in range(n):
        pi = c.pub_inputs[j] if j < c.num_inputs else 0
        if gate_eval([c.selectors[s][j] for s in range(NUM_SELECTORS)], [c.wires[i][j] for i in range(NUM_WIRES)], pi):
            raise PlonkError(f"gate {j} not satisfied")
    pos = {}
    x = 1
    for j in range(n):
        for i in range(NUM_WIRES):
            pos[K[i] * x % R] = (i, j)
        x = x * omega % R
    for i in range(NUM_WIRES):
        for j in range(n):
            i2, j2 = pos[c.sigma[i][j]]
            if c.wires[i][j] != c.wires[i2][j2]:
                raise PlonkError(f"copy constraint ({i},{j}) -> ({i2},{j2}) violated")


# ---------------------------------------------------------------------------------------------
# preprocess / prove / verify
# ---------------------------------------------------------------------------------------------
def preprocess(c: Circuit, tau: int) -> ProvingKeyOracle:
    ln = log2(c.n)
    sel_polys = [bn.intt(col, ln) for col in c.selectors]
    sig_polys = [bn.intt(col, ln) for col in c.sigma]
    return ProvingKeyOracle(
        n=c.n, num_inputs=c.num_inputs, selector_polys=sel_polys, sigma_polys=sig_polys,
        sigma_evals=[list(col) for col in c.sigma],
        selector_comms=[commit(p, tau) for p in sel_polys], sigma_comms=[commit(p, tau) for p in sig_polys], tau=tau)


def prove(pk: ProvingKeyOracle, wires, pub_inputs, blinders, ext_msg: bytes | None = None, trace: dict | None = None):
    """blinders: 13 field elements in jf-plonk's draw order (2 per wire poly, then 3 for z)."""
    n, ln = pk.n, log2(pk.n)
    m = 8 * n
    lm = ln + 3
    tau = pk.tau
    omega = bn.root_of_unity(ln)
    assert len(blinders) == 13 and len(pub_inputs) == pk.num_inputs

    tr = SolidityTranscript()
    if ext_msg is not None:
        tr.append_message(ext_msg)
    tr.append_vk_and_pub_input(n, pk.num_inputs, pk.selector_comms, pk.sigma_comms, pub_inputs)

    # ---- round 1
    wire_polys = [mask(bn.intt(wires[i], ln), n, blinders[2 * i:2 * i + 2]) for i in range(NUM_WIRES)]
    pi_evals = list(pub_inputs) + [0] * (n - len(pub_inputs))
    pi_poly = bn.intt(pi_evals, ln)
    wire_comms = [commit(p, tau) for p in wire_polys]
    for cm in wire_comms:
        tr.append_commitment(cm)
    _tau_plookup = tr.get_and_append_challenge()   # drawn by jf-plonk even without lookup gates

    # ---- round 2
    beta = tr.get_and_append_challenge()
    gamma = tr.get_and_append_challenge()
    z_evals = [1]
    x = 1
    for j in range(n - 1):
        num = den = 1
        for i in range(NUM_WIRES):
            num = num * ((wires[i][j] + beta * K[i] * x + gamma) % R) % R
            den = den * ((wires[i][j] + beta * pk.sigma_evals[i][j] + gamma) % R) % R
        z_evals.append(z_evals[-1] * num % R * inv_mod(den, R) % R)
        x = x * omega % R
    z_poly = mask(bn.intt(z_evals, ln), n, blinders[10:13])
    z_comm = commit(z_poly, tau)
    tr.append_commitment(z_comm)

    # ---- round 3
    alpha = tr.get_and_append_challenge()
    sel_c = [bn.coset_ntt(p, lm) for p in pk.selector_polys]
    sig_c = [bn.coset_ntt(p, lm) for p in pk.sigma_polys]
    w_c = [bn.coset_ntt(p, lm) for p in wire_polys]
    z_c = bn.coset_ntt(z_poly, lm)
    pi_c = bn.coset_ntt(pi_poly, lm)
    omega_m = bn.root_of_unity(lm)
    g = bn.FR_GENERATOR
    n_inv_dummy = None
    quot = []
    xp = g
    alpha2 = alpha * alpha % R
    for i in range(m):
        w = [w_c[j][i] for j in range(NUM_WIRES)]
        q = [sel_c[s][i] for s in range(NUM_SELECTORS)]
        t_circ = gate_eval(q, w, pi_c[i])
        zx, zwx = z_c[i], z_c[(i + 8) % m]
        a = zx
        b = zwx
        for j in range(NUM_WIRES):
            a = a * ((w[j] + beta * K[j] * xp + gamma) % R) % R
            b = b * ((w[j] + beta * sig_c[j][i] + gamma) % R) % R
        t_perm1 = alpha * (a - b) % R
        t_perm2 = alpha2 * (zx - 1) % R * inv_mod(n * (xp - 1), R) % R
        zh_inv = inv_mod(pow(xp, n, R) - 1, R)
        quot.append(((t_circ + t_perm1) * zh_inv + t_perm2) % R)
        xp = xp * omega_m % R
    t_poly = bn.coset_intt(quot, lm)
    deg = max((i for i, cf in enumerate(t_poly) if cf), default=-1)
    if deg != NUM_WIRES * (n + 1) + 2:
        raise PlonkError(f"quotient degree {deg} != {NUM_WIRES * (n + 1) + 2} (circuit not satisfied?)")
    split = []
    for i in range(NUM_WIRES):
        end = (i + 1) * (n + 2) if i < NUM_WIRES - 1 else deg + 1
        split.append(t_poly[i * (n + 2):end])
    split_comms = [commit(p, tau) for p in split]
    for cm in split_comms:
        tr.append_commitment(cm)

    # ---- round 4
    zeta = tr.get_and_append_challenge()
    w_ev = [bn.poly_eval(p, zeta) for p in wire_polys]
    s_ev = [bn.poly_eval(p, zeta) for p in pk.sigma_polys[:NUM_WIRES - 1]]
    z_next = bn.poly_eval(z_poly, zeta * omega % R)
    for e in w_ev + s_ev + [z_next]:
        tr.append_fr(e)

    # ---- linearisation polynomial (no constant term)
    zh_zeta = (pow(zeta, n, R) - 1) % R
    l1_zeta = zh_zeta * inv_mod(n * (zeta - 1), R) % R
    lin = [0]
    sp = pk.selector_polys
    for j in range(4):
        poly_add_scaled(lin, sp[Q_LC + j], w_ev[j])
    poly_add_scaled(lin, sp[Q_MUL], w_ev[0] * w_ev[1] % R)
    poly_add_scaled(lin, sp[Q_MUL + 1], w_ev[2] * w_ev[3] % R)
    for j in range(4):
        poly_add_scaled(lin, sp[Q_HASH + j], pow(w_ev[j], 5, R))
    poly_add_scaled(lin, sp[Q_O], (-w_ev[4]) % R)
    poly_add_scaled(lin, sp[Q_C], 1)
    poly_add_scaled(lin, sp[Q_ECC], w_ev[0] * w_ev[1] % R * w_ev[2] % R * w_ev[3] % R * w_ev[4] % R)
    cz = alpha
    for j in range(NUM_WIRES):
        cz = cz * ((w_ev[j] + beta * K[j] * zeta + gamma) % R) % R
    cz = (cz + alpha2 * l1_zeta) % R
    poly_add_scaled(lin, z_poly, cz)
    cs = alpha * beta % R * z_next % R
    for j in range(NUM_WIRES - 1):
        cs = cs * ((w_ev[j] + beta * s_ev[j] + gamma) % R) % R
    poly_add_scaled(lin, pk.sigma_polys[NUM_WIRES - 1], (-cs) % R)
    zp = pow(zeta, n + 2, R)
    cq = (-zh_zeta) % R
    for j in range(NUM_WIRES):
        poly_add_scaled(lin, split[j], cq)
        cq = cq * zp % R

    # ---- round 5
    v = tr.get_and_append_challenge()
    batch = list(lin)
    cf = v
    for p in wire_polys + pk.sigma_polys[:NUM_WIRES - 1]:
        poly_add_scaled(batch, p, cf)
        cf = cf * v % R
    open_poly = divide_by_linear(batch, zeta)
    shifted_poly = divide_by_linear(z_poly, zeta * omega % R)
    proof = Proof(wire_comms, z_comm, split_comms, commit(open_poly, tau), commit(shifted_poly, tau),
                  w_ev, s_ev, z_next)
    if trace is not None:
        trace.update(dict(beta=beta, gamma=gamma, alpha=alpha, zeta=zeta, v=v, wire_polys=wire_polys, z_poly=z_poly,
                          t_poly=t_poly, split=split, lin=lin, open_poly=open_poly, shifted_poly=shifted_poly,
                          z_evals=z_evals, pi_poly=pi_poly))
    return proof


def verify(n, num_inputs, selector_comms, sigma_comms, pub_inputs, proof: Proof, tau: int,
           ext_msg: bytes | None = None) -> bool:
    """Standard PLONK verifier; the pairing check e(A,[tau]H) = e(B,H) is done as tau*A == B in G1
    (tau is known for the synthetic SRS - SURVEY §8c.5)."""
    ab = verifier_pairing_inputs(n, num_inputs, selector_comms, sigma_comms, pub_inputs, proof, ext_msg)
    if ab is None:
        return False
    a_pt, b_pt = ab
    return bn.g1_mul(a_pt, tau) == b_pt


def verify_pairing(n, num_inputs, selector_comms, sigma_comms, pub_inputs, proof: Proof, h_g2, beta_h_g2,
                   ext_msg: bytes | None = None) -> bool:
    """The same verifier with the real check e(A, [tau]H) * e(-B, H) == 1 (oracle/pairing.py)."""
    from . import pairing as pr
    ab = verifier_pairing_inputs(n, num_inputs, selector_comms, sigma_comms, pub_inputs, proof, ext_msg)
    if ab is None:
        return False
    a_pt, b_pt = ab
    return pr.pairing_product_is_one([(a_pt, beta_h_g2), (bn.g1_neg(b_pt), h_g2)])


def verifier_pairing_inputs(n, num_inputs, selector_comms, sigma_comms, pub_inputs, proof: Proof,
                            ext_msg: bytes | None = None):
    """-> (A, B) with the proof valid iff e(A, [tau]H) == e(B, H); None for malformed input."""
    ln = log2(n)
    omega = bn.root_of_unity(ln)
    if len(pub_inputs) != num_inputs:
        return None
    tr = SolidityTranscript()
    if ext_msg is not None:
        tr.append_message(ext_msg)
    tr.append_vk_and_pub_input(n, num_inputs, selector_comms, sigma_comms, pub_inputs)
    for cm in proof.wires_poly_comms:
        tr.append_commitment(cm)
    tr.get_and_append_challenge()
    beta = tr.get_and_append_challenge()
    gamma = tr.get_and_append_challenge()
    tr.append_commitment(proof.prod_perm_poly_comm)
    alpha = tr.get_and_append_challenge()
    for cm in proof.split_quot_poly_comms:
        tr.append_commitment(cm)
    zeta = tr.get_and_append_challenge()
    for e in proof.wires_evals + proof.wire_sigma_evals + [proof.perm_next_eval]:
        tr.append_fr(e)
    v = tr.get_and_append_challenge()
    tr.append_commitment(proof.opening_proof)
    tr.append_commitment(proof.shifted_opening_proof)
    u = tr.get_and_append_challenge()

    w_ev, s_ev, z_next = proof.wires_evals, proof.wire_sigma_evals, proof.perm_next_eval
    alpha2 = alpha * alpha % R
    zh = (pow(zeta, n, R) - 1) % R
    if zh == 0 or zeta == 1:
        return None
    l1 = zh * inv_mod(n * (zeta - 1), R) % R
    pi = 0
    x = 1
    for p in pub_inputs:
        pi = (pi + p * zh % R * x % R * inv_mod(n * (zeta - x), R)) % R
        x = x * omega % R
    # constant term r0
    t = alpha * z_next % R * ((w_ev[4] + gamma) % R) % R
    for j in range(NUM_WIRES - 1):
        t = t * ((w_ev[j] + beta * s_ev[j] + gamma) % R) % R
    r0 = (pi - alpha2 * l1 - t) % R

    def smul(pt, s):
        return bn.g1_mul(pt, s % R)

    acc = bn.INF
    sc = selector_comms
    for j in range(4):
        acc = bn.g1_add(acc, smul(sc[Q_LC + j], w_ev[j]))
    acc = bn.g1_add(acc, smul(sc[Q_MUL], w_ev[0] * w_ev[1]))
    acc = bn.g1_add(acc, smul(sc[Q_MUL + 1], w_ev[2] * w_ev[3]))
    for j in range(4):
        acc = bn.g1_add(acc, smul(sc[Q_HASH + j], pow(w_ev[j], 5, R)))
    acc = bn.g1_add(acc, smul(sc[Q_O], -w_ev[4]))
    acc = bn.g1_add(acc, sc[Q_C])
    acc = bn.g1_add(acc, smul(sc[Q_ECC], w_ev[0] * w_ev[1] % R * w_ev[2] % R * w_ev[3] % R * w_ev[4]))
    cz = alpha
    for j in range(NUM_WIRES):
        cz = cz * ((w_ev[j] + beta * K[j] * zeta + gamma) % R) % R
    cz = (cz + alpha2 * l1) % R
    acc = bn.g1_add(acc, smul(proof.prod_perm_poly_comm, cz))
    cs = alpha * beta % R * z_next % R
    for j in range(NUM_WIRES - 1):
        cs = cs * ((w_ev[j] + beta * s_ev[j] + gamma) % R) % R
    acc = bn.g1_add(acc, smul(sigma_comms[NUM_WIRES - 1], -cs))
    zp = pow(zeta, n + 2, R)
    cq = (-zh) % R
    for j in range(NUM_WIRES):
        acc = bn.g1_add(acc, smul(proof.split_quot_poly_comms[j], cq))
        cq = cq * zp % R
    # batch the openings at zeta
    F = acc
    E = (-r0) % R
    cf = v
    for cm, ev in zip(proof.wires_poly_comms + sigma_comms[:NUM_WIRES - 1], w_ev + s_ev):
        F = bn.g1_add(F, smul(cm, cf))
        E = (E + cf * ev) % R
        cf = cf * v % R
    # shifted opening folded in with u
    F = bn.g1_add(F, smul(proof.prod_perm_poly_comm, u))
    E = (E + u * z_next) % R
    lhs_pt = bn.g1_add(proof.opening_proof, smul(proof.shifted_opening_proof, u))
    rhs = bn.g1_add(smul(proof.opening_proof, zeta), smul(proof.shifted_opening_proof, u * zeta % R * omega))
    rhs = bn.g1_add(rhs, F)
    rhs = bn.g1_add(rhs, bn.g1_neg(smul(bn.G1_GEN, E)))
    return lhs_pt, rhs
