"""Synthetic TurboPlonk workloads with the shape of CAP note circuits.

Host-side harness, the counterpart of the reference's `src/bench_utils/mod.rs`
(get_builder_transfer :317-341) and `src/utils/params_builder.rs`
(TransferParamsBuilder :288-930, TxnsParams::generate_txns :64-241).  The
reference's circuit builder (src/circuit/*.rs, Rust, out of scope per SURVEY §8)
cannot run here, and prover cost depends only on the circuit *shape*: domain
size n, 5 wire columns, 13 selector columns, the permutation and the number of
public inputs - the proving key comes from a dummy witness
(src/circuit/transfer.rs:36-49).  So the harness builds a random satisfiable
TurboPlonk circuit with the pinned shape of each note type
(src/utils/mod.rs:136-193) and random witnesses for it.

This is workload synthesis (setup, outside every timed region), written with
Python integers; it does not use oracle/ and is not part of the hot path.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

# BN254 scalar field (ark_bn254::Fr; src/config.rs:77-84 fixes the curve)
R = 0x30644E72E131A029B85045B68181585D2833E84879B9709143E1F593F0000001
MONT = (1 << 256) % R
NUM_WIRES = 5
NUM_SELECTORS = 13
Q_LC, Q_MUL, Q_HASH, Q_O, Q_C, Q_ECC = 0, 4, 6, 10, 11, 12
K = [
    1,
    0x2F8DD1F1A7583C42C4E12A44E110404C73CA6C94813F85835DA4FB7BB1301D4A,
    0x1EE678A0470A75A6EAA8FE837060498BA828A3703B311D0F77F010424AFEB025,
    0x2042A587A90C187B0A087C03E29C968B950B1DB26D5C82D666905A6895790C0A,
    0x2E2B91456103698ADF57B799969DEA1C8F739DA5D8D40DD3EB9222DB7C81E881,
]
ROOT_28 = pow(5, (R - 1) >> 28, R)

def transfer_num_public_inputs(n_in: int, n_out: int) -> int:
    """TransferPublicInput::to_scalars (src/proof/transfer.rs:443-458): merkle root, native asset code, valid_until,
    fee, one nullifier per input, one commitment per output, then the viewing memo = ElGamal ciphertext of
    [asset code, 12 values per non-fee input, 4 per non-fee output] (src/structs.rs:1300-1385; VIEWABLE_DATA_LEN = 12,
    src/constants.rs:17-28) with its 2-scalar ephemeral key.  2-in/2-out -> 27 (the value the verifier contract pins)."""
    return 4 + n_in + n_out + 2 + 1 + 12 * (n_in - 1) + 4 * (n_out - 1)


# note shapes: (log2 domain size, number of public inputs)
#   transfer 2-in/2-out: n = 2^15 at depth 10 (src/utils/mod.rs:149-153), 27 public inputs
#     (src/proof/transfer.rs:443-458); depth 26 may need 2^16 - both are benchmarked
#   mint depth 26: n = 2^14 (src/utils/mod.rs:160-165), 22 public inputs (src/proof/mint.rs:262-277)
#   freeze k inputs: n = 2^14 for k = 2 (src/utils/mod.rs:172-177), 3 + 2k public inputs (src/proof/freeze.rs:331-344)
#   the reference's largest pinned shapes, Transfer(3, 5, 26) and Freeze(5, 26), need 2^16 (src/utils/mod.rs:139-187)
NOTE_SHAPES = {
    "transfer_2x2": (15, transfer_num_public_inputs(2, 2)),
    "transfer_2x2_d26": (16, transfer_num_public_inputs(2, 2)),
    "transfer_2x3": (15, transfer_num_public_inputs(2, 3)),      # generate_txns' shape (params_builder.rs:74-76)
    "transfer_3x5_d26": (16, transfer_num_public_inputs(3, 5)),
    "mint": (14, 22),
    "freeze_2": (14, 7),
    "freeze_3": (15, 9),
    "freeze_5_d26": (16, 13),
}


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

    def field(self) -> int:
        v = 0
        for i in range(4):
            v |= self.next() << (64 * i)
        return v % R


def to_mont_array(vals) -> np.ndarray:
    """canonical ints -> (len, 4) uint64 Montgomery limbs (arkworks memory layout)."""
    b = b"".join(((v * MONT) % R).to_bytes(32, "little") for v in vals)
    return np.frombuffer(b, dtype=np.uint64).reshape(-1, 4).copy()


def to_canonical_array(vals) -> np.ndarray:
    b = b"".join((v % R).to_bytes(32, "little") for v in vals)
    return np.frombuffer(b, dtype=np.uint64).reshape(-1, 4).copy()


def from_mont_array(a: np.ndarray):
    rinv = pow(MONT, R - 2, R)
    a = np.ascontiguousarray(a, dtype=np.uint64).reshape(-1, 4)
    raw = a.tobytes()
    return [int.from_bytes(raw[32 * i:32 * i + 32], "little") * rinv % R for i in range(a.shape[0])]


@dataclass
class SyntheticCircuit:
    """Fixed circuit shape: selectors, wire -> variable map, permutation."""
    log_n: int
    num_inputs: int
    selectors: list      # 13 x n canonical ints
    wire_vars: list      # 5 x n variable ids
    num_vars: int
    free_vars: list      # variable ids assigned freely by a witness (incl. public inputs)
    pub_vars: list       # variable ids of the public inputs (rows 0..num_inputs-1, wire 4)
    gate_rows: int       # rows [num_inputs, gate_rows) define wire 4 from wires 0..3
    sigma: list          # 5 x n canonical ints: sigma_i(omega^j)

    @property
    def n(self) -> int:
        return 1 << self.log_n

    # ---- device / ABI views -------------------------------------------------------------------
    def selectors_mont(self) -> np.ndarray:
        return np.concatenate([to_mont_array(col) for col in self.selectors]).reshape(NUM_SELECTORS, self.n, 4)

    def sigma_mont(self) -> np.ndarray:
        return np.concatenate([to_mont_array(col) for col in self.sigma]).reshape(NUM_WIRES, self.n, 4)

    def witness(self, seed: int):
        """Random satisfying assignment: returns (wires 5 x n ints, pub_inputs)."""
        rng = SplitMix64(seed)
        val = [None] * self.num_vars
        val[0], val[1] = 0, 1
        for v in self.free_vars:
            val[v] = rng.field()
        sel, wv = self.selectors, self.wire_vars
        for j in range(self.num_inputs, self.gate_rows):
            w = [val[wv[i][j]] for i in range(4)]
            rest = (sel[Q_C][j]
                    + sel[Q_LC][j] * w[0] + sel[Q_LC + 1][j] * w[1] + sel[Q_LC + 2][j] * w[2] + sel[Q_LC + 3][j] * w[3]
                    + sel[Q_MUL][j] * w[0] * w[1] + sel[Q_MUL + 1][j] * w[2] * w[3])
            for i in range(4):
                if sel[Q_HASH + i][j]:
                    rest += sel[Q_HASH + i][j] * pow(w[i], 5, R)
            rest %= R
            # (q_ecc * w0 w1 w2 w3 - q_o) * w4 + rest = 0
            d = (sel[Q_O][j] - sel[Q_ECC][j] * w[0] * w[1] % R * w[2] % R * w[3]) % R
            if d == 0:
                raise RuntimeError("degenerate synthetic gate; pick another seed")
            val[wv[4][j]] = rest * pow(d, R - 2, R) % R
        wires = [[val[wv[i][j]] for j in range(self.n)] for i in range(NUM_WIRES)]
        pubs = [val[v] for v in self.pub_vars]
        return wires, pubs

    @staticmethod
    def wires_mont(wires) -> np.ndarray:
        n = len(wires[0])
        return np.concatenate([to_mont_array(col) for col in wires]).reshape(NUM_WIRES, n, 4)


def synthetic_circuit(log_n: int, num_inputs: int, seed: int = 2, fill: float = 0.94) -> SyntheticCircuit:
    """Random TurboPlonk circuit on a domain of 2^log_n rows.

    rows [0, num_inputs): IO gates (q_o = 1, wire 4 = public input);
    rows [num_inputs, gate_rows): arithmetic / Rescue-power / ECC-product gates whose inputs reuse
    earlier variables (copy constraints) or introduce fresh ones;
    remaining rows: padding on the zero variable (as jf-relation pads to the domain size).
    """
    n = 1 << log_n
    rng = SplitMix64(seed)
    gate_rows = max(num_inputs + 1, min(n - 1, int(n * fill)))
    if n <= num_inputs + 1:
        raise ValueError("domain too small for the public inputs")
    sel = [[0] * n for _ in range(NUM_SELECTORS)]
    wv = [[0] * n for _ in range(NUM_WIRES)]
    num_vars = 2            # 0 -> value 0, 1 -> value 1
    free_vars, pub_vars = [], []
    pool = [0, 1]           # variables that later gates may reuse
    for j in range(num_inputs):
        sel[Q_O][j] = 1
        v = num_vars
        num_vars += 1
        wv[4][j] = v
        free_vars.append(v)
        pub_vars.append(v)
        pool.append(v)
    for j in range(num_inputs, gate_rows):
        kind = rng.next() & 7
        for i in range(4):
            r = rng.next()
            if (r & 3) and pool:
                wv[i][j] = pool[(r >> 8) % len(pool)] if (r & 4) else pool[-1 - ((r >> 8) % min(len(pool), 16))]
            else:
                v = num_vars
                num_vars += 1
                free_vars.append(v)
                wv[i][j] = v
        sel[Q_O][j] = 1
        if kind <= 3:
            for i in range(4):
                sel[Q_LC + i][j] = rng.field()
            sel[Q_MUL][j] = rng.field()
            sel[Q_MUL + 1][j] = rng.field()
            sel[Q_C][j] = rng.field()
        elif kind <= 5:
            for i in range(4):
                sel[Q_HASH + i][j] = rng.field()
            sel[Q_LC][j] = rng.field()
            sel[Q_C][j] = rng.field()
        elif kind == 6:
            sel[Q_ECC][j] = rng.field()
            sel[Q_C][j] = rng.field()
            sel[Q_LC + 1][j] = rng.field()
        else:
            sel[Q_MUL][j] = 1
            sel[Q_LC + 2][j] = R - 1
            sel[Q_O][j] = rng.field() | 1
        v = num_vars
        num_vars += 1
        wv[4][j] = v
        pool.append(v)
    # permutation: cells of one variable form a cycle; sigma value of cell (i, j) is k_i' * omega^j'
    omega = pow(ROOT_28, 1 << (28 - log_n), R)
    om = [1] * n
    for j in range(1, n):
        om[j] = om[j - 1] * omega % R
    cells = [[] for _ in range(num_vars)]
    for i in range(NUM_WIRES):
        col = wv[i]
        for j in range(n):
            cells[col[j]].append((i, j))
    sigma = [[0] * n for _ in range(NUM_WIRES)]
    for lst in cells:
        m = len(lst)
        for t in range(m):
            i, j = lst[t]
            i2, j2 = lst[(t + 1) % m]
            sigma[i][j] = K[i2] * om[j2] % R
    return SyntheticCircuit(log_n=log_n, num_inputs=num_inputs, selectors=sel, wire_vars=wv, num_vars=num_vars,
                            free_vars=free_vars, pub_vars=pub_vars, gate_rows=gate_rows, sigma=sigma)


def note_circuit(kind: str, seed: int = 2) -> SyntheticCircuit:
    log_n, num_inputs = NOTE_SHAPES[kind]
    return synthetic_circuit(log_n, num_inputs, seed)


def weighted_scalar_sums(sc: np.ndarray, lo: int = 0):
    """(sum k_i, sum (lo + i) k_i) for canonical scalars (n, 4) uint64 - exact, numpy on 16-bit pieces.  With bases
    P_i = [a + i b] G (capgpu_srs_generate_affine_seq) the MSM must equal [a * s0 + b * s1] G: the full-size known
    answer of BASELINE config 5."""
    sc = np.ascontiguousarray(sc, dtype=np.uint64).reshape(-1, 4)
    n = sc.shape[0]
    pieces = sc.view(np.uint16).reshape(n, 16)
    s0 = s1 = 0
    blk_len = 1 << 18
    for start in range(0, n, blk_len):
        blk = pieces[start:start + blk_len].astype(np.uint64)
        idx = np.arange(start, start + blk.shape[0], dtype=np.uint64) + np.uint64(lo)
        col = blk.sum(axis=0)
        wcol = (blk * idx[:, None]).sum(axis=0)        # < 2^18 * 2^16 * 2^25: fits 64 bits
        s0 += sum(int(col[j]) << (16 * j) for j in range(16))
        s1 += sum(int(wcol[j]) << (16 * j) for j in range(16))
    return s0, s1


def random_canonical_scalars(seed: int, n: int) -> np.ndarray:
    """n uniformly random canonical scalars below 2^253 (< r), (n, 4) uint64 - numpy speed for 2^24 of them."""
    rng = np.random.default_rng(seed)
    sc = rng.integers(0, 1 << 63, size=(n, 4), dtype=np.uint64) * np.uint64(2) + \
        rng.integers(0, 2, size=(n, 4), dtype=np.uint64)
    sc[:, 3] &= np.uint64((1 << 61) - 1)
    return sc


def blinders(seed: int, count: int = 13):
    rng = SplitMix64(seed)
    return [rng.field() for _ in range(count)]
