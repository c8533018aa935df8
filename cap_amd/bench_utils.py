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

import ctypes
import os
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
VAR_UNIFORM, VAR_BOOL, VAR_U64 = 0, 1, 2     # how a free variable's value is drawn (witgen.c uses the same numbers)

_WITGEN = None


def _witgen():
    """cap_amd/libcapwitgen.so (cap_amd/csrc/witgen.c; built by `make -C cap_amd/csrc`): harness code, not the product."""
    global _WITGEN
    if _WITGEN is None:
        here = os.path.dirname(os.path.abspath(__file__))
        path = os.path.join(here, "libcapwitgen.so")
        if not os.path.exists(path):          # (normally built by __graft_entry__.build(); one C file, no dependencies)
            import subprocess
            subprocess.check_call([os.environ.get("CC", "cc"), "-O2", "-fPIC", "-shared", "-pthread", "-o", path,
                                   os.path.join(here, "csrc", "witgen.c")])
        L = ctypes.CDLL(path)
        vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
        L.capwit_fill_many.argtypes = [u32, u32, u32, u32, vp, vp, vp, vp, u32, vp, u32, u32, ctypes.c_int, vp, vp, vp]
        L.capwit_fill_many.restype = ctypes.c_int
        L.capwit_value_classes.argtypes = [vp, u64, vp]
        L.capwit_value_classes.restype = None
        L.capwit_msm_work.argtypes = [vp, u64, u32, vp, vp]
        L.capwit_msm_work.restype = None
        _WITGEN = L
    return _WITGEN


def _vp(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def msm_work(scalars_mont: np.ndarray, c: int = 15):
    """(bucket-list entries, non-empty buckets) of ONE MSM on these scalars with signed base-2^c digits on a shared bucket
    set - msm.hip's digit rule; mixed additions of its accumulation = entries - non-empty buckets"""
    a = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    e, b = np.zeros(1, np.uint64), np.zeros(1, np.uint64)
    _witgen().capwit_msm_work(_vp(a), a.shape[0], c, _vp(e), _vp(b))
    return int(e[0]), int(b[0])


def value_classes(wires_mont: np.ndarray) -> dict:
    """Shares of zeros / ones / other values below 2^64 / full-width values in a table of Montgomery field elements -
    what decides how many non-zero digits a commitment taken from evaluations has."""
    a = np.ascontiguousarray(wires_mont, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(4, np.uint64)
    _witgen().capwit_value_classes(_vp(a), a.shape[0], _vp(out))
    tot = float(a.shape[0])
    return {"zero": out[0] / tot, "one": out[1] / tot, "below_2^64": out[2] / tot, "full_width": out[3] / tot, "cells": int(tot)}

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
    gate_rows: int       # rows [num_inputs, gate_rows): a row whose wire-4 variable has no value yet defines it from
                         # wires 0..3; any other row is a constraint between variables that already hold values
    sigma: list          # 5 x n canonical ints: sigma_i(omega^j)
    free_class: list = None   # per free variable: VAR_UNIFORM / VAR_BOOL / VAR_U64 (None: all uniform)
    gadget_rows: dict = None  # cap_like_circuit: rows per gadget family (documentation of the shape)

    @property
    def n(self) -> int:
        return 1 << self.log_n

    # ---- device / ABI views -------------------------------------------------------------------
    def selectors_mont(self) -> np.ndarray:
        return np.concatenate([to_mont_array(col) for col in self.selectors]).reshape(NUM_SELECTORS, self.n, 4)

    def sigma_mont(self) -> np.ndarray:
        return np.concatenate([to_mont_array(col) for col in self.sigma]).reshape(NUM_WIRES, self.n, 4)

    def witness(self, seed: int):
        """Satisfying assignment drawn from `seed`: returns (wires 5 x n ints, pub_inputs).  Pure Python - the definition;
        `witnesses_mont` computes the same values in C (cap_amd/csrc/witgen.c)."""
        rng = SplitMix64(seed)
        val = [None] * self.num_vars
        val[0], val[1] = 0, 1
        cls = self.free_class
        for k, v in enumerate(self.free_vars):
            c = cls[k] if cls else VAR_UNIFORM
            val[v] = (rng.next() & 1) if c == VAR_BOOL else rng.next() if c == VAR_U64 else rng.field()
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
            out = wv[4][j]
            if val[out] is None:
                if d == 0:
                    raise RuntimeError("degenerate synthetic gate; pick another seed")
                val[out] = rest * pow(d, R - 2, R) % R
            elif d * val[out] % R != rest:
                raise RuntimeError(f"constraint row {j} does not hold")
        wires = [[val[wv[i][j]] for j in range(self.n)] for i in range(NUM_WIRES)]
        pubs = [val[v] for v in self.pub_vars]
        return wires, pubs

    def witnesses_mont(self, seeds, threads: int | None = None, verify: bool = False):
        """len(seeds) witnesses at C speed: (wires (count, 5, n, 4) uint64 Montgomery, pubs (count, num_inputs, 4)).
        Same values as `witness(seed)` for every seed (tests/test_bench_utils.py)."""
        L = _witgen()
        seeds = np.ascontiguousarray(seeds, dtype=np.uint64)
        cnt = int(seeds.shape[0])
        cache = self.__dict__.setdefault("_c_tables", None)
        if cache is None:
            cache = (np.ascontiguousarray(self.selectors_mont()),
                     np.ascontiguousarray(np.array(self.wire_vars, dtype=np.int32)),
                     np.ascontiguousarray(np.array(self.free_vars, dtype=np.int32)),
                     np.ascontiguousarray(np.array(self.free_class if self.free_class else [VAR_UNIFORM] * len(self.free_vars),
                                                   dtype=np.uint8)))
            self.__dict__["_c_tables"] = cache
        selm, wv, fv, fc = cache
        wires = np.empty((cnt, NUM_WIRES, self.n, 4), np.uint64)
        pubs = np.empty((cnt, self.num_inputs, 4), np.uint64)
        rcs = np.zeros(cnt, np.int32)
        thr = threads if threads else max(1, min(64, os.cpu_count() or 1))
        bad = L.capwit_fill_many(self.n, self.num_inputs, self.gate_rows, self.num_vars, _vp(selm), _vp(wv), _vp(fv), _vp(fc),
                                 len(self.free_vars), _vp(seeds), cnt, thr, 1 if verify else 0, _vp(wires), _vp(pubs), _vp(rcs))
        if bad:
            i = int(np.nonzero(rcs)[0][0])
            raise RuntimeError(f"witness {i} (seed {int(seeds[i])}): " + ("degenerate gate" if rcs[i] < 0 else "violated constraint")
                               + f" at row {abs(int(rcs[i])) - 1}")
        return wires, pubs

    @staticmethod
    def wires_mont(wires) -> np.ndarray:
        n = len(wires[0])
        return np.concatenate([to_mont_array(col) for col in wires]).reshape(NUM_WIRES, n, 4)


def synthetic_circuit(log_n: int, num_inputs: int, seed: int = 2, fill: float = 0.94, skew: float = 0.0) -> SyntheticCircuit:
    """Random TurboPlonk circuit on a domain of 2^log_n rows.

    skew > 0: that share of the free variables is drawn as booleans (3 in 4) or 64-bit values (1 in 4) instead of
    uniformly - witnesses with heavy buckets and sparse digits, for the fuzzers of the commitments taken from evaluations.

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
    sigma = _permutation(wv, num_vars, log_n)
    free_class = None
    if skew > 0:
        crng = SplitMix64(seed ^ 0x5EED)
        free_class = []
        for _ in free_vars:
            r = crng.next()
            free_class.append((VAR_BOOL if (r >> 20) & 3 else VAR_U64) if (r & 0xFFFFF) < skew * (1 << 20) else VAR_UNIFORM)
    return SyntheticCircuit(log_n=log_n, num_inputs=num_inputs, selectors=sel, wire_vars=wv, num_vars=num_vars,
                            free_vars=free_vars, pub_vars=pub_vars, gate_rows=gate_rows, sigma=sigma, free_class=free_class)


def _permutation(wv, num_vars: int, log_n: int):
    """sigma_i(omega^j) for the copy constraints of a wire -> variable map: the cells of one variable form a cycle"""
    n = 1 << log_n
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
    return sigma


class _GadgetBuilder:
    """Rows of a TurboPlonk circuit from gadget macros shaped like jf-relation's (the crate the reference's circuits are
    written with, Cargo.toml:33; absent here): what matters for the prover is which cells hold 0, 1, small and full-width
    values, so every macro reproduces the WIRE PATTERN of its counterpart - not its constants."""

    def __init__(self, log_n: int, num_inputs: int, seed: int):
        self.n = 1 << log_n
        self.rng = SplitMix64(seed)
        self.sel = [[0] * self.n for _ in range(NUM_SELECTORS)]
        self.wv = [[0] * self.n for _ in range(NUM_WIRES)]
        self.num_vars = 2
        self.free_vars, self.free_class, self.pub_vars = [], [], []
        self.j = 0
        self.family = {}
        self._fam = "io"
        for _ in range(num_inputs):                      # IO gates: q_o = 1, wire 4 = the public input
            v = self.free(VAR_UNIFORM)
            self.pub_vars.append(v)
            self.row(0, 0, 0, 0, out=v, q_o=1)

    def free(self, cls):
        v = self.num_vars
        self.num_vars += 1
        self.free_vars.append(v)
        self.free_class.append(cls)
        return v

    def row(self, w0, w1, w2, w3, out=None, lc=(0, 0, 0, 0), mul=(0, 0), hash_=(0, 0, 0, 0), q_o=1, q_c=0, q_ecc=0):
        """one gate; `out` None: a new variable defined by this row"""
        j = self.j
        if j >= self.n:
            raise ValueError("circuit does not fit the domain")
        if out is None:
            out = self.num_vars
            self.num_vars += 1
        for i, v in enumerate((w0, w1, w2, w3, out)):
            self.wv[i][j] = v
        for i in range(4):
            self.sel[Q_LC + i][j] = lc[i] % R
            self.sel[Q_HASH + i][j] = hash_[i] % R
        self.sel[Q_MUL][j], self.sel[Q_MUL + 1][j] = mul[0] % R, mul[1] % R
        self.sel[Q_O][j], self.sel[Q_C][j], self.sel[Q_ECC][j] = q_o % R, q_c % R, q_ecc % R
        self.j += 1
        self.family[self._fam] = self.family.get(self._fam, 0) + 1
        return out

    def c(self):
        return self.rng.field() | 1                      # a circuit constant (round constant, MDS entry, curve parameter)

    # ---- jf-relation look-alikes -----------------------------------------------------------------------------------
    def boolean(self):
        """bool_gate: b * b = b on wires (b, b, 0, 0, b)"""
        b = self.free(VAR_BOOL)
        self.row(b, b, 0, 0, out=b, mul=(1, 0))
        return b

    def logic_not(self, b):
        return self.row(b, 0, 0, 0, lc=(-1, 0, 0, 0), q_c=1)

    def logic_and(self, a, b):
        return self.row(a, b, 0, 0, mul=(1, 0))

    def logic_or(self, a, b):
        return self.row(a, b, 0, 0, lc=(1, 1, 0, 0), mul=(-1, 0))

    def select(self, b, nb, x, y):
        """conditional_select: b x + (1 - b) y"""
        return self.row(b, x, nb, y, mul=(1, 1))

    def lin(self, ws, coeffs, q_c=0):
        ws = list(ws) + [0] * (4 - len(ws))
        coeffs = list(coeffs) + [0] * (4 - len(coeffs))
        return self.row(*ws, lc=coeffs, q_c=q_c)

    def decompose(self, bits):
        """the linear-combination chain of a bit decomposition: four bits, then the running sum + three bits per gate
        (jf-relation's `unpack` / range gates); returns the recomposed value"""
        acc = self.lin(bits[:4], [1 << i for i in range(len(bits[:4]))])
        k = 4
        while k < len(bits):
            grp = bits[k:k + 3]
            acc = self.lin([acc] + grp, [1] + [1 << (k + i) for i in range(len(grp))])
            k += 3
        return acc

    def range_value(self, nbits):
        """enforce_in_range: nbits boolean variables + the chain; the value the chain recomposes is the checked variable"""
        return self.decompose([self.boolean() for _ in range(nbits)])

    def rescue_perm(self, st):
        """RescuePermutation gadget: 4 key additions + 12 rounds of (4 fifth-root gates on wires (y, 0, 0, 0, x), 4 affine
        gates, 4 gates sum M_ij x_j^5 + c) = 148 gates.  The fifth-root gate y^5 = x is laid down with the roles of the two
        cells exchanged (the new variable is the fifth power): same wire pattern, no 254-bit exponentiation per gate in the
        harness."""
        st = [self.lin([s], [1], q_c=self.c()) for s in st]
        for _ in range(12):
            st = [self.row(s, 0, 0, 0, hash_=(1, 0, 0, 0)) for s in st]
            st = [self.lin(st, [self.c() for _ in range(4)], q_c=self.c()) for _ in range(4)]
            st = [self.row(*st, hash_=tuple(self.c() for _ in range(4)), q_c=self.c()) for _ in range(4)]
        return st

    def sponge(self, elems):
        """Rescue sponge of rate 3: absorb three elements (one addition gate each), permute"""
        st = [0, 0, 0, 0]
        for k in range(0, len(elems), 3):
            chunk = elems[k:k + 3]
            for i, e in enumerate(chunk):
                st[i] = self.row(st[i], e, 0, 0, lc=(1, 1, 0, 0))
            st = self.rescue_perm(st)
        return st[0]

    def ecc_add(self, p, q):
        """Edwards addition: one gate per output coordinate, x3 (1 + d x1 x2 y1 y2) = x1 y2 + x2 y1 on wires
        (x1, y2, x2, y1, x3) with q_ecc = -d, and the same shape for y3"""
        d = self.c()
        x3 = self.row(p[0], q[1], q[0], p[1], mul=(1, 1), q_ecc=-d)
        y3 = self.row(p[0], q[0], p[1], q[1], mul=(self.c(), 1), q_ecc=d)
        return (x3, y3)

    def scalar_mul_variable(self, point):
        """variable-base scalar multiplication: per bit of the 254-bit scalar one boolean gate, a doubling, the selected
        point (b x, b y + 1 - b) and an addition; plus the scalar's recomposition chain"""
        bits = [self.boolean() for _ in range(254)]
        acc = point
        for b in bits:
            acc = self.ecc_add(acc, acc)
            sx = self.row(b, point[0], 0, 0, mul=(1, 0))
            sy = self.row(b, point[1], 0, 0, mul=(1, 0), lc=(-1, 0, 0, 0), q_c=1)
            acc = self.ecc_add(acc, (sx, sy))
        self.decompose(bits)
        return acc

    def scalar_mul_fixed(self):
        """fixed-base scalar multiplication: two bits select one of four constant points (a linear combination of b0, b1
        and b0 b1 per coordinate), one addition per bit pair"""
        bits = [self.boolean() for _ in range(254)]
        acc = None
        for k in range(0, 254, 2):
            b0, b1 = bits[k], bits[k + 1]
            b01 = self.logic_and(b0, b1)
            sx = self.lin([b0, b1, b01], [self.c(), self.c(), self.c()], q_c=self.c())
            sy = self.lin([b0, b1, b01], [self.c(), self.c(), self.c()], q_c=self.c())
            acc = (sx, sy) if acc is None else self.ecc_add(acc, (sx, sy))
        self.decompose(bits)
        return acc

    def merkle_level(self, cur):
        """one level of the 3-ary Merkle path: two position bits, the three children ordered by them, one Rescue hash"""
        s1, s2 = self.free(VAR_UNIFORM), self.free(VAR_UNIFORM)
        b0, b1 = self.boolean(), self.boolean()
        nb0, nb1 = self.logic_not(b0), self.logic_not(b1)
        left = self.select(b0, nb0, s1, cur)
        mid = self.select(b1, nb1, s2, self.select(b0, nb0, cur, s1))
        right = self.select(b1, nb1, cur, s2)
        return self.rescue_perm([left, mid, right, 0])[0]

    def fam(self, name):
        self._fam = name


def cap_like_circuit(kind: str = "transfer_2x2", seed: int = 2, tree_depth: int = 10) -> SyntheticCircuit:
    """A circuit with the COMPOSITION of the reference's transfer circuit (src/circuit/transfer.rs:53-193 and the gadgets
    it calls, src/circuit/gadgets.rs, structs.rs): per input a record commitment, a Merkle path of `tree_depth` levels,
    the nullifier PRF, the ownership check (fixed-base scalar multiplication), the credential's Schnorr verification (two
    variable-base scalar multiplications + a hash) and a handful of logic gates; per output a commitment and a 64-bit
    range check; the balance; and the viewing memo (ElGamal: one fixed- and one variable-base scalar multiplication and
    Rescue in counter mode over the 25 viewable scalars).  jf-relation, which builds the real one, is not in this image
    (Cargo.toml:33), so the gate counts per gadget follow its published constructions (Rescue permutation 148 gates,
    bits as boolean gates + a 3-bits-per-gate recomposition chain, Edwards additions as one gate per coordinate) and the
    constants are random: the model reproduces WHICH CELLS HOLD 0 / 1 / SMALL / FULL-WIDTH VALUES - what decides the
    prover's work when wire commitments are taken from evaluations - and lands, like the reference, on n = 2^15 for
    2-in/2-out at depth 10 (src/utils/mod.rs:149-153).  `value_classes()` of a witness is reported with every number
    measured on it."""
    log_n, num_inputs = NOTE_SHAPES[kind]
    n_in, n_out = {"transfer_2x2": (2, 2), "transfer_2x3": (2, 3), "transfer_2x2_d26": (2, 2), "transfer_3x5_d26": (3, 5)}[kind]
    b = _GadgetBuilder(log_n, num_inputs, seed)
    amounts_in, amounts_out = [], []
    for i in range(n_in):
        b.fam("record commitment")
        amount = b.free(VAR_U64)
        amounts_in.append(amount)
        freeze = b.boolean()
        fields = [amount] + [b.free(VAR_UNIFORM) for _ in range(10)] + [freeze]      # asset code, address, policy, blind
        rc = b.sponge(fields)
        b.fam("logic")
        is_dummy, is_zero_amt = b.boolean(), b.boolean()
        b.logic_or(b.logic_not(is_dummy), is_zero_amt)
        b.logic_or(is_dummy, b.boolean())
        b.row(freeze, 0, 0, 0, out=0, lc=(0, 0, 0, 0), q_o=1)                         # enforce_constant-style row
        b.fam("merkle path")
        cur = rc
        for _ in range(tree_depth):
            cur = b.merkle_level(cur)
        b.fam("nullifier")
        b.sponge([b.free(VAR_UNIFORM), rc, cur])
        b.fam("ownership (fixed-base)")
        b.scalar_mul_fixed()
        b.fam("credential (variable-base x2 + hash)")
        pk = (b.free(VAR_UNIFORM), b.free(VAR_UNIFORM))
        r1 = b.scalar_mul_variable(pk)
        r2 = b.scalar_mul_variable((b.free(VAR_UNIFORM), b.free(VAR_UNIFORM)))
        b.sponge([r1[0], r1[1], r2[0], r2[1], b.free(VAR_UNIFORM), b.free(VAR_UNIFORM)])
        b.ecc_add(r1, r2)
    for i in range(n_out):
        b.fam("range check")
        amount = b.range_value(64)
        amounts_out.append(amount)
        b.fam("record commitment")
        b.sponge([amount] + [b.free(VAR_UNIFORM) for _ in range(10)] + [b.boolean()])
    b.fam("balance")
    tot_in = b.lin(amounts_in[:4], [1] * len(amounts_in[:4]))
    tot_out = b.lin(amounts_out[:4], [1] * len(amounts_out[:4]))
    diff = b.lin([tot_in, tot_out], [1, -1])
    b.fam("range check")
    b.range_value(64)                                                                # reveal threshold - amount in range
    b.fam("viewing memo (ElGamal)")
    b.scalar_mul_fixed()
    shared = b.scalar_mul_variable((b.free(VAR_UNIFORM), b.free(VAR_UNIFORM)))
    data = [diff] + [b.free(VAR_UNIFORM) for _ in range(24)]
    key = b.sponge([shared[0], shared[1]])
    for k in range(0, 25, 3):                                                        # counter mode: a permutation per 3 scalars
        ks = b.rescue_perm([key, b.lin([1], [k]), 0, 0])
        for i, d_ in enumerate(data[k:k + 3]):
            b.row(d_, ks[i], 0, 0, lc=(1, 1, 0, 0))
    gate_rows = b.j
    if gate_rows <= (1 << (log_n - 1)):
        raise ValueError("model circuit is smaller than half the pinned domain")
    sigma = _permutation(b.wv, b.num_vars, log_n)
    fam = dict(b.family)
    fam["padding"] = b.n - gate_rows
    return SyntheticCircuit(log_n=log_n, num_inputs=num_inputs, selectors=b.sel, wire_vars=b.wv, num_vars=b.num_vars,
                            free_vars=b.free_vars, pub_vars=b.pub_vars, gate_rows=gate_rows, sigma=sigma,
                            free_class=b.free_class, gadget_rows=fam)


def note_circuit(kind: str, seed: int = 2) -> SyntheticCircuit:
    log_n, num_inputs = NOTE_SHAPES[kind]
    return synthetic_circuit(log_n, num_inputs, seed)


def closed_loop_callers(prove_fn, pk: int, wires, pubs, num_inputs: int, msg: bytes, blinders, proofs, threads: int,
                        calls: int):
    """`threads` NATIVE threads (cap_amd/csrc/witgen.c: capwit_closed_loop_callers), each calling `prove_fn` - the ctypes
    function object of capgpu_plonk_prove_ex - `calls` times, one proof per call, straight after one another: the
    reference's rayon workers (src/utils/params_builder.rs:194-226).  Call k of thread t proves entry t * calls + k of
    wires / pubs / blinders (lists of C-contiguous uint64 arrays) into proofs[t * calls + k] (ctypes structures).
    Returns (failed calls, seconds from the common start to the last return)."""
    import ctypes
    total = threads * calls
    assert len(wires) == len(pubs) == len(blinders) == len(proofs) == total
    vp = ctypes.c_void_p
    arr = lambda xs: (vp * total)(*[x.ctypes.data for x in xs])      # noqa: E731
    pr = (vp * total)(*[ctypes.addressof(p) for p in proofs])
    mbuf = (ctypes.c_uint8 * max(len(msg), 1)).from_buffer_copy(msg or b"\0")
    secs = ctypes.c_double(0)
    fn = _witgen().capwit_closed_loop_callers
    fn.restype = ctypes.c_int
    failed = fn(ctypes.cast(prove_fn, vp), ctypes.c_uint64(pk), ctypes.c_int(threads), ctypes.c_int(calls), arr(wires),
                arr(pubs), ctypes.c_size_t(num_inputs), mbuf, ctypes.c_size_t(len(msg)), arr(blinders), pr,
                ctypes.byref(secs))
    return failed, secs.value


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
