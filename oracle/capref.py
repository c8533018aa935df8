"""ctypes loader for oracle/_build/libcapref.so (the C restatement).

TEST INFRASTRUCTURE ONLY - see the header of oracle/capref.c.  Data crosses as
numpy uint64 arrays in arkworks' in-memory layout: field element = 4 x u64
little-endian limbs (Montgomery unless stated), G1 affine = 8 words (x, y; the
point at infinity is (0, 0)), G1 Jacobian = 12 words (X, Y, Z).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libcapref.so")
_lib = None

u64p = ctypes.POINTER(ctypes.c_uint64)


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("capref.c", "capref_plonk.c", "Makefile")]
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_SO)
        _lib.capref_msm_window.restype = ctypes.c_uint
        _lib.capref_msm_window.argtypes = [ctypes.c_size_t]
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


# ---- int <-> limb helpers ---------------------------------------------------
def int_to_limbs(v: int) -> np.ndarray:
    return np.array([(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)], dtype=np.uint64)


def ints_to_array(vals) -> np.ndarray:
    out = np.empty((len(vals), 4), dtype=np.uint64)
    for i, v in enumerate(vals):
        for j in range(4):
            out[i, j] = (v >> (64 * j)) & 0xFFFFFFFFFFFFFFFF
    return out


def array_to_ints(a: np.ndarray):
    a = np.asarray(a, dtype=np.uint64).reshape(-1, 4)
    return [int(r[0]) | (int(r[1]) << 64) | (int(r[2]) << 128) | (int(r[3]) << 192) for r in a]


# ---- wrappers ---------------------------------------------------------------
def fp_op(which: int, op: str, a: int, b: int = 0) -> int:
    out = np.zeros(4, dtype=np.uint64)
    lib().capref_fp_op(which, ord(op), _p(int_to_limbs(a)), _p(int_to_limbs(b)), _p(out))
    return array_to_ints(out)[0]


def vec_to_mont(which: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().capref_fp_vec_to_mont(which, _p(a), ctypes.c_size_t(a.size // 4))
    return a


def vec_from_mont(which: int, a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64).copy()
    lib().capref_fp_vec_from_mont(which, _p(a), ctypes.c_size_t(a.size // 4))
    return a


def random_field(seed: int, which: int, n: int, mont: bool) -> np.ndarray:
    out = np.zeros((n, 4), dtype=np.uint64)
    lib().capref_random_field(ctypes.c_uint64(seed), which, int(mont), _p(out), ctypes.c_size_t(n))
    return out


def msm_g1(bases: np.ndarray, scalars: np.ndarray, c: int = 0) -> np.ndarray:
    """bases (n,8) Montgomery affine, scalars (n,4) canonical -> Jacobian (12,) Montgomery."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64)
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = min(bases.size // 8, scalars.size // 4)
    out = np.zeros(12, dtype=np.uint64)
    lib().capref_msm_g1(_p(bases), _p(scalars), ctypes.c_size_t(n), ctypes.c_uint(c), _p(out))
    return out


def g1_to_affine(jac: np.ndarray) -> np.ndarray:
    jac = np.ascontiguousarray(jac, dtype=np.uint64)
    out = np.zeros(8, dtype=np.uint64)
    lib().capref_g1_to_affine(_p(jac), _p(out))
    return out


def g1_mul(aff: np.ndarray, k: int) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().capref_g1_mul(_p(np.ascontiguousarray(aff, dtype=np.uint64)), _p(int_to_limbs(k)), _p(out))
    return out


def g1_add(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    out = np.zeros(12, dtype=np.uint64)
    lib().capref_g1_add(_p(np.ascontiguousarray(a, dtype=np.uint64)),
                        _p(np.ascontiguousarray(b, dtype=np.uint64)), _p(out))
    return out


def g1_fixed_base_batch(scalars: np.ndarray) -> np.ndarray:
    """[s_i] G for canonical scalars (n,4) -> (n,8) Montgomery affine."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    n = scalars.size // 4
    out = np.zeros((n, 8), dtype=np.uint64)
    lib().capref_g1_fixed_base_batch(_p(scalars), ctypes.c_size_t(n), _p(out))
    return out


def ntt_fr(data: np.ndarray, log_n: int, inverse: bool, coset: bool) -> np.ndarray:
    data = np.ascontiguousarray(data, dtype=np.uint64).copy()
    assert data.size == 4 << log_n
    lib().capref_ntt_fr(_p(data), ctypes.c_uint(log_n), int(inverse), int(coset))
    return data


def poly_eval_fr(coeffs: np.ndarray, x_mont: int) -> int:
    coeffs = np.ascontiguousarray(coeffs, dtype=np.uint64)
    out = np.zeros(4, dtype=np.uint64)
    lib().capref_poly_eval_fr(_p(coeffs), ctypes.c_size_t(coeffs.size // 4), _p(int_to_limbs(x_mont)), _p(out))
    return array_to_ints(out)[0]


def affine_to_ints(aff: np.ndarray):
    """Montgomery affine (8 words) -> canonical (x, y) ints, or None for infinity."""
    from . import bn254
    x, y = array_to_ints(np.asarray(aff, dtype=np.uint64).reshape(2, 4))
    if x == 0 and y == 0:
        return None
    return (bn254.from_mont(x, bn254.P), bn254.from_mont(y, bn254.P))


def points_to_array(points) -> np.ndarray:
    """list of canonical affine (x, y) / None -> (n, 8) Montgomery array."""
    from . import bn254
    vals = []
    for pt in points:
        if pt is None:
            vals += [0, 0]
        else:
            vals += [bn254.to_mont(pt[0], bn254.P), bn254.to_mont(pt[1], bn254.P)]
    return ints_to_array(vals).reshape(-1, 8)


# ---- PLONK prover restatement (oracle/capref_plonk.c) ----------------------------------------------
PROOF_WORDS = 13 * 8 + 10 * 4


class PlonkKey:
    def __init__(self, srs_bases: np.ndarray, n: int, num_inputs: int, selectors: np.ndarray, sigma_evals: np.ndarray):
        L = lib()
        L.capref_plonk_preprocess.restype = ctypes.c_void_p
        srs_bases = np.ascontiguousarray(srs_bases, dtype=np.uint64).reshape(-1)
        assert srs_bases.size >= 8 * (n + 3)
        selectors = np.ascontiguousarray(selectors, dtype=np.uint64).reshape(-1)
        sigma_evals = np.ascontiguousarray(sigma_evals, dtype=np.uint64).reshape(-1)
        self.vk_comms = np.zeros((18, 8), dtype=np.uint64)
        self.n, self.num_inputs = n, num_inputs
        self.h = ctypes.c_void_p(L.capref_plonk_preprocess(_p(srs_bases), ctypes.c_size_t(n), ctypes.c_size_t(num_inputs),
                                                           _p(selectors), _p(sigma_evals), _p(self.vk_comms)))

    def prove(self, wires: np.ndarray, pub_inputs: np.ndarray, blinders: np.ndarray, ext_msg: bytes | None = None):
        """all Montgomery; returns (rc, comms (13,8), evals (10,4))."""
        L = lib()
        wires = np.ascontiguousarray(wires, dtype=np.uint64).reshape(-1)
        pub = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1)
        if pub.size == 0:
            pub = np.zeros(4, dtype=np.uint64)
        bl = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1)
        out = np.zeros(PROOF_WORDS, dtype=np.uint64)
        msg = ext_msg or b""
        buf = (ctypes.c_uint8 * max(len(msg), 1)).from_buffer_copy(msg + (b"\0" if not msg else b""))
        rc = L.capref_plonk_prove(self.h, _p(wires), _p(pub), buf, ctypes.c_size_t(len(msg)), _p(bl), _p(out))
        return rc, out[:104].reshape(13, 8).copy(), out[104:].reshape(10, 4).copy()

    def free(self):
        if self.h:
            lib().capref_plonk_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def keccak256(data: bytes) -> bytes:
    out = (ctypes.c_uint8 * 32)()
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data + (b"\0" if not data else b""))
    lib().capref_keccak256(buf, ctypes.c_size_t(len(data)), out)
    return bytes(out)
