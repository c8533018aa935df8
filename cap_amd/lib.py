"""ctypes binding of include/capgpu.h (the drop-in C ABI).  No compute happens in Python."""
from __future__ import annotations

import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("CAPGPU_LIBRARY") or os.path.join(_HERE, "libcapgpu.so")  # override: A/B builds in tools/
_lib = None

NUM_WIRE_TYPES = 5
NUM_SELECTORS = 13

CAPGPU_OK = 0
ERR_NAMES = {
    -1: "CAPGPU_ERR_INVALID_ARG", -2: "CAPGPU_ERR_NO_DEVICE", -3: "CAPGPU_ERR_HIP", -4: "CAPGPU_ERR_BAD_HANDLE",
    -5: "CAPGPU_ERR_OOM", -6: "CAPGPU_ERR_NOT_INITIALISED", -7: "CAPGPU_ERR_PROOF", -8: "CAPGPU_ERR_SERIALIZATION",
    -9: "CAPGPU_ERR_COMM",
}

u64p = ctypes.POINTER(ctypes.c_uint64)


class CapGpuError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"{ERR_NAMES.get(code, code)}: {msg}")
        self.code = code


class Proof(ctypes.Structure):
    """capgpu_proof (include/capgpu.h) == jf_plonk::proof_system::structs::Proof fields."""
    _fields_ = [
        ("wires_poly_comms", (ctypes.c_uint64 * 8) * NUM_WIRE_TYPES),
        ("prod_perm_poly_comm", ctypes.c_uint64 * 8),
        ("split_quot_poly_comms", (ctypes.c_uint64 * 8) * NUM_WIRE_TYPES),
        ("opening_proof", ctypes.c_uint64 * 8),
        ("shifted_opening_proof", ctypes.c_uint64 * 8),
        ("wires_evals", (ctypes.c_uint64 * 4) * NUM_WIRE_TYPES),
        ("wire_sigma_evals", (ctypes.c_uint64 * 4) * (NUM_WIRE_TYPES - 1)),
        ("perm_next_eval", ctypes.c_uint64 * 4),
    ]


class VerifyingKey(ctypes.Structure):
    _fields_ = [
        ("domain_size", ctypes.c_uint64),
        ("num_inputs", ctypes.c_uint64),
        ("k", (ctypes.c_uint64 * 4) * NUM_WIRE_TYPES),
        ("selector_comms", (ctypes.c_uint64 * 8) * NUM_SELECTORS),
        ("sigma_comms", (ctypes.c_uint64 * 8) * NUM_WIRE_TYPES),
    ]


def lib_path() -> str:
    return _SO


def load():
    """Load libcapgpu.so.  Fails loudly when it has not been built (no fallback)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(_SO):
        raise CapGpuError(-2, f"{_SO} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(make -C cap_amd/csrc).  There is no CPU fallback.")
    L = ctypes.CDLL(_SO)
    L.capgpu_last_error.restype = ctypes.c_char_p
    L.capgpu_version.restype = ctypes.c_char_p
    _lib = L
    return L


def check(rc: int):
    if rc != CAPGPU_OK:
        raise CapGpuError(rc, load().capgpu_last_error().decode())


def init(device: int | None = None, devices=None):
    """One device (`device`; default LOCAL_RANK, the process-per-GPU launch) or, with `devices`, the list of HIP device
    ids this ONE process drives (capgpu_init(device_ids, n): one context per id)."""
    L = load()
    if devices is None:
        if device is None:
            device = int(os.environ.get("LOCAL_RANK", "0"))
        devices = [device]
    ids = (ctypes.c_int * len(devices))(*devices)
    check(L.capgpu_init(ids, len(devices)))
    return L


def shutdown():
    load().capgpu_shutdown()


def device_count() -> int:
    n = ctypes.c_int(0)
    check(load().capgpu_device_count(ctypes.byref(n)))
    return n.value


def physical_device_count() -> int:
    """distinct HIP devices behind the contexts (device_count() counts CONTEXTS: four on one bound GPU by default)"""
    out = ctypes.c_int(0)
    check(load().capgpu_physical_device_count(ctypes.byref(out)))
    return out.value


def timer_begin():
    check(load().capgpu_timer_begin())


def timer_end() -> float:
    """milliseconds of device time on the calling thread's context since timer_begin (HIP events on its stream)"""
    out = ctypes.c_double(0)
    check(load().capgpu_timer_end(ctypes.byref(out)))
    return out.value


def set_device(slot: int):
    """bind the calling thread to context `slot` (-1: unbind)"""
    check(load().capgpu_set_device(int(slot)))


def get_device():
    s, d = ctypes.c_int(0), ctypes.c_int(0)
    check(load().capgpu_get_device(ctypes.byref(s), ctypes.byref(d)))
    return s.value, d.value


def mem_info():
    """(free, total) device memory in bytes"""
    f, t = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load().capgpu_mem_info(ctypes.byref(f), ctypes.byref(t)))
    return f.value, t.value


def trim():
    """capgpu_trim: release the scratch, pinned result areas and captured graphs of every idle context.
    Returns (device bytes released, contexts skipped because a call was running on them)."""
    b, busy = ctypes.c_uint64(0), ctypes.c_int(0)
    check(load().capgpu_trim(ctypes.byref(b), ctypes.byref(busy)))
    return b.value, busy.value


def set_memory_limit(scratch_bytes_per_device: int):
    """capgpu_set_memory_limit: cap on the scratch the library holds per device (0 = none); a call that would grow past it
    fails with CAPGPU_ERR_OOM (CapGpuError code -5) after trimming the device's idle contexts."""
    check(load().capgpu_set_memory_limit(ctypes.c_uint64(scratch_bytes_per_device)))


def scratch_info():
    """(scratch bytes held on the calling thread's device, the cap - 0 = none)"""
    b, lim = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load().capgpu_scratch_info(ctypes.byref(b), ctypes.byref(lim)))
    return b.value, lim.value


def trace_enable(on: bool):
    check(load().capgpu_trace_enable(int(bool(on))))


def trace_dump(path: str) -> int:
    n = ctypes.c_uint64(0)
    check(load().capgpu_trace_dump(path.encode(), ctypes.byref(n)))
    return n.value


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(u64p)


# ---- device buffers ----------------------------------------------------------------------------
class DevBuf:
    def __init__(self, nbytes: int):
        self.ptr = ctypes.c_void_p()
        self.nbytes = nbytes
        check(load().capgpu_malloc(ctypes.byref(self.ptr), ctypes.c_size_t(nbytes)))

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DevBuf":
        a = np.ascontiguousarray(a)
        b = cls(a.nbytes)
        check(load().capgpu_memcpy_h2d(b.ptr, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes)))
        return b

    def view(self, offset_bytes: int, nbytes: int) -> "DevBuf":
        """a non-owning window of this buffer (the parent must outlive it)"""
        assert 0 <= offset_bytes and offset_bytes + nbytes <= self.nbytes
        v = DevBuf.__new__(DevBuf)
        v.ptr = ctypes.c_void_p(self.ptr.value + offset_bytes)
        v.nbytes = nbytes
        v._view = True
        return v

    def upload(self, a: np.ndarray):
        """overwrite the buffer's first a.nbytes bytes (same device address: resident inputs of a replayed schedule)"""
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        check(load().capgpu_memcpy_h2d(self.ptr, a.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(a.nbytes)))

    def to_numpy(self, dtype=np.uint64, count: int | None = None, offset_bytes: int = 0) -> np.ndarray:
        itemsize = np.dtype(dtype).itemsize
        n = (self.nbytes - offset_bytes) // itemsize if count is None else count
        out = np.empty(n, dtype=dtype)
        src = ctypes.c_void_p(self.ptr.value + offset_bytes)
        check(load().capgpu_memcpy_d2h(out.ctypes.data_as(ctypes.c_void_p), src, ctypes.c_size_t(n * itemsize)))
        return out

    def free(self):
        if getattr(self, "_view", False):
            self.ptr = ctypes.c_void_p()
            return
        if self.ptr and self.ptr.value:
            check(load().capgpu_free(self.ptr))
            self.ptr = ctypes.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def sync_all():
    """capgpu_sync_all: hipDeviceSynchronize on every bound device"""
    check(load().capgpu_sync_all())


def runtime_info():
    """(HIP runtime version, HIP driver version) the process runs on - the first libamdhip64.so.7 the loader met"""
    r, d = ctypes.c_int(0), ctypes.c_int(0)
    check(load().capgpu_runtime_info(ctypes.byref(r), ctypes.byref(d)))
    return r.value, d.value


def sync():
    check(load().capgpu_sync())


# ---- SRS -----------------------------------------------------------------------------------------
def srs_upload(bases: np.ndarray, montgomery: bool = True) -> int:
    """bases: (n, 8) uint64 packed affine points."""
    bases = np.ascontiguousarray(bases, dtype=np.uint64).reshape(-1, 8)
    h = ctypes.c_uint64()
    check(load().capgpu_srs_upload(bases.ctypes.data_as(ctypes.c_void_p), ctypes.c_size_t(bases.shape[0]),
                                   ctypes.c_size_t(64), int(montgomery), ctypes.byref(h)))
    return h.value


def _limbs(v: int):
    return (ctypes.c_uint64 * 4)(*[(v >> (64 * i)) & 0xFFFFFFFFFFFFFFFF for i in range(4)])


def srs_generate(tau: int, n: int) -> int:
    h = ctypes.c_uint64()
    check(load().capgpu_srs_generate(_limbs(tau), ctypes.c_size_t(n), ctypes.byref(h)))
    return h.value


def srs_generate_hiding(tau: int, gamma: int, n: int) -> int:
    h = ctypes.c_uint64()
    check(load().capgpu_srs_generate_hiding(_limbs(tau), _limbs(gamma), ctypes.c_size_t(n), ctypes.byref(h)))
    return h.value


def srs_generate_affine_seq(a: int, b: int, n: int) -> int:
    h = ctypes.c_uint64()
    check(load().capgpu_srs_generate_affine_seq(_limbs(a), _limbs(b), ctypes.c_size_t(n), ctypes.byref(h)))
    return h.value


def srs_download(handle: int, offset: int, n: int) -> np.ndarray:
    out = np.empty((n, 8), dtype=np.uint64)
    check(load().capgpu_srs_download(ctypes.c_uint64(handle), ctypes.c_size_t(offset), ctypes.c_size_t(n),
                                     out.ctypes.data_as(ctypes.c_void_p)))
    return out


def srs_free(handle: int):
    check(load().capgpu_srs_free(ctypes.c_uint64(handle)))


def srs_shards(handle: int) -> int:
    n = ctypes.c_int(0)
    check(load().capgpu_srs_shards(ctypes.c_uint64(handle), ctypes.byref(n)))
    return n.value


def srs_size(handle: int) -> int:
    n = ctypes.c_size_t(0)
    check(load().capgpu_srs_size(ctypes.c_uint64(handle), ctypes.byref(n)))
    return n.value


# ---- MSM -----------------------------------------------------------------------------------------
def msm_g1(handle: int, scalars: np.ndarray, offset: int = 0) -> np.ndarray:
    """scalars (n,4) canonical -> Jacobian (12,) Montgomery."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    check(load().capgpu_msm_g1(ctypes.c_uint64(handle), ctypes.c_size_t(offset), _p(scalars),
                               ctypes.c_size_t(scalars.shape[0]), _p(out)))
    return out


def lagrange_commit(handle: int, log_n: int, scalars_mont: np.ndarray) -> np.ndarray:
    """KZG commitment from VALUES on the 2^log_n domain (+ up to three blinders): MSM on the Lagrange-form commit key"""
    sc = np.ascontiguousarray(scalars_mont, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, np.uint64)
    check(load().capgpu_msm_g1_lagrange(ctypes.c_uint64(handle), ctypes.c_uint32(log_n), _p(sc), ctypes.c_size_t(sc.shape[0]),
                                        1, _p(out)))
    return out


def msm_g1_batch(handle: int, scalar_list, offsets=None) -> np.ndarray:
    cnt = len(scalar_list)
    arrs = [np.ascontiguousarray(s, dtype=np.uint64).reshape(-1, 4) for s in scalar_list]
    offs = (ctypes.c_size_t * cnt)(*([0] * cnt if offsets is None else offsets))
    ns = (ctypes.c_size_t * cnt)(*[a.shape[0] for a in arrs])
    ptrs = (u64p * cnt)(*[_p(a) for a in arrs])
    out = np.zeros((cnt, 12), dtype=np.uint64)
    check(load().capgpu_msm_g1_batch(ctypes.c_uint64(handle), offs, ptrs, ns, cnt, _p(out)))
    return out


def msm_g1_dev(handle: int, d_scalars: DevBuf, n: int, count: int = 1, stride: int | None = None,
               montgomery: bool = False, offset: int = 0, d_out: DevBuf | None = None) -> DevBuf:
    if d_out is None:
        d_out = DevBuf(96 * count)
    check(load().capgpu_msm_g1_dev(ctypes.c_uint64(handle), ctypes.c_size_t(offset), d_scalars.ptr,
                                   ctypes.c_size_t(n if stride is None else stride), ctypes.c_size_t(n), count,
                                   int(montgomery), d_out.ptr))
    return d_out


def msm_scalars_upload(handle: int, scalars: np.ndarray, offset: int = 0) -> int:
    """scalars (count, n, 4) or (n, 4) in host memory -> a scalar set resident with the SRS's point ranges
    (capgpu_msm_scalars_upload: each slice goes to the device that holds its points)."""
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64)
    scalars = scalars.reshape(1, -1, 4) if scalars.ndim == 2 else scalars
    count, n = scalars.shape[0], scalars.shape[1]
    h = ctypes.c_uint64(0)
    check(load().capgpu_msm_scalars_upload(ctypes.c_uint64(handle), ctypes.c_size_t(offset), _p(scalars.reshape(-1)),
                                           ctypes.c_size_t(n), ctypes.c_size_t(n), count, ctypes.byref(h)))
    return h.value


def msm_scalars_scatter_dev(handle: int, d_scalars: DevBuf, n: int, count: int = 1, stride: int | None = None,
                            offset: int = 0) -> int:
    """the same from device memory of the calling thread's context (one peer copy per slice, once)"""
    h = ctypes.c_uint64(0)
    check(load().capgpu_msm_scalars_scatter_dev(ctypes.c_uint64(handle), ctypes.c_size_t(offset), d_scalars.ptr,
                                                ctypes.c_size_t(n if stride is None else stride), ctypes.c_size_t(n),
                                                count, ctypes.byref(h)))
    return h.value


def msm_scalars_free(scalars_handle: int):
    check(load().capgpu_msm_scalars_free(ctypes.c_uint64(scalars_handle)))


def msm_g1_resident(handle: int, scalars_handle: int, count: int = 1, montgomery: bool = False,
                    d_out: DevBuf | None = None) -> DevBuf:
    """MSM(s) on a resident scalar set: only the 96-byte partials move between devices."""
    if d_out is None:
        d_out = DevBuf(96 * count)
    check(load().capgpu_msm_g1_resident(ctypes.c_uint64(handle), ctypes.c_uint64(scalars_handle), int(montgomery),
                                        d_out.ptr))
    return d_out


def msm_shard_stats() -> dict:
    """bytes moved between device contexts (or from the host) by sharded MSMs since init, sharded calls, replications"""
    a, b, c, d = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load().capgpu_msm_shard_stats(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c), ctypes.byref(d)))
    return {"scalar_bytes": a.value, "partial_bytes": b.value, "sharded_calls": c.value, "replications": d.value}


def device_peer_info(slot_a: int, slot_b: int) -> int:
    """1: direct peer access between the two contexts' devices, 0: none (copies staged by the runtime), 2: same device"""
    x = ctypes.c_int(-1)
    check(load().capgpu_device_peer_info(slot_a, slot_b, ctypes.byref(x)))
    return x.value


def msm_plan(handle: int, n: int, count: int = 1) -> dict:
    """Which table / sort / split `count` MSMs of n points would take: {'c': 15, 'windows': 18, 'sort': 'two-level',
    'parts': 256, 'n_sub': 65536, 'slice': 1}."""
    buf = ctypes.create_string_buffer(256)
    check(load().capgpu_msm_plan(ctypes.c_uint64(handle), ctypes.c_size_t(n), count, buf, ctypes.c_size_t(256)))
    out = {}
    for kv in buf.value.decode().split():
        if "=" not in kv:
            continue
        k, v = kv.split("=")
        out[k] = int(v) if v.isdigit() else v
    return out


# ---- multi-GPU (one process per GPU; RCCL inside the library) -----------------------------------------------------
def comm_unique_id() -> bytes:
    buf = (ctypes.c_uint8 * 128)()
    check(load().capgpu_comm_unique_id(buf))
    return bytes(buf)


def comm_init(rank: int, world: int, unique_id: bytes):
    assert len(unique_id) == 128
    check(load().capgpu_comm_init(rank, world, (ctypes.c_uint8 * 128).from_buffer_copy(unique_id)))


def comm_init_loopback(world: int):
    """test communicator: this process plays `world` ranks one after the other (capgpu.h)"""
    check(load().capgpu_comm_init_loopback(int(world)))


def comm_loopback_set_rank(rank: int):
    check(load().capgpu_comm_loopback_set_rank(int(rank)))


def comm_destroy():
    check(load().capgpu_comm_destroy())


def comm_info():
    r, w = ctypes.c_int(0), ctypes.c_int(0)
    check(load().capgpu_comm_info(ctypes.byref(r), ctypes.byref(w)))
    return r.value, w.value


def msm_g1_sharded_dev(handle: int, d_scalars: DevBuf, n_local: int, count: int = 1, stride: int | None = None,
                       montgomery: bool = False, offset: int = 0, d_out: DevBuf | None = None) -> DevBuf:
    if d_out is None:
        d_out = DevBuf(96 * count)
    check(load().capgpu_msm_g1_sharded_dev(ctypes.c_uint64(handle), ctypes.c_size_t(offset), d_scalars.ptr,
                                           ctypes.c_size_t(n_local if stride is None else stride),
                                           ctypes.c_size_t(n_local), count, int(montgomery), d_out.ptr))
    return d_out


def msm_g1_sharded(handle: int, scalars: np.ndarray, offset: int = 0) -> np.ndarray:
    scalars = np.ascontiguousarray(scalars, dtype=np.uint64).reshape(-1, 4)
    out = np.zeros(12, dtype=np.uint64)
    check(load().capgpu_msm_g1_sharded(ctypes.c_uint64(handle), ctypes.c_size_t(offset),
                                       _p(scalars) if scalars.size else None, ctypes.c_size_t(scalars.shape[0]),
                                       _p(out)))
    return out


def plonk_shard_msm(on: bool):
    check(load().capgpu_plonk_shard_msm(int(on)))


def g1_sum(points: np.ndarray) -> np.ndarray:
    """(k, 12) Jacobian points -> their group sum (12,)."""
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 12)
    out = np.zeros(12, dtype=np.uint64)
    check(load().capgpu_g1_sum(_p(points.reshape(-1)), ctypes.c_size_t(points.shape[0]), _p(out)))
    return out


# ---- NTT -----------------------------------------------------------------------------------------
def ntt_fr(data: np.ndarray, log_n: int, inverse: bool = False, coset: bool = False) -> np.ndarray:
    data = np.ascontiguousarray(data, dtype=np.uint64).copy()
    assert data.size == 4 << log_n
    check(load().capgpu_ntt_fr(_p(data), ctypes.c_uint32(log_n), int(inverse), int(coset)))
    return data


def ntt_fr_batch(arrays, log_n: int, inverse: bool = False, coset: bool = False):
    arrs = [np.ascontiguousarray(a, dtype=np.uint64).copy() for a in arrays]
    ptrs = (u64p * len(arrs))(*[_p(a) for a in arrs])
    check(load().capgpu_ntt_fr_batch(ptrs, len(arrs), ctypes.c_uint32(log_n), int(inverse), int(coset)))
    return arrs


def ntt_fr_dev(d_data: DevBuf, log_n: int, count: int = 1, stride: int | None = None, inverse: bool = False,
               coset: bool = False):
    check(load().capgpu_ntt_fr_dev(d_data.ptr, ctypes.c_size_t((1 << log_n) if stride is None else stride), count,
                                   ctypes.c_uint32(log_n), int(inverse), int(coset)))


# ---- instrumentation ---------------------------------------------------------------------------
def ubench_mad_rate() -> float:
    """measured v_mad_u64_u32 lane-operations per second of the bound device"""
    out = ctypes.c_double(0)
    check(load().capgpu_ubench_mad_rate(ctypes.byref(out)))
    return out.value


ISSUE_CLASSES = ("v_mad_u64_u32", "v_add_u32", "v_and_b32", "v_mov_b32", "v_lshl_add_u64", "v_lshrrev_b64",
                 "v_alignbit_b32", "v_mul_lo_u32", "mixed_3mad_1plain", "mixed_3mad_1plain_at_3_waves_per_simd")


def ubench_issue_rates() -> dict:
    """measured issue rate (lane-operations per second) of each instruction class, at equal occupancy"""
    out = (ctypes.c_double * len(ISSUE_CLASSES))()
    check(load().capgpu_ubench_issue_rates(out, len(ISSUE_CLASSES)))
    return dict(zip(ISSUE_CLASSES, [float(v) for v in out]))


def profile_enable(on: bool):
    check(load().capgpu_profile_enable(int(on)))


def profile_reset():
    check(load().capgpu_profile_reset())


def profile_stats() -> dict:
    buf = ctypes.create_string_buffer(1 << 16)
    check(load().capgpu_profile_dump(buf, ctypes.c_size_t(len(buf))))
    out = {}
    for line in buf.value.decode().splitlines():
        name, ms, cnt = line.split()
        out[name] = (float(ms), int(cnt))
    return out


# ---- PLONK ---------------------------------------------------------------------------------------
INPUT_EVALS, INPUT_COEFFS = 0, 1  # CAPGPU_INPUT_* of include/capgpu.h


def _form(input_form) -> int:
    """'evals' / 'coeffs' / 0 / 1 -> the ABI's input_form integer (anything else is passed on for the library to refuse)"""
    if isinstance(input_form, str):
        return {"evals": INPUT_EVALS, "coeffs": INPUT_COEFFS}[input_form]
    return int(input_form)


def plonk_preprocess(srs_handle: int, n: int, num_inputs: int, selectors: np.ndarray, sigma_evals: np.ndarray,
                     input_form=INPUT_EVALS):
    """selectors (13, n, 4), sigma_evals (5, n, 4) Montgomery -> (pk handle, VerifyingKey).  input_form = 'coeffs':
    the 18 columns are polynomials in coefficient form (capgpu_plonk_preprocess_ex)."""
    selectors = np.ascontiguousarray(selectors, dtype=np.uint64)
    sigma_evals = np.ascontiguousarray(sigma_evals, dtype=np.uint64)
    assert selectors.size == NUM_SELECTORS * n * 4 and sigma_evals.size == NUM_WIRE_TYPES * n * 4
    h = ctypes.c_uint64()
    vk = VerifyingKey()
    check(load().capgpu_plonk_preprocess_ex(ctypes.c_uint64(srs_handle), ctypes.c_size_t(n),
                                            ctypes.c_size_t(num_inputs), _p(selectors.reshape(-1)),
                                            _p(sigma_evals.reshape(-1)), ctypes.c_int(_form(input_form)),
                                            ctypes.byref(h), ctypes.byref(vk)))
    return h.value, vk


def plonk_key_info(pk_handle: int):
    """-> (domain size n, number of public inputs, SRS handle) of a resident proving key."""
    n, ni, srs = ctypes.c_size_t(0), ctypes.c_size_t(0), ctypes.c_uint64(0)
    check(load().capgpu_plonk_key_info(ctypes.c_uint64(pk_handle), ctypes.byref(n), ctypes.byref(ni), ctypes.byref(srs)))
    return n.value, ni.value, srs.value


def _check_prove_shapes(pk_handle: int, count: int, wires_elems, pub_inputs: np.ndarray, blinders: np.ndarray) -> int:
    """The C ABI takes bare pointers: a mis-shaped array would be read out of bounds.  Every size is checked against
    the key here (wires: count * 5 * n elements, pub_inputs: count * num_inputs, blinders: count * 13)."""
    n, num_inputs, _ = plonk_key_info(pk_handle)
    if count < 1:
        raise CapGpuError(-1, f"count must be >= 1, got {count}")
    if wires_elems != count * NUM_WIRE_TYPES * n:
        raise CapGpuError(-1, f"wires hold {wires_elems} field elements, key (n = {n}) needs count * 5 * n = "
                              f"{count * NUM_WIRE_TYPES * n}")
    if blinders.size != count * 13 * 4:
        raise CapGpuError(-1, f"blinders hold {blinders.size // 4} field elements, need count * 13 = {count * 13}")
    if pub_inputs.size % 4 or pub_inputs.size != count * num_inputs * 4:
        raise CapGpuError(-1, f"pub_inputs hold {pub_inputs.size / 4:g} field elements, key expects count * "
                              f"{num_inputs} = {count * num_inputs}")
    return num_inputs


def plonk_free_key(pk_handle: int):
    check(load().capgpu_plonk_free_key(ctypes.c_uint64(pk_handle)))


def _bytes_arg(b):
    if b is None:
        return None, 0
    buf = (ctypes.c_uint8 * len(b)).from_buffer_copy(bytes(b)) if len(b) else None
    return buf, len(b)


def plonk_prove_batch(pk_handle: int, wires: np.ndarray, pub_inputs: np.ndarray, blinders: np.ndarray,
                      ext_msg: bytes | None = None, count: int = 1, input_form=INPUT_EVALS):
    """wires (count, 5, n, 4), pub_inputs (count, l, 4), blinders (count, 13, 4), all Montgomery.  input_form =
    'coeffs': wires are the unblinded wire polynomials in coefficient form."""
    wires = np.ascontiguousarray(wires, dtype=np.uint64)
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1)
    blinders = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1)
    num_inputs = _check_prove_shapes(pk_handle, count, wires.size // 4 if wires.size % 4 == 0 else -1, pub_inputs,
                                     blinders)
    proofs = (Proof * count)()
    mbuf, mlen = _bytes_arg(ext_msg)
    pub_ptr = _p(pub_inputs) if pub_inputs.size else None
    check(load().capgpu_plonk_prove_batch_ex(ctypes.c_uint64(pk_handle), count, _p(wires.reshape(-1)), pub_ptr,
                                             ctypes.c_size_t(num_inputs), mbuf, ctypes.c_size_t(mlen), _p(blinders),
                                             ctypes.c_int(_form(input_form)), proofs))
    return list(proofs)


def plonk_prove(pk_handle: int, wires: np.ndarray, pub_inputs: np.ndarray, blinders: np.ndarray,
                ext_msg: bytes | None = None, input_form=INPUT_EVALS) -> Proof:
    """capgpu_plonk_prove: ONE proof per call - the entry point many host threads call at once (it coalesces their
    calls into device batches when plonk_set_coalescing is on).  wires (5, n, 4), pub_inputs (l, 4), blinders (13, 4)."""
    wires = np.ascontiguousarray(wires, dtype=np.uint64)
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1)
    blinders = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1)
    num_inputs = _check_prove_shapes(pk_handle, 1, wires.size // 4 if wires.size % 4 == 0 else -1, pub_inputs, blinders)
    proof = Proof()
    mbuf, mlen = _bytes_arg(ext_msg)
    check(load().capgpu_plonk_prove_ex(ctypes.c_uint64(pk_handle), _p(wires.reshape(-1)),
                                       _p(pub_inputs) if pub_inputs.size else None, ctypes.c_size_t(num_inputs), mbuf,
                                       ctypes.c_size_t(mlen), _p(blinders), ctypes.c_int(_form(input_form)),
                                       ctypes.byref(proof)))
    return proof


def has_lagrange_commit() -> bool:
    """the library can take round 1's wire commitments from the witness columns (Lagrange-form commit key per proving key)"""
    return hasattr(load(), "capgpu_plonk_set_wire_commit")


def plonk_set_wire_commit_from_evals(on):
    """True: wire commitments as MSMs of the witness VALUES on the key's Lagrange-form commit key; False: from the
    coefficients (jf-plonk's way); None: the library default.  Same group elements, same proof bytes."""
    if not has_lagrange_commit():
        return
    check(load().capgpu_plonk_set_wire_commit(ctypes.c_int(-1 if on is None else (1 if on else 0))))


def plonk_set_coalescing(window_us: int, max_batch: int = 0):
    check(load().capgpu_plonk_set_coalescing(ctypes.c_uint32(window_us), ctypes.c_uint32(max_batch)))


def plonk_graph_stats():
    """(segments captured, segments replayed) of the small-batch hipGraph path since process start"""
    a, b = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load().capgpu_plonk_graph_stats(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def plonk_coalescing_stats():
    b, p = ctypes.c_uint64(0), ctypes.c_uint64(0)
    check(load().capgpu_plonk_coalescing_stats(ctypes.byref(b), ctypes.byref(p)))
    return b.value, p.value


def plonk_prove_batch_dev(pk_handle: int, d_wires: DevBuf, pub_inputs: np.ndarray, blinders: np.ndarray,
                          ext_msg: bytes | None = None, count: int = 1, input_form=INPUT_EVALS):
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1)
    blinders = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1)
    num_inputs = _check_prove_shapes(pk_handle, count, d_wires.nbytes // 32 if d_wires.nbytes % 32 == 0 else -1,
                                     pub_inputs, blinders)
    proofs = (Proof * count)()
    mbuf, mlen = _bytes_arg(ext_msg)
    pub_ptr = _p(pub_inputs) if pub_inputs.size else None
    check(load().capgpu_plonk_prove_batch_dev_ex(ctypes.c_uint64(pk_handle), count, d_wires.ptr, pub_ptr,
                                                 ctypes.c_size_t(num_inputs), mbuf, ctypes.c_size_t(mlen),
                                                 _p(blinders), ctypes.c_int(_form(input_form)), proofs))
    return list(proofs)


def plonk_prove_multi(pk_handles, wires, pub_rows: np.ndarray, blinders: np.ndarray, ext_msgs=None,
                      input_form=INPUT_EVALS):
    """Proofs of several proving keys (one domain size, one SRS) in one device batch: pk_handles[i] is proof i's key.
    wires: (count, 5, n, 4) numpy array or a DevBuf of that content; pub_rows: (count, max_inputs, 4) - a key with fewer
    public inputs uses the first of its row; ext_msgs: one bytes object per proof, or None."""
    count = len(pk_handles)
    shapes = [plonk_key_info(h) for h in pk_handles]
    n = shapes[0][0]
    max_in = max(sh[1] for sh in shapes)
    pub_rows = np.ascontiguousarray(pub_rows, dtype=np.uint64).reshape(-1)
    blinders = np.ascontiguousarray(blinders, dtype=np.uint64).reshape(-1)
    if any(sh[0] != n for sh in shapes):
        raise ValueError("plonk_prove_multi: the keys of one batch must share the domain size")
    if pub_rows.size != count * max_in * 4 or blinders.size != count * 13 * 4:
        raise ValueError(f"plonk_prove_multi: pub_rows must hold {count} x {max_in} and blinders {count} x 13 elements")
    handles = (ctypes.c_uint64 * count)(*pk_handles)
    msgs_arg = lens_arg = None
    keep = []
    if ext_msgs is not None:
        if len(ext_msgs) != count:
            raise ValueError("plonk_prove_multi: one message per proof")
        msgs_arg = (ctypes.c_char_p * count)()
        lens_arg = (ctypes.c_size_t * count)()
        for i, m in enumerate(ext_msgs):
            keep.append(bytes(m) if m else b"")
            msgs_arg[i] = keep[-1] if keep[-1] else None
            lens_arg[i] = len(keep[-1])
    proofs = (Proof * count)()
    pub_ptr = _p(pub_rows) if pub_rows.size else None
    if isinstance(wires, DevBuf):
        if wires.nbytes != count * NUM_WIRE_TYPES * n * 32:
            raise ValueError("plonk_prove_multi: the wire buffer does not hold count x 5 x n elements")
        check(load().capgpu_plonk_prove_multi_dev_ex(handles, count, wires.ptr, pub_ptr, ctypes.c_size_t(max_in),
                                                     msgs_arg, lens_arg, _p(blinders), ctypes.c_int(_form(input_form)),
                                                     proofs))
    else:
        wires = np.ascontiguousarray(wires, dtype=np.uint64).reshape(-1)
        if wires.size != count * NUM_WIRE_TYPES * n * 4:
            raise ValueError("plonk_prove_multi: wires must hold count x 5 x n elements")
        check(load().capgpu_plonk_prove_multi_ex(handles, count, _p(wires), pub_ptr, ctypes.c_size_t(max_in), msgs_arg,
                                                 lens_arg, _p(blinders), ctypes.c_int(_form(input_form)), proofs))
    return list(proofs)


def proof_to_arrays(pr: Proof) -> dict:
    """ctypes Proof -> dict of numpy arrays (Montgomery words)."""
    def a(x):
        return np.ctypeslib.as_array(x).copy()
    return {
        "wires_poly_comms": a(pr.wires_poly_comms), "prod_perm_poly_comm": a(pr.prod_perm_poly_comm),
        "split_quot_poly_comms": a(pr.split_quot_poly_comms), "opening_proof": a(pr.opening_proof),
        "shifted_opening_proof": a(pr.shifted_opening_proof), "wires_evals": a(pr.wires_evals),
        "wire_sigma_evals": a(pr.wire_sigma_evals), "perm_next_eval": a(pr.perm_next_eval),
    }


# ---- verification (host only) -------------------------------------------------------------------------
def g2_generator() -> np.ndarray:
    out = np.zeros(16, dtype=np.uint64)
    check(load().capgpu_g2_generator(_p(out)))
    return out


def g2_mul(q: np.ndarray, scalar: int) -> np.ndarray:
    out = np.zeros(16, dtype=np.uint64)
    check(load().capgpu_g2_mul(_p(np.ascontiguousarray(q, dtype=np.uint64)), _limbs(scalar), _p(out)))
    return out


def pairing_check(g1_points: np.ndarray, g2_points: np.ndarray) -> bool:
    g1_points = np.ascontiguousarray(g1_points, dtype=np.uint64).reshape(-1, 8)
    g2_points = np.ascontiguousarray(g2_points, dtype=np.uint64).reshape(-1, 16)
    ok = ctypes.c_int(0)
    check(load().capgpu_pairing_check(_p(g1_points.reshape(-1)), _p(g2_points.reshape(-1)),
                                      ctypes.c_size_t(g1_points.shape[0]), ctypes.byref(ok)))
    return bool(ok.value)


def plonk_verify(vk: VerifyingKey, g2_h: np.ndarray, g2_beta_h: np.ndarray, pub_inputs: np.ndarray, proof: Proof,
                 ext_msg: bytes | None = None) -> bool:
    pub_inputs = np.ascontiguousarray(pub_inputs, dtype=np.uint64).reshape(-1)
    mbuf, mlen = _bytes_arg(ext_msg)
    ok = ctypes.c_int(0)
    check(load().capgpu_plonk_verify(ctypes.byref(vk), _p(np.ascontiguousarray(g2_h, dtype=np.uint64)),
                                     _p(np.ascontiguousarray(g2_beta_h, dtype=np.uint64)),
                                     _p(pub_inputs) if pub_inputs.size else None, ctypes.c_size_t(pub_inputs.size // 4),
                                     ctypes.byref(proof), mbuf, ctypes.c_size_t(mlen), ctypes.byref(ok)))
    return bool(ok.value)


def plonk_batch_verify(vks, g2_h: np.ndarray, g2_beta_h: np.ndarray, pub_inputs_list, proofs, ext_msgs=None,
                       on_device: bool = False) -> bool:
    """One pairing product for all proofs.  on_device: the group arithmetic (two MSMs over ~35 terms per proof) runs on
    the GPU through the prover's MSM kernels (capgpu_plonk_batch_verify_dev; needs init()), otherwise on host threads."""
    cnt = len(proofs)
    vk_arr = (ctypes.POINTER(VerifyingKey) * cnt)(*[ctypes.pointer(v) for v in vks])
    pubs = [np.ascontiguousarray(p, dtype=np.uint64).reshape(-1) for p in pub_inputs_list]
    pub_arr = (u64p * cnt)(*[(_p(p) if p.size else None) for p in pubs])
    nin = (ctypes.c_size_t * cnt)(*[p.size // 4 for p in pubs])
    pr_arr = (ctypes.POINTER(Proof) * cnt)(*[ctypes.pointer(p) for p in proofs])
    msgs = [(m if m is not None else b"") for m in (ext_msgs or [None] * cnt)]
    bufs = [(ctypes.c_uint8 * max(len(m), 1)).from_buffer_copy(m + (b"\0" if not m else b"")) for m in msgs]
    msg_arr = (ctypes.POINTER(ctypes.c_uint8) * cnt)(*[ctypes.cast(b, ctypes.POINTER(ctypes.c_uint8)) for b in bufs])
    len_arr = (ctypes.c_size_t * cnt)(*[len(m) for m in msgs])
    ok = ctypes.c_int(0)
    fn = load().capgpu_plonk_batch_verify_dev if on_device else load().capgpu_plonk_batch_verify
    check(fn(vk_arr, _p(np.ascontiguousarray(g2_h, dtype=np.uint64)),
             _p(np.ascontiguousarray(g2_beta_h, dtype=np.uint64)), pub_arr, nin, pr_arr,
             msg_arr, len_arr, ctypes.c_size_t(cnt), ctypes.byref(ok)))
    return bool(ok.value)


def proof_serialize(proof: Proof) -> bytes:
    buf = (ctypes.c_uint8 * 1024)()
    n = ctypes.c_size_t(0)
    check(load().capgpu_proof_serialize(ctypes.byref(proof), buf, ctypes.c_size_t(1024), ctypes.byref(n)))
    return bytes(buf[:n.value])


def proof_deserialize(data: bytes):
    """-> (Proof, bytes consumed); raises CapGpuError(CAPGPU_ERR_SERIALIZATION) on a malformed encoding."""
    pr, used = Proof(), ctypes.c_size_t(0)
    buf = (ctypes.c_uint8 * max(len(data), 1)).from_buffer_copy(data if data else b"\0")
    check(load().capgpu_proof_deserialize(buf, ctypes.c_size_t(len(data)), ctypes.byref(pr), ctypes.byref(used)))
    return pr, used.value


# ---- on-disk parameter formats (include/capgpu.h) ----------------------------------------
CAPGPU_ERR_SERIALIZATION = -8


def _u8buf(b: bytes):
    return (ctypes.c_uint8 * max(len(b), 1)).from_buffer_copy(b if b else b"\0")


def _opt_words(a, n):
    return _p(np.ascontiguousarray(a, dtype=np.uint64).reshape(n)) if a is not None else None


def g1_decompress(data: bytes) -> np.ndarray:
    """n x 32 bytes -> (n, 8) affine Montgomery words, (0, 0) = infinity."""
    n = len(data) // 32
    out = np.zeros((n, 8), dtype=np.uint64)
    check(load().capgpu_g1_decompress(_u8buf(data), ctypes.c_size_t(n), _p(out.reshape(-1)) if n else None))
    return out


def g1_compress(points: np.ndarray) -> bytes:
    points = np.ascontiguousarray(points, dtype=np.uint64).reshape(-1, 8)
    n = points.shape[0]
    out = (ctypes.c_uint8 * max(32 * n, 1))()
    check(load().capgpu_g1_compress(_p(points.reshape(-1)) if n else None, ctypes.c_size_t(n), out))
    return bytes(out[:32 * n])


def srs_deserialize(data: bytes, max_degree: int = 0):
    """-> (handle, h, beta_h, consumed)"""
    handle, used = ctypes.c_uint64(0), ctypes.c_size_t(0)
    h, bh = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    check(load().capgpu_srs_deserialize(_u8buf(data), ctypes.c_size_t(len(data)), ctypes.c_size_t(max_degree),
                                        ctypes.byref(handle), _p(h), _p(bh), ctypes.byref(used)))
    return handle.value, h, bh, used.value


def _sized_call(fn, *head):
    n = ctypes.c_size_t(0)
    check(fn(*head, None, ctypes.c_size_t(0), ctypes.byref(n)))
    buf = (ctypes.c_uint8 * max(n.value, 1))()
    check(fn(*head, buf, ctypes.c_size_t(n.value), ctypes.byref(n)))
    return bytes(buf[:n.value])


def srs_serialize(handle: int, h: np.ndarray, beta_h: np.ndarray) -> bytes:
    return _sized_call(load().capgpu_srs_serialize, ctypes.c_uint64(handle), _opt_words(h, 16), _opt_words(beta_h, 16))


def plonk_vk_serialize(vk: VerifyingKey, g: np.ndarray, h: np.ndarray, beta_h: np.ndarray, gamma_g=None) -> bytes:
    return _sized_call(load().capgpu_plonk_vk_serialize, ctypes.byref(vk), _opt_words(g, 8), _opt_words(gamma_g, 8),
                       _opt_words(h, 16), _opt_words(beta_h, 16))


def plonk_vk_deserialize(data: bytes):
    """-> (vk, g, gamma_g, h, beta_h, consumed)"""
    vk, used = VerifyingKey(), ctypes.c_size_t(0)
    g, gg = np.zeros(8, dtype=np.uint64), np.zeros(8, dtype=np.uint64)
    h, bh = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    check(load().capgpu_plonk_vk_deserialize(_u8buf(data), ctypes.c_size_t(len(data)), ctypes.byref(vk), _p(g), _p(gg),
                                             _p(h), _p(bh), ctypes.byref(used)))
    return vk, g, gg, h, bh, used.value


def plonk_key_serialize(pk_handle: int, h: np.ndarray, beta_h: np.ndarray, gamma_g=None) -> bytes:
    return _sized_call(load().capgpu_plonk_key_serialize, ctypes.c_uint64(pk_handle), _opt_words(gamma_g, 8),
                       _opt_words(h, 16), _opt_words(beta_h, 16))


def plonk_key_deserialize(data: bytes):
    """-> (srs_handle, pk_handle, vk, h, beta_h, consumed)"""
    srs, pk, used = ctypes.c_uint64(0), ctypes.c_uint64(0), ctypes.c_size_t(0)
    vk = VerifyingKey()
    h, bh = np.zeros(16, dtype=np.uint64), np.zeros(16, dtype=np.uint64)
    check(load().capgpu_plonk_key_deserialize(_u8buf(data), ctypes.c_size_t(len(data)), ctypes.byref(srs),
                                              ctypes.byref(pk), ctypes.byref(vk), _p(h), _p(bh), ctypes.byref(used)))
    return srs.value, pk.value, vk, h, bh, used.value
