"""The C-ABI library loads on a machine without a GPU, exports every symbol include/capgpu.h declares,
and fails loudly (no CPU fallback) when asked to compute.  (`-m "not gpu"`)"""
import ctypes
import os
import re

import numpy as np
import pytest

from cap_amd import lib as cg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "capgpu.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(capgpu_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_boundary():
    syms = declared_symbols()
    for must in ("capgpu_init", "capgpu_msm_g1", "capgpu_msm_g1_batch", "capgpu_ntt_fr", "capgpu_ntt_fr_batch",
                 "capgpu_srs_upload", "capgpu_plonk_preprocess", "capgpu_plonk_prove", "capgpu_last_error"):
        assert must in syms


def test_library_exports_every_declared_symbol():
    L = cg.load()
    for name in declared_symbols():
        assert hasattr(L, name), f"{name} declared in include/capgpu.h but not exported by libcapgpu.so"


def test_rust_bindings_cover_the_header():
    """bindings/capgpu-sys is unbuilt source (no Rust toolchain here): at least it must name exactly the symbols the
    header declares and the error codes the header defines, so that it cannot drift unnoticed."""
    rs = open(os.path.join(ROOT, "bindings", "capgpu-sys", "src", "lib.rs")).read()
    bound = set(re.findall(r"pub fn (capgpu_[a-z0-9_]+)\s*\(", rs))
    declared = set(declared_symbols())
    assert declared - bound == set(), f"declared in capgpu.h, missing in lib.rs: {sorted(declared - bound)}"
    assert bound - declared == set(), f"bound in lib.rs, not in capgpu.h: {sorted(bound - declared)}"
    hdr = open(os.path.join(ROOT, "include", "capgpu.h")).read()
    for name, val in re.findall(r"#define (CAPGPU_(?:OK|ERR_[A-Z_]+)) \(?(-?\d+)\)?", hdr):
        assert re.search(rf"pub const {name}: c_int = {val};", rs), name


def test_struct_layouts_match_header():
    assert ctypes.sizeof(cg.Proof) == 13 * 64 + 10 * 32
    assert ctypes.sizeof(cg.VerifyingKey) == 16 + 5 * 32 + 18 * 64


def test_no_silent_fallback_without_device():
    """On a box without a GPU every compute entry point must refuse; on the GPU box this test only checks
    the not-initialised path before anything else has called capgpu_init in this process."""
    from tests import helpers as H
    L = cg.load()
    if H.gpu_present():
        pytest.skip("GPU present: the refusal path is covered on the CPU-only runner")
    rc = L.capgpu_init(None, 0)
    assert rc == -2 and b"no CPU fallback" in L.capgpu_last_error()
    data = np.zeros(4 * 8, dtype=np.uint64)
    assert L.capgpu_ntt_fr(data.ctypes.data_as(cg.u64p), ctypes.c_uint32(3), 0, 0) == -6
    out = np.zeros(12, dtype=np.uint64)
    assert L.capgpu_msm_g1(ctypes.c_uint64(1), ctypes.c_size_t(0), data.ctypes.data_as(cg.u64p), ctypes.c_size_t(1),
                           out.ctypes.data_as(cg.u64p)) == -6
    with pytest.raises(cg.CapGpuError):
        cg.init(0)
    with pytest.raises(cg.CapGpuError):
        cg.ntt_fr(data, 3)
    from cap_amd import proof
    with pytest.raises(proof.TxnApiError):
        proof.universal_setup(10, 5)


def test_product_does_not_import_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/."""
    pkg = os.path.join(ROOT, "cap_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                txt = open(os.path.join(dirpath, f), errors="ignore").read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M), f
                assert "capref" not in txt, f
                assert "/root/reference" not in txt or f.endswith(".py"), f
