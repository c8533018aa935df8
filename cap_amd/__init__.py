"""cap_amd - MI355X-native PLONK prover hot path for CAP notes.

The product is ``libcapgpu.so`` (hand-written HIP for gfx950 behind the C ABI in
``include/capgpu.h``).  This package is the thin host-side mirror of the
reference's proof API (``/root/reference/src/proof/{mod,transfer,mint,freeze}.rs``)
on top of that ABI.  Importing it never touches ``oracle/``; there is no CPU
fallback - without the built library or without a gfx950 device it raises.
"""
from .lib import CapGpuError, load, lib_path  # noqa: F401

__all__ = ["CapGpuError", "load", "lib_path"]
