"""Probe: capgpu_comm_init for a world of two of which only this rank exists.  Prints what happens and how long it takes.
Run under `timeout` - an RCCL that blocks inside its bootstrap cannot be interrupted from Python."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("CAPGPU_COMM_TIMEOUT_MS", "1500")
os.environ.setdefault("CAPGPU_COMM_DEBUG", "1")
from cap_amd import lib as cg  # noqa: E402

cg.init(0)
t0 = time.time()
try:
    cg.comm_init(0, 2, cg.comm_unique_id())
    print("comm_init returned OK?!", flush=True)
except cg.CapGpuError as e:
    print(f"comm_init failed after {time.time() - t0:.2f} s: {e}", flush=True)
print("comm_info", cg.comm_info(), flush=True)
