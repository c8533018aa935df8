"""Times the on-disk parameter path at the size of the reference's CRS (2^17 + 3 powers): serialise, deserialise,
and the decompression kernel alone (HIP events via the library's profiler)."""
import json
import sys
import time

sys.path.insert(0, ".")
from cap_amd import lib as cg  # noqa: E402

cg.init(0)
n = (1 << 17) + 3
tau = 0x1234567890ABCDEF1234567
h = cg.srs_generate(tau, n)
g2 = cg.g2_generator()
bh = cg.g2_mul(g2, tau)
t0 = time.perf_counter(); blob = cg.srs_serialize(h, g2, bh); t1 = time.perf_counter()
cg.srs_deserialize(blob)  # warm
cg.profile_enable(True); cg.profile_reset()
t2 = time.perf_counter(); h2, _, _, _ = cg.srs_deserialize(blob); t3 = time.perf_counter()
st = cg.profile_stats()
cg.profile_enable(False)
t4 = time.perf_counter(); pts = cg.g1_decompress(blob[8:8 + 32 * n]); t5 = time.perf_counter()
out = {"points": n, "blob_bytes": len(blob), "serialize_ms": (t1 - t0) * 1e3, "deserialize_ms": (t3 - t2) * 1e3,
       "g1_decompress_call_ms": (t5 - t4) * 1e3,
       "kernels_ms": {k: v[0] for k, v in st.items()}}
print(json.dumps(out))
