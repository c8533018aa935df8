"""A/B: one device context against two on the same GPU (CAPGPU_CONTEXTS_PER_DEVICE=2): the batch halves are proved
concurrently from two host threads, each on its own stream, so that one half's latency-bound launches and host transcript
phases run under the other half's issue-bound kernels.  Usage: python tools/gpu_two_ctx.py [log_n] [P]"""
import json
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
K = int(os.environ.get("CTX", "2"))
os.environ["CAPGPU_CONTEXTS_PER_DEVICE"] = str(K)
from cap_amd import bench_utils as bu  # noqa: E402
from cap_amd import lib as cg  # noqa: E402

log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 15
P = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(os.environ.get("STEPS", "6"))
cg.init(0)
assert cg.device_count() == K
tau = bu.SplitMix64(0xCA9).field()
n, ni = 1 << log_n, 27
srs = cg.srs_generate(tau, n + 3)
sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
pk, vk = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
wit = [sc.witness(3 + i) for i in range(4)]
wires = np.stack([sc.wires_mont(wit[i % 4][0]) for i in range(P)])
pubs = np.stack([bu.to_mont_array(wit[i % 4][1]) for i in range(P)])
blind = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
msg = bytes(range(32))
out = {"log_n": log_n, "P": P, "contexts": K}


def run(parts):
    """parts contexts, each proving P / parts proofs per step from its own thread"""
    res = [None] * parts
    bufs = [None] * parts
    bar = threading.Barrier(parts + 1)
    times = {}

    def worker(i):
        cg.set_device(i)
        lo, hi = P * i // parts, P * (i + 1) // parts
        bufs[i] = cg.DevBuf.from_numpy(wires[lo:hi])
        for _ in range(2):
            cg.plonk_prove_batch_dev(pk, bufs[i], pubs[lo:hi], blind[lo:hi], msg, hi - lo)
        bar.wait()
        for _ in range(steps):
            res[i] = cg.plonk_prove_batch_dev(pk, bufs[i], pubs[lo:hi], blind[lo:hi], msg, hi - lo)
        bar.wait()
        bufs[i].free()

    th = [threading.Thread(target=worker, args=(i,)) for i in range(parts)]
    for t in th:
        t.start()
    bar.wait()
    t0 = time.perf_counter()
    bar.wait()
    dt = time.perf_counter() - t0
    for t in th:
        t.join()
    return P * steps / dt, [bytes(p) for r in res for p in r]


r1, p1 = run(1)
out["one_context_proofs_per_s"] = r1
for parts in range(2, K + 1):
    r, p = run(parts)
    out[f"{parts}_contexts_proofs_per_s"] = r
    out[f"{parts}_contexts_same_proofs"] = p == p1
r1b, _ = run(1)
out["one_context_again"] = r1b
print(json.dumps(out))
