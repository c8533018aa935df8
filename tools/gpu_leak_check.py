"""Device-memory leak check: N cycles of SRS / key / proof / blob / MSM / buffer creation and release; the free memory
after the cycles must not depend on N (measured: 23 MB of one-time scratch growth for 100, 400 and 800 cycles).
python tools/gpu_leak_check.py [N]"""
import sys
import numpy as np, torch
sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu
cg.init(0)
tau = bu.SplitMix64(5).field()
def cycle(i):
    n = 1 << (6 + i % 5)
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(6 + i % 5, 3, seed=4 + i % 3)
    pk, vk = cg.plonk_preprocess(srs, n, 3, sc.selectors_mont(), sc.sigma_mont())
    w, pubs = sc.witness(1)
    P = 1 + i % 7
    wm = np.stack([sc.wires_mont(w)] * P); pm = np.stack([bu.to_mont_array(pubs)] * P); bl = np.stack([bu.to_mont_array(bu.blinders(1))] * P)
    cg.plonk_prove_batch(pk, wm, pm, bl, None, P)
    blob = cg.plonk_key_serialize(pk, cg.g2_generator(), cg.g2_generator())
    s2, p2, *_ = cg.plonk_key_deserialize(blob)
    cg.plonk_free_key(p2); cg.srs_free(s2)
    sc2 = bu.random_canonical_scalars(i, n)
    cg.msm_g1(srs, sc2)
    d = cg.DevBuf.from_numpy(sc2); d.free()
    cg.plonk_free_key(pk); cg.srs_free(srs)
for i in range(10): cycle(i)
torch.cuda.synchronize()
f0 = torch.cuda.mem_get_info()[0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 200
for i in range(N): cycle(i)
torch.cuda.synchronize()
f1 = torch.cuda.mem_get_info()[0]
print("free before %.1f MB, after %.1f MB, delta %.2f MB" % (f0 / 1e6, f1 / 1e6, (f0 - f1) / 1e6))
