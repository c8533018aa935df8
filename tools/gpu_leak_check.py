"""Device-memory leak check: N cycles of SRS / key / proof / blob / MSM / buffer creation and release; the free memory
after the cycles must not depend on N (measured: 23 MB of one-time scratch growth for 100, 400 and 800 cycles).
python tools/gpu_leak_check.py [N]
(also run, bounded, by tests/test_gpu_fuzz.py)"""
import sys

import numpy as np

sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu  # noqa: E402


def cycle(i, tau):
    n = 1 << (6 + i % 5)
    srs = cg.srs_generate(tau, n + 3)
    sc = bu.synthetic_circuit(6 + i % 5, 3, seed=4 + i % 3)
    pk, vk = cg.plonk_preprocess(srs, n, 3, sc.selectors_mont(), sc.sigma_mont())
    w, pubs = sc.witness(1)
    P = 1 + i % 7
    wm = np.stack([sc.wires_mont(w)] * P)
    pm = np.stack([bu.to_mont_array(pubs)] * P)
    bl = np.stack([bu.to_mont_array(bu.blinders(1))] * P)
    cg.plonk_prove_batch(pk, wm, pm, bl, None, P)
    blob = cg.plonk_key_serialize(pk, cg.g2_generator(), cg.g2_generator())
    s2, p2, *_ = cg.plonk_key_deserialize(blob)
    cg.plonk_free_key(p2)
    cg.srs_free(s2)
    sc2 = bu.random_canonical_scalars(i, n)
    cg.msm_g1(srs, sc2)
    d = cg.DevBuf.from_numpy(sc2)
    d.free()
    cg.plonk_free_key(pk)
    cg.srs_free(srs)


def run(cycles=200, warm=10):
    """returns the device memory (bytes) that `cycles` create / use / free cycles left allocated"""
    cg.init(0)
    tau = bu.SplitMix64(5).field()
    for i in range(warm):
        cycle(i, tau)
    cg.sync()
    f0 = cg.mem_info()[0]
    for i in range(cycles):
        cycle(i, tau)
    cg.sync()
    f1 = cg.mem_info()[0]
    return f0 - f1


if __name__ == "__main__":
    delta = run(int(sys.argv[1]) if len(sys.argv) > 1 else 200)
    print("device memory still held after the cycles: %.2f MB" % (delta / 1e6))
