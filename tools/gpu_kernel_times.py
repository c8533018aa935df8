"""Every kernel's time per step of the headline workload (library HIP-event profiler), optionally for a second library
build on the same box:  python tools/gpu_kernel_times.py [other_lib.so]"""
import json
import os
import subprocess
import sys

CODE = r'''
import json, sys
import numpy as np
sys.path.insert(0, ".")
from cap_amd import lib as cg, bench_utils as bu
cg.init(0)
P, log_n, ni = 256, 15, 27
n = 1 << log_n
srs = cg.srs_generate(bu.SplitMix64(0xCA9).field(), n + 3)
sc = bu.synthetic_circuit(log_n, ni, seed=2 + log_n + ni)
pk, vk = cg.plonk_preprocess(srs, n, ni, sc.selectors_mont(), sc.sigma_mont())
wit = [sc.witness(3 + i) for i in range(4)]
wires = np.stack([sc.wires_mont(wit[i % 4][0]) for i in range(P)])
pubs = np.stack([bu.to_mont_array(wit[i % 4][1]) for i in range(P)])
bl = np.stack([bu.to_mont_array(bu.blinders(7000 + i)) for i in range(P)])
d = cg.DevBuf.from_numpy(wires)
for _ in range(2): cg.plonk_prove_batch_dev(pk, d, pubs, bl, b"x" * 32, P)
cg.profile_enable(True); cg.profile_reset()
S = 3
for _ in range(S): cg.plonk_prove_batch_dev(pk, d, pubs, bl, b"x" * 32, P)
st = cg.profile_stats()
print(json.dumps({k: round(v[0] / S, 3) for k, v in sorted(st.items(), key=lambda kv: -kv[1][0])}))
'''
res = {}
for name, lib in [("in-tree", None)] + [(a, a) for a in sys.argv[1:]]:
    env = dict(os.environ)
    if lib:
        env["CAPGPU_LIBRARY"] = os.path.abspath(lib)
    out = subprocess.run([sys.executable, "-c", CODE], capture_output=True, text=True, env=env)
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    res[name] = json.loads(line[-1]) if line else {"error": out.stderr[-400:]}
names = list(res)
keys = list(res[names[0]])
print("%-28s" % "kernel (ms per step)", *["%12s" % nm[-12:] for nm in names])
for k in keys:
    print("%-28s" % k[:28], *["%12s" % res[nm].get(k, "-") for nm in names])
print("%-28s" % "sum", *["%12.2f" % sum(v for v in res[nm].values() if isinstance(v, float)) for nm in names])
