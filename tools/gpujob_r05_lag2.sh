#!/bin/bash
# round 5: (1) headline batch (uniform witness), wire commitments from coefficients vs from evaluations, same box, alternating;
# (2) what the Lagrange-form key costs to build; (3) the whole GPU suite with the new default
OUT=gpurun_out/r05_lag2
mkdir -p $OUT
AB="--steps 6 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-extras --no-msm --no-mixed"
CAPGPU_WIRE_COMMIT=coeffs python bench.py $AB > $OUT/coeffs_a.json 2> $OUT/coeffs_a.err
python bench.py $AB > $OUT/evals_a.json 2> $OUT/evals_a.err
CAPGPU_WIRE_COMMIT=coeffs python bench.py $AB > $OUT/coeffs_b.json 2> $OUT/coeffs_b.err
python bench.py $AB > $OUT/evals_b.json 2> $OUT/evals_b.err
python - <<PY
import json, time
for n in ("coeffs_a", "evals_a", "coeffs_b", "evals_b"):
    try:
        d = json.load(open("$OUT/%s.json" % n))
        print(n, round(d["value"], 1), "proofs/s; one ctx", round(d.get("one_context_profiled_pass", {}).get("proofs_per_s", 0), 1),
              {k: round(v / d["top_kernels_steps"], 2) for k, v in d["top_kernels_ms"].items()})
    except Exception as e:
        print(n, "failed", e, open("$OUT/%s.err" % n).read()[-1500:])
import numpy as np
from cap_amd import bench_utils as bu, lib as cg
cg.init(0)
tau = bu.SplitMix64(0xCA9).field()
for ln in (10, 14, 15, 16, 17):
    n = 1 << ln
    h = cg.srs_generate(tau, n + 3)
    sc = np.zeros((n + 2, 4), np.uint64); sc[:, 0] = 1
    f0, _ = cg.mem_info()
    cg.profile_reset(); cg.profile_enable(True)
    t0 = time.perf_counter(); cg.lagrange_commit(h, ln, sc); t1 = time.perf_counter(); cg.lagrange_commit(h, ln, sc); t2 = time.perf_counter()
    st = cg.profile_stats(); cg.profile_enable(False)
    f1, _ = cg.mem_info()
    print("lagrange key 2^%d: first call %.1f ms (build + MSM), second %.2f ms; device memory %.1f MB; kernels ms:" % (ln, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (f0 - f1) / 1e6),
          {k: round(v[0], 2) for k, v in st.items() if k.startswith("lag_") or k == "msm_precompute_kernel"})
    cg.srs_free(h)
PY
timeout 2400 python -m pytest tests -m gpu -x -q > $OUT/pytest_gpu.txt 2>&1
tail -8 $OUT/pytest_gpu.txt
