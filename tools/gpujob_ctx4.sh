#!/bin/bash
# The whole default bench line with 2 and with 4 device contexts on the one GPU  -> gpurun_out/bench_ctx{2,4}.json
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
for c in 2 4 2 4; do
  CAPGPU_CONTEXTS_PER_DEVICE=$c timeout 900 python bench.py --no-cpu-baseline --no-reference-schedule > $OUT/bench_ctx$c.json 2> $OUT/bench_ctx$c.err
  python - <<PY
import json
b = json.load(open("$OUT/bench_ctx$c.json"))
print("contexts=$c", "headline", round(b["value"], 1), "one_ctx", round(b["one_context_profiled_pass"]["proofs_per_s"], 1), "lat1", round(b["latency_ms_batch1"]["median"], 3),
      "coalesced", round(b["coalesced_single_calls"]["proofs_per_s"], 1), "pcie", round(b["pcie_inclusive"]["proofs_per_s"], 1), round(b["pcie_inclusive_coeffs"]["proofs_per_s"], 1),
      "mixed", round(b["mixed64"]["one_batch_per_domain_proofs_per_s"], 1), round(b["mixed64"]["domains_on_two_contexts_proofs_per_s"], 1), "n2p16", round(b["n2p16"]["proofs_per_s"], 1))
PY
done
