#!/bin/bash
# four contexts by default + two parts per device for dealt batches: multi-device / graph / input-form tests, then the bench line
OUT=gpurun_out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests/test_gpu_multidev.py tests/test_gpu_graphs.py tests/test_gpu_input_forms.py tests/test_gpu_comm.py tests/test_bench_launch.py -x -q -m gpu > $OUT/ctx4b_pytest.txt 2>&1; grep -E "passed|failed|error" $OUT/ctx4b_pytest.txt | tail -3
for r in 1; do
timeout 900 python bench.py --no-cpu-baseline --no-reference-schedule > $OUT/bench_ctx4b.json 2> $OUT/bench_ctx4b.err
python - <<PY
import json
b = json.load(open("$OUT/bench_ctx4b.json"))
print("contexts", b["config"]["device_contexts"], "headline", round(b["value"], 1), "one_ctx", round(b["one_context_profiled_pass"]["proofs_per_s"], 1), "lat1", round(b["latency_ms_batch1"]["median"], 3),
      "coalesced", round(b["coalesced_single_calls"]["proofs_per_s"], 1), "pcie", round(b["pcie_inclusive"]["proofs_per_s"], 1), round(b["pcie_inclusive_coeffs"]["proofs_per_s"], 1),
      "mixed", round(b["mixed64"]["one_batch_per_domain_proofs_per_s"], 1), round(b["mixed64"]["domains_on_two_contexts_proofs_per_s"], 1), "n2p16", round(b["n2p16"]["proofs_per_s"], 1))
PY
done
