"""Issue rates of the instruction classes (capgpu_ubench_issue_rates) as a function of how long one measurement launch
runs: a sub-millisecond burst runs at the boost clock, the prover's 30 ms launches at whatever clock the power limit
sustains - the honest ceiling for them is the sustained rate."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cap_amd import lib as cg  # noqa: E402

cg.init(0)
for iters in (1000, 5000, 20000, 60000):
    os.environ["CAPGPU_UBENCH_ITERS"] = str(iters)
    r = cg.ubench_issue_rates()
    print(json.dumps({"iters": iters, "launch_ms_mad": round(256 * 8 * 256 * iters * 64 * 64 / r["v_mad_u64_u32"] * 1e3, 2),
                      "T_lane_ops_per_s": {k: round(v / 1e12, 2) for k, v in r.items()}}), flush=True)
