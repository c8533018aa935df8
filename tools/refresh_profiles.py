"""Copy the outputs of tools/gpujob.sh final TAG (gpurun_out/final_TAG, gpurun_out/prof_TAG) into profiles/ under the
round's names.  Run here after the gpurun call:  python tools/refresh_profiles.py TAG [ROUND]"""
import ast
import glob
import json
import os
import shutil
import sys

tag = sys.argv[1]
rnd = sys.argv[2] if len(sys.argv) > 2 else "r04"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src, prof, dst = (os.path.join(root, "gpurun_out", "final_" + tag), os.path.join(root, "gpurun_out", "prof_" + tag),
                  os.path.join(root, "profiles"))


def one_line_json(path_in, path_out):
    """bench.py prints one JSON line on stdout; keep exactly that line"""
    lines = [ln for ln in open(path_in).read().splitlines() if ln.startswith("{")]
    open(path_out, "w").write(lines[-1] + "\n")
    return json.loads(lines[-1])


b = one_line_json(os.path.join(src, "bench.json"), os.path.join(dst, f"bench_{rnd}_final.json"))
b20 = one_line_json(os.path.join(src, "bench_20steps.json"), os.path.join(dst, f"bench_20steps_{rnd}.json"))
bm = one_line_json(os.path.join(src, "bench_mixed64.json"), os.path.join(dst, f"bench_mixed64_{rnd}.json"))
shutil.copy(os.path.join(src, "msm_single_profile.json"), os.path.join(dst, f"msm_single_profile_{rnd}.json"))
for name, out_name in (("bench_single_process.json", f"bench_single_process_dev0x2_{rnd}.json"),
                       ("msm_deep_ab.jsonl", f"msm_deep_ab_{rnd}.jsonl"), ("two_ctx.json", f"two_contexts_ab_{rnd}.json"),
                       ("latency_ab.jsonl", f"latency_ab_final_{rnd}.jsonl")):
    if os.path.exists(os.path.join(src, name)) and os.path.getsize(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out_name))
shutil.copy(os.path.join(src, "traffic.json"), os.path.join(dst, f"traffic_{rnd}.json"))
shutil.copy(os.path.join(src, "pytest_gpu.txt"), os.path.join(dst, f"pytest_gpu_{rnd}_final.txt"))
w2 = [open(os.path.join(src, n)).read().strip() for n in ("bench_w2.json", "bench_w2_mixed.json")]
open(os.path.join(dst, f"bench_world2_gloo_sameGPU_{rnd}.jsonl"), "w").write("\n".join(w2) + "\n")
os.makedirs(os.path.join(dst, "rocprof_" + rnd), exist_ok=True)
shutil.copy(os.path.join(prof, "summary.txt"), os.path.join(dst, "rocprof_" + rnd, "summary.txt"))
stats = glob.glob(os.path.join(prof, "trace", "**", "*kernel_stats.csv"), recursive=True)
shutil.copy(stats[0], os.path.join(dst, "rocprof_" + rnd, "kernel_stats.csv"))

# clock / busy-CU pass (tools/gpujob.sh clock): counter per nanosecond of the same dispatch -> clock and busy share.
# "PAIR" (round 4): GRBM_GUI_ACTIVE and SQ_BUSY_CU_CYCLES collected in ONE pass, so that the busy share is a ratio of two
# counters of the same dispatches (separate passes may run at different clocks: round 4's gave 1.04 for msm_accumulate).
if os.path.exists(os.path.join(src, "clock.json")):
    raw = json.load(open(os.path.join(src, "clock.json")))
    pair = raw.get("PAIR", {})
    derived = {}
    for k, v in raw.get("GRBM_GUI_ACTIVE", {}).items():
        ghz = v["counter_per_ns"] / 8.0                       # the counter sums over the 8 XCDs
        e = {"clock_GHz": round(ghz, 3)}
        if k in pair:
            e = {"clock_GHz": round(pair[k]["GRBM_GUI_ACTIVE_per_ns"] / 8.0, 3),
                 "cu_busy_frac": round(pair[k]["SQ_BUSY_CU_CYCLES"] / 256.0 / (pair[k]["GRBM_GUI_ACTIVE"] / 8.0), 3),
                 "same_pass": True}
        else:
            bc = raw.get("SQ_BUSY_CU_CYCLES", {}).get(k)
            if bc and ghz > 0:
                e["cu_busy_frac"] = round(bc["counter_per_ns"] / 256.0 / ghz, 3)   # ... over the 256 CUs
        derived[k] = e
    json.dump({"source": "tools/gpujob.sh clock: rocprofv3 --pmc <counter(s)> --kernel-trace (no other trace domain) on "
                         "bench.py --one-context --steps 1 --warmup 1 --no-msm --no-extras, batch 256; counter value / "
                         "duration of the same dispatch, launches >= 0.2 ms only; entries marked same_pass come from the "
                         "pass that collected GRBM_GUI_ACTIVE and SQ_BUSY_CU_CYCLES together",
               "how_to_read": "GRBM_GUI_ACTIVE and GRBM_COUNT sum over the 8 XCDs: / 8 = clock in GHz.  SQ_BUSY_CU_CYCLES sums "
                              "over 256 CUs: / 256 against GRBM_GUI_ACTIVE / 8 = fraction of the launch a CU is busy.",
               "derived": derived, "raw_counter_per_ns": {k: v for k, v in raw.items() if k != "PAIR"}, "pair_pass": pair},
              open(os.path.join(dst, f"clock_{rnd}.json"), "w"), indent=1)

# instruction counters: tools/gpujob.sh insts prints  name {counter: (sum, launches), ...}  per kernel
kern = {}
for ln in open(os.path.join(src, "insts.txt")):
    if " {'SQ_" not in ln:
        continue
    name, rest = ln.split(" {", 1)
    d = ast.literal_eval("{" + rest.strip())
    launches = max(v[1] for v in d.values())
    kern[name.strip()] = dict(launches=launches, **{k: round(v[0] / v[1]) if isinstance(v[0], float) else v[0] for k, v in d.items()})
prev = os.path.join(dst, f"inst_counters_{rnd}.json")
old = json.load(open(prev if os.path.exists(prev) else os.path.join(dst, "inst_counters_r03.json")))
adds = b["alu_roofline"]["mixed_adds_per_step"] / 4           # four large msm_accumulate launches per step
acc = kern["msm_accumulate"]
# the PMC run has 8 large launches and the small one of preprocess; its counters are per-launch averages over all 9
per_add = acc["SQ_INSTS_VALU"] * acc["launches"] / (acc["launches"] - 1) * 64 / adds
old["msm_accumulate_valu_instructions_per_mixed_addition"] = int(round(per_add, -1))
old["kernels"] = kern
json.dump(old, open(os.path.join(dst, f"inst_counters_{rnd}.json"), "w"), indent=1)
print("bench", round(b["value"], 1), "20 steps", round(b20["value"], 1), "mixed64", round(bm["value"], 1),
      "instructions per addition", round(per_add))
print(open(os.path.join(src, "pytest_gpu.txt")).read().strip().splitlines()[-1])
