#!/bin/bash
# What clock do the hot kernels really run at, and how much of a launch do the CUs sit idle in its tail?
# GRBM_GUI_ACTIVE (chip-level active cycles) / SQ_BUSY_CU_CYCLES (per-CU busy) against the launch duration of the same
# dispatch (--kernel-trace in the same pass; no other trace domain).  One counter per pass.
tag=$1
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd $R
mkdir -p gpurun_out/clock_$tag
for ctr in GRBM_GUI_ACTIVE GRBM_COUNT SQ_BUSY_CU_CYCLES SQ_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY; do
  timeout 240 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d gpurun_out/clock_$tag/$ctr -- python3 bench.py --one-context --steps 1 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/clock_$tag/$ctr.err
done
# the two counters of the busy share in ONE pass (the same dispatches, the same clock)
timeout 240 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES --kernel-trace --output-format csv -d gpurun_out/clock_$tag/PAIR -- python3 bench.py --one-context --steps 1 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/clock_$tag/PAIR.err
python3 - <<PY
import csv, glob, collections, json
out = {}
def short(name):
    return name.replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
pd = "gpurun_out/clock_$tag/PAIR/"
dur = {}
for f in glob.glob(pd + "**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
pagg = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(pd + "**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        t = dur.get(r["Dispatch_Id"])
        if t is None or t < 200000:
            continue
        a = pagg[short(r["Kernel_Name"])]
        a[r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            a["ns"] += t
            a["launches"] += 1
out["PAIR"] = {k: {"launches": int(v["launches"]), "GRBM_GUI_ACTIVE": v["GRBM_GUI_ACTIVE"], "SQ_BUSY_CU_CYCLES": v["SQ_BUSY_CU_CYCLES"],
                   "GRBM_GUI_ACTIVE_per_ns": round(v["GRBM_GUI_ACTIVE"] / v["ns"], 4)}
               for k, v in pagg.items() if v["ns"] and v["GRBM_GUI_ACTIVE"] and v["SQ_BUSY_CU_CYCLES"]}
print("PAIR", {k[:24]: (round(v["GRBM_GUI_ACTIVE_per_ns"] / 8, 3), round(v["SQ_BUSY_CU_CYCLES"] / 256 / (v["GRBM_GUI_ACTIVE"] / 8), 3)) for k, v in list(out["PAIR"].items())[:12]})
for d in sorted(glob.glob("gpurun_out/clock_$tag/*/")):
    ctr = d.rstrip("/").split("/")[-1]
    if ctr == "PAIR":
        continue
    dur = {}
    for f in glob.glob(d + "**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Dispatch_Id"]] = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
    agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
    for f in glob.glob(d + "**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
            t = dur.get(r["Dispatch_Id"])
            if t is None or t < 200000:
                continue
            a = agg[k]
            a[0] += float(r["Counter_Value"]); a[1] += t; a[2] += 1
    out[ctr] = {k: {"launches": v[2], "counter_per_ns": round(v[0] / v[1], 4), "avg_ms": round(v[1] / v[2] / 1e6, 3)} for k, v in agg.items() if v[2]}
    top = sorted(out[ctr], key=lambda k: -out[ctr][k]["avg_ms"] * out[ctr][k]["launches"])[:8]
    print(ctr, {k[:24]: out[ctr][k] for k in top})
json.dump(out, open("gpurun_out/clock_$tag/clock.json", "w"), indent=1)
PY
find gpurun_out/clock_$tag -name "*.csv" -size +2M -delete
