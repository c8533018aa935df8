"""profiles/traffic_r01.json from a tools/gpuprof.sh output directory: python tools/make_traffic.py gpurun_out/prof_X BATCH"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, batch = sys.argv[1], int(sys.argv[2])


def short(name):
    name = name.replace("(anonymous namespace)", "anon").split("(")[0]
    for p in ("void ", "cap::anon::", "cap::pk::", "cap::"):
        name = name.replace(p, "")
    return name.strip()


out = {}
for kind in ("fetch", "write"):
    f = glob.glob(f"{root}/pmc_{kind}/**/*counter_collection.csv", recursive=True)[0]
    agg = defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        agg[k][0] += float(r["Counter_Value"])
        agg[k][1] += 1
    for k, (v, c) in agg.items():
        out.setdefault(k, {})[kind + "_KB_per_launch"] = v / c
        out[k]["launches"] = c
keep = lambda k: k.startswith(("msm_", "ntt_", "k_quot"))
res = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no trace domains) on `bench.py --steps 2 "
              f"--warmup 1 --batch {batch} --no-msm`, MI355X, round 1 final kernels",
    "batch": batch,
    "note": "raw counter values are KB; bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 with NO read-side x2 correction "
            "(MI355X_MICROARCH.md: FETCH_SIZE under-reports wide coalesced streams by 2x; msm_accumulate reads are 64-B "
            "gathers from the 42 MB window table, uncalibrated, and Infinity-Cache hits are counted) - a lower bound",
    "per_launch_bytes": {k: (v.get("fetch_KB_per_launch", 0) + v.get("write_KB_per_launch", 0)) * 1024
                         for k, v in out.items() if keep(k)},
    "raw": {k: v for k, v in out.items() if keep(k)},
}
json.dump(res, open("profiles/traffic_r01.json", "w"), indent=1)
print({k: round(v / 1e6, 1) for k, v in res["per_launch_bytes"].items()})
