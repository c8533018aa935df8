"""profiles/traffic_rNN.json from a tools/gpuprof.sh output directory:
    python tools/make_traffic.py gpurun_out/prof_X BATCH [profiles/traffic_r02.json]"""
import csv
import glob
import json
import sys
from collections import defaultdict

root, batch = sys.argv[1], int(sys.argv[2])
dest = sys.argv[3] if len(sys.argv) > 3 else "profiles/traffic_r02.json"


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
              f"--warmup 1 --batch {batch} --no-msm --no-extras`, MI355X",
    "batch": batch,
    "correction": "MI355X_MICROARCH.md (HBM): counter values are KB; on gfx950 FETCH_SIZE reports half the bytes of wide "
                  "(16 B per lane) coalesced reads, so per_launch_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024.  Every "
                  "hot kernel here reads with global_load_dwordx4 (16 B per lane): streams in the NTT / quotient / sort "
                  "kernels, 64-byte gathers (4 x dwordx4 per lane) in msm_accumulate - the gather pattern is not one the "
                  "guide calibrates, and Infinity-Cache hits on the 38 MB window table are counted, so that row is an "
                  "upper estimate of HBM bytes.  raw holds the uncorrected counters.",
    "per_launch_bytes": {k: (2 * v.get("fetch_KB_per_launch", 0) + v.get("write_KB_per_launch", 0)) * 1024
                         for k, v in out.items() if keep(k)},
    "raw": {k: v for k, v in out.items() if keep(k)},
}
json.dump(res, open(dest, "w"), indent=1)
print({k: round(v / 1e6, 1) for k, v in res["per_launch_bytes"].items()})
