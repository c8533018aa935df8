#!/bin/bash
# Where a single proof's 4 ms go: kernel trace of bench.py --batch ${BATCH:-1} (tools/gpujob_batch1.sh TAG)
TAG=${1:-b1}
OUT=gpurun_out/batchN_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python bench.py --batch ${BATCH:-1} --steps 20 --warmup 5 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > $OUT/bench.json 2> $OUT/bench.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 bench.py --batch ${BATCH:-1} --steps 20 --warmup 5 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > $OUT/prof.json 2> $OUT/prof.err
python - <<PY
import csv, glob, json, collections
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# last 20 steps: find the last 20 occurrences of k_blind<1> (first prover kernel after the wire iNTT) as step markers
names = [r["Kernel_Name"] for r in rows]
def short(n):
    n = n.replace("cap::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0].strip()
idx = [i for i, n in enumerate(names) if "k_quotient" in n]
idx = idx[-20:]
# a step = from the previous step's last kernel to this one's; use windows between consecutive k_quotient launches
per = collections.defaultdict(lambda: [0, 0.0])
tot_busy = 0.0; tot_wall = 0.0; nk = 0
for a, b in zip(idx[:-1], idx[1:]):
    seg = rows[a:b]
    t0 = int(seg[0]["Start_Timestamp"]); t1 = int(rows[b]["Start_Timestamp"])
    tot_wall += (t1 - t0) / 1e6
    for r in seg:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        tot_busy += d; nk += 1
        p = per[short(r["Kernel_Name"])]; p[0] += 1; p[1] += d
n = len(idx) - 1
out = {"steps": n, "wall_ms_per_proof": tot_wall / n, "gpu_busy_ms_per_proof": tot_busy / n, "kernels_per_proof": nk / n,
       "kernels": {k: {"launches": v[0] / n, "ms": v[1] / n} for k, v in sorted(per.items(), key=lambda kv: -kv[1][1])}}
json.dump(out, open("$OUT/summary.json", "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "kernels"}))
for k, v in list(out["kernels"].items())[:25]:
    print("%-40s %5.1f launches %7.3f ms" % (k[:40], v["launches"], v["ms"]))
PY
