#!/bin/bash
# Where do the 3.4 ms of a single proof go?  Kernel trace of batch-1 steps: busy time (union of kernel intervals), gaps,
# per-kernel sums, per step.  BATCH=32 bash tools/gpujob_lat1.sh: the same for a mid-size batch.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/lat1; mkdir -p gpurun_out/lat1
timeout 300 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/lat1 -- python3 bench.py --batch ${BATCH:-1} --steps 40 --warmup 5 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/lat1/bench.json 2> gpurun_out/lat1/err.txt
python3 - <<'PY'
import csv, glob, collections, json
f = glob.glob("gpurun_out/lat1/**/*kernel_trace.csv", recursive=True)[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 40 steps: find by total count / 45 steps
per_step = len(rows) // 45 if len(rows) > 4500 else None
tail = rows[-(per_step * 30):] if per_step else rows[len(rows) // 2:]
t0, t1 = tail[0][0], tail[-1][1]
busy = 0; cur_s, cur_e = tail[0][0], tail[0][1]
gaps = []
for s, e, _ in tail[1:]:
    if s > cur_e:
        busy += cur_e - cur_s; gaps.append(s - cur_e); cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
steps = 30 if per_step else None
print("launches per step", per_step, "window ms", (t1 - t0) / 1e6, "busy ms", busy / 1e6, "per step: wall", (t1 - t0) / 1e6 / steps, "busy", busy / 1e6 / steps)
gaps.sort()
big = [g for g in gaps if g > 20000]
print("gaps per step", len(gaps) / steps, "sum gaps ms/step", sum(gaps) / 1e6 / steps, "gaps > 20 us per step", len(big) / steps, "their sum ms/step", sum(big) / 1e6 / steps, "median gap us", gaps[len(gaps) // 2] / 1e3)
agg = collections.defaultdict(lambda: [0, 0])
for s, e, k in tail:
    k = k.replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
    agg[k][0] += e - s; agg[k][1] += 1
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][0])[:28]:
    print(f"{k[:44]:44s} {v[0] / 1e3 / steps:8.1f} us/step  {v[1] / steps:5.1f} launches/step  avg {v[0] / v[1] / 1e3:6.1f} us")
print(open("gpurun_out/lat1/bench.json").read()[:300])
PY
find gpurun_out/lat1 -name "*.csv" -size +3M -delete
