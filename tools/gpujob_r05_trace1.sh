#!/bin/bash
# kernel timeline of ONE proof (the last of 30): start offset, duration, queue, name - tools/gpujob_r05_trace1.sh TAG [env...]
TAG=${1:-t1}; shift
OUT=gpurun_out/trace1_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
for kv in "$@"; do export "$kv"; done
python tools/gpu_prove1.py 30 2> $OUT/plain.err
rocprofv3 --kernel-trace --output-format csv -d $OUT/trace -o run -- python3 tools/gpu_prove1.py 30 > $OUT/prof.out 2> $OUT/prof.err
python - <<PY
import csv, glob
f = glob.glob("$OUT/trace/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
def short(n):
    n = n.replace("cap::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0].strip()[:44]
marks = [i for i, r in enumerate(rows) if "k_quotient" in r["Kernel_Name"]]
a, b = marks[-2], marks[-1]
# one proof = from the first kernel after the previous proof's last kernel ... find previous proof end: the kernel before the gap > 100 us preceding b's proof
seg = rows[a:b + 60]
# locate start of last proof: largest gap between consecutive kernels in rows[a:b]
gaps = [(int(rows[i + 1]["Start_Timestamp"]) - int(rows[i]["End_Timestamp"]), i) for i in range(a, b)]
g, gi = max(gaps)
start = gi + 1
end = start
t0 = int(rows[start]["Start_Timestamp"])
lines = []
busy = 0
last_end = t0
i = start
while i < len(rows):
    r = rows[i]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if i > start and s - last_end > 100000 and i > b: break
    lines.append("%9.1f us  +%7.1f us  gap %6.1f  q%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last_end) / 1e3, r.get("Queue_Id", "?"), short(r["Kernel_Name"])))
    busy += e - s
    last_end = max(last_end, e)
    i += 1
open("$OUT/timeline.txt", "w").write("\n".join(lines) + "\n")
print("kernels", len(lines), "span ms", (last_end - t0) / 1e6, "sum of kernel durations ms", busy / 1e6)
print(open("$OUT/plain.err").read()[-200:])
PY
head -150 $OUT/timeline.txt
