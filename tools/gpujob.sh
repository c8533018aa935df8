#!/bin/bash
# ONE runner for the jobs that go to the GPU box through gpurun (round-5 VERDICT item 7: the ~60 one-off gpujob_*.sh of
# rounds 1-5 are gone - their results live in profiles/, their text in the git history, tools/README.md has the table).
#   /usr/local/graft/bin/gpurun --timeout 2400 -- 'bash tools/gpujob.sh JOB [ARGS...]'
# Outputs land under gpurun_out/ (scratch); what is kept is copied to profiles/ by hand or by tools/refresh_profiles.py.
#   suite TAG                 pytest -m gpu, smoke(), default bench.py
#   final TAG                 round-end evidence in one call: suite, 20-step / mixed64 / single-process bench, single-MSM profile,
#                             rocprofv3 stats + PMC traffic, instruction counters, clocks, world-2 (gloo) lines, phase traces
#   prof TAG                  rocprofv3 --kernel-trace --stats, then FETCH_SIZE / WRITE_SIZE passes (separate runs) + summary
#   insts TAG                 SQ_INSTS_VALU / SALU / VMEM / LDS, waves, cycles per kernel (separate PMC passes)
#   clock TAG                 GRBM_GUI_ACTIVE / SQ_BUSY_CU_CYCLES ... per dispatch: clock under each kernel, busy share
#   ab TAG LIB...             same-box A/B of library builds (CAPGPU_LIBRARY), in-tree library first and last
#   env TAG VAR=VAL...        short bench under environment overrides
#   ntt TAG LIB...            device time of batches of coset NTTs for several library builds (tools/gpu_ntt_time.py)
#   phase TAG [VAR=VAL...]    host-side phase traces of the host-witness and coalesced paths (tools/gpu_phase_trace.py)
#   campaign TAG [SEED]       fuzz / stress campaign: prover, primitives, parameter blobs, threads, leaks, graph capture
#   w2                        bench.py --gpus 2 on this one GPU (gloo, shared device): the N > 1 line and its wall time
#   ubench NAME               build and run tools/ubench_NAME.hip (microbenchmarks behind DESIGN.md's numbers)
job=$1; shift
case "$job" in
suite)
  tag=$1
  O=gpurun_out/suite_$tag; mkdir -p $O
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 --timeout=300 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
  S0=$SECONDS; timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench.py wall $((SECONDS-S0)) s"; tail -c 2000 $O/bench.json; echo
  ;;
final)
  tag=$1
  O=gpurun_out/final_$tag
  mkdir -p $O
  timeout 1500 python -m pytest tests -m gpu -q --durations=10 --timeout=300 > $O/pytest_gpu.txt 2>&1; tail -3 $O/pytest_gpu.txt
  python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1; tail -1 $O/smoke.txt
  S0=$SECONDS; timeout 900 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "driver command wall $((SECONDS-S0)) s"; tail -c 300 $O/bench.err
  bash tools/gpujob.sh phase $tag > $O/phase.log 2>&1; cp gpurun_out/phase_$tag/phase.jsonl $O/phase.jsonl
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-reference-schedule --no-msm > $O/bench_20steps.json 2>/dev/null
  timeout 600 python bench.py --workload mixed64 --steps 12 --warmup 3 --no-msm > $O/bench_mixed64.json 2>/dev/null
  CAPGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --single-process --devices 0,0 --batch 128 --steps 4 --warmup 1 --msm-log-n 22 > $O/bench_single_process.json 2>/dev/null
  MINLOG=21 timeout 300 python tools/gpu_msm_deep_ab.py 24 > $O/msm_deep_ab.jsonl 2>/dev/null
  timeout 300 python tools/gpu_two_ctx.py 15 256 2>/dev/null | tail -1 > $O/two_ctx.json
  timeout 600 python tools/gpu_latency_ab.py --quick > $O/latency_ab.jsonl 2>/dev/null
  MSM_LOGS=15,17,20,22,24 timeout 600 python tools/gpu_msm_profile.py > $O/msm_single_profile.json 2>/dev/null
  bash tools/gpujob.sh prof $tag > $O/gpuprof.log 2>&1
  python tools/make_traffic.py gpurun_out/prof_$tag 256 $O/traffic.json > $O/traffic.log 2>&1
  bash tools/gpujob.sh insts $tag > $O/insts.txt 2>&1
  bash tools/gpujob.sh clock $tag > $O/clock.log 2>&1; cp gpurun_out/clock_$tag/clock.json $O/clock.json
  timeout 120 tools/ubench_mix.bin > $O/ubench_mix.txt 2>&1
  bash tools/gpujob.sh w2 > $O/w2.log 2>&1; cp gpurun_out/bench_w2.json $O/bench_w2.json; cp gpurun_out/bench_w2_mixed.json $O/bench_w2_mixed.json
  python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("value", round(d["value"], 1), "ms/step", round(d["ms_per_step"], 2), {k: round(v / d["steps"], 2) for k, v in d["top_kernels_ms"].items()})
for k in ("alu_roofline", "reference_schedule", "latency_ms_batch1", "pcie_inclusive", "n2p16", "cpu_baseline", "cpu_baseline_64_threads", "two_contexts_per_device", "mixed64", "coalesced_single_calls"):
    print(k, d.get(k))
print([ (l.get("log_n"), round(l.get("ms", 0), 3), l.get("identity_check")) for l in d.get("msm", [])])
print("20 steps:", round(json.load(open("$O/bench_20steps.json"))["value"], 1), "mixed64:", round(json.load(open("$O/bench_mixed64.json"))["value"], 1))
PY

  ;;
prof)
  tag=$1
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT
  mkdir -p $R/gpurun_out/prof_$tag
  cd $R
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag/trace -- python3 bench.py --one-context --steps 3 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > gpurun_out/prof_$tag/bench_under_trace.json 2> gpurun_out/prof_$tag/trace.err
  ls -R gpurun_out/prof_$tag/trace | head -20
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d gpurun_out/prof_$tag/pmc_fetch -- python3 bench.py --one-context --steps 2 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/prof_$tag/pmc_fetch.err
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d gpurun_out/prof_$tag/pmc_write -- python3 bench.py --one-context --steps 2 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/prof_$tag/pmc_write.err
  python3 tools/summarize_prof.py gpurun_out/prof_$tag > gpurun_out/prof_$tag/summary.txt 2>&1
  cat gpurun_out/prof_$tag/summary.txt | head -60
  # keep the merge small: drop the raw per-dispatch CSVs except stats
  find gpurun_out/prof_$tag -name "*kernel_trace.csv" -size +20M -delete
  find gpurun_out/prof_$tag -name "*counter_collection.csv" -size +20M -delete
  du -sh gpurun_out/prof_$tag

  ;;
insts)
  tag=$1
  cd /tmp && export TMPDIR=/tmp
  R=$GRAFT_REPO_ROOT
  cd $R
  mkdir -p gpurun_out/insts_$tag
  for ctr in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_WAVES SQ_INSTS_VMEM" "SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES" "SQ_BUSY_CYCLES SQ_INSTS_LDS"; do
    name=$(echo $ctr | tr ' ' '_')
    rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/insts_$tag/$name -- python3 bench.py --one-context --steps 1 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras > /dev/null 2> gpurun_out/insts_$tag/$name.err
  done
  python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for f in glob.glob("gpurun_out/insts_$tag/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").replace("cap::pk::", "").replace("cap::", "").split("(")[0]
        a = agg[k][r["Counter_Name"]]
        a[0] += float(r["Counter_Value"]); a[1] += 1
for k in sorted(agg, key=lambda k: -agg[k].get("SQ_INSTS_VALU", [0])[0])[:10]:
    print(k[:40], {c: (round(v[0] / v[1]), v[1]) for c, v in agg[k].items()})
PY
  find gpurun_out/insts_$tag -name "*counter_collection.csv" -size +5M -delete

  ;;
clock)
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

  ;;
ab)
  TAG=$1; shift
  OUT=gpurun_out/abn_$TAG
  mkdir -p $OUT
  ARGS="--steps 4 --warmup 2 --no-cpu-baseline --no-reference-schedule --no-msm --no-extras"
  python bench.py $ARGS > $OUT/base.json 2> $OUT/base.err
  i=0
  for L in "$@"; do
    i=$((i+1))
    CAPGPU_LIBRARY=$PWD/$L python bench.py $ARGS > $OUT/v$i.json 2> $OUT/v$i.err
  done
  python bench.py $ARGS > $OUT/base2.json 2> $OUT/base2.err
  python - "$OUT" <<'PY'
import json, sys, glob, os
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.load(open(f))
        print(os.path.basename(f), round(d["value"], 1), "proofs/s", {k: round(v / d["steps"], 2) for k, v in d["top_kernels_ms"].items()})
    except Exception as e:
        print(f, "failed", e)
PY

  ;;
env)
  tag=$1; shift
  mkdir -p gpurun_out
  for kv in "$@"; do export "$kv"; done
  python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-reference-schedule --no-msm > gpurun_out/bench_$tag.json 2> gpurun_out/bench_$tag.err
  python - <<PY
import json
d=json.load(open("gpurun_out/bench_$tag.json"))
print("$tag", "value", round(d["value"],1), "ms/step", round(d["ms_per_step"],2)); print({k:round(v/3,1) for k,v in d["top_kernels_ms"].items()})
PY
  for kv in "$@"; do unset "${kv%%=*}"; done

  ;;
ntt)
  tag=$1; shift
  O=gpurun_out/ntt_$tag; mkdir -p $O; rm -f $O/ntt_ab.jsonl
  for lib in cap_amd/libcapgpu.so "$@" cap_amd/libcapgpu.so; do
    CAPGPU_LIBRARY=$PWD/$lib timeout 600 python tools/gpu_ntt_time.py $(basename $lib .so) >> $O/ntt_ab.jsonl 2>> $O/ntt_ab.err
  done
  cat $O/ntt_ab.jsonl
  ;;
phase)
  tag=$1; shift
  O=gpurun_out/phase_$tag; mkdir -p $O; rm -f $O/phase.jsonl
  for mode in resident host host2 coalesce; do
    env CAPGPU_X=1 "$@" timeout 600 python tools/gpu_phase_trace.py $mode $tag --reps 6 --calls 16 >> $O/phase.jsonl 2>> $O/phase.err
  done
  cat $O/phase.jsonl; tail -3 $O/phase.err
  ;;
campaign)
  # fuzz / stress campaign on the round's binary (seeds differ from the bounded slices under -m gpu): TAG [SEED_BASE]
  tag=$1; seed=${2:-61000}
  O=gpurun_out/campaign_$tag.txt; : > $O
  step() { name=$1; shift; echo "== $name" >> $O; S0=$SECONDS; "$@" 2>&1 | tail -2 >> $O; echo "wall $((SECONDS-S0)) s" >> $O; }
  step fuzz_prover timeout 900 python tools/gpu_fuzz_prover.py 120 $((seed+1))
  step fuzz_prover_big timeout 900 python tools/gpu_fuzz_prover.py 8 $((seed+2)) big
  step fuzz_prims timeout 900 python tools/gpu_fuzz_prims.py 3000 $((seed+3))
  step fuzz_params timeout 300 python tools/gpu_fuzz_params.py 300 $((seed+4))
  step thread_stress timeout 300 python tools/gpu_thread_stress.py 45
  step leak timeout 300 python tools/gpu_leak_check.py 200
  step capture_stress timeout 600 python tools/gpu_capture_stress.py 400
  step fuzz_prover_coeff_commit_no_graphs env CAPGPU_WIRE_COMMIT=coeffs CAPGPU_GRAPH_MAX_BATCH=0 timeout 900 python tools/gpu_fuzz_prover.py 60 $((seed+5))
  step fuzz_prover_prestage_inflight3 env CAPGPU_COALESCE_PRESTAGE=1 CAPGPU_COALESCE_INFLIGHT=3 timeout 900 python tools/gpu_fuzz_prover.py 60 $((seed+6))
  step thread_stress_prestage env CAPGPU_COALESCE_PRESTAGE=1 timeout 300 python tools/gpu_thread_stress.py 30
  cat $O
  ;;
w2)
  mkdir -p gpurun_out
  S0=$SECONDS
  CAPGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 900 python bench.py --gpus 2 > gpurun_out/bench_w2.json 2> gpurun_out/bench_w2.err
  echo "rc=$? bench.py --gpus 2 (one GPU shared by both ranks, gloo) wall $((SECONDS-S0)) s"; tail -c 1500 gpurun_out/bench_w2.json; echo
  CAPGPU_ALLOW_DUPLICATE_DEVICES=1 timeout 600 python bench.py --gpus 2 --workload mixed64 --steps 2 --warmup 1 --no-msm > gpurun_out/bench_w2_mixed.json 2> gpurun_out/bench_w2_mixed.err
  echo "rc=$?"; tail -c 600 gpurun_out/bench_w2_mixed.json; echo
  ;;
ubench)
  name=$1
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I cap_amd/csrc tools/ubench_$name.hip -o tools/ubench_$name.bin 2>/dev/null
  mkdir -p gpurun_out; timeout 600 tools/ubench_$name.bin | tee gpurun_out/ubench_$name.txt
  ;;
*)
  echo "usage: tools/gpujob.sh suite|final|prof|insts|clock|ab|env|ntt|phase|campaign|w2|ubench ..." >&2; exit 2
  ;;
esac
