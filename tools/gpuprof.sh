#!/bin/bash
# rocprofv3 evidence for bench.py (run on the GPU box): kernel-trace stats, then PMC passes (separately, no trace domains).
# --one-context: the same batch on ONE stream with the HIP-event profiler on - the pass bench.py takes its per-kernel
# durations from (with the default two contexts, kernels of the two streams overlap and a trace times them inflated)
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
