#!/bin/bash
# Fuzz / stress campaign on the final binary (seeds differ from the bounded slices under -m gpu); two contexts per device,
# as bench.py runs the library, plus one pass of the multi-device tests' configuration on the prover fuzz.
O=gpurun_out/campaign_$1; mkdir -p $O
export CAPGPU_CONTEXTS_PER_DEVICE=2
( time timeout 600 python tools/gpu_fuzz_prover.py 150 9103 ) > $O/fuzz_prover.txt 2>&1; tail -2 $O/fuzz_prover.txt
( time timeout 300 python tools/gpu_fuzz_prover.py 12 9104 big ) > $O/fuzz_prover_big.txt 2>&1; tail -2 $O/fuzz_prover_big.txt
( time timeout 600 python tools/gpu_fuzz_prims.py 4000 9105 ) > $O/fuzz_prims.txt 2>&1; tail -2 $O/fuzz_prims.txt
( time timeout 300 python tools/gpu_fuzz_params.py 800 9106 ) > $O/fuzz_params.txt 2>&1; tail -2 $O/fuzz_params.txt
( time timeout 200 python tools/gpu_thread_stress.py 60 ) > $O/thread_stress.txt 2>&1; tail -2 $O/thread_stress.txt
( time timeout 300 python tools/gpu_leak_check.py 300 ) > $O/leak.txt 2>&1; tail -2 $O/leak.txt
timeout 200 python -m pytest tests/test_bookkeeping.py -m gpu -q > $O/bookkeeping.txt 2>&1; tail -1 $O/bookkeeping.txt
