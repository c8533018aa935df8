#!/bin/bash
# Round-4 fuzz / stress campaign on the final binary (seeds differ from the bounded slices under -m gpu): default
# configuration (two contexts on the one device, hipGraph replay of small batches, grid reductions, host-side inversion),
# then the same prover fuzz with one context and graphs off.  The prover fuzz now draws the input form (tables /
# coefficients) for keys and wires and proves every batch three times (capture + replay of small batches).
O=gpurun_out/campaign_r04; mkdir -p $O
( time timeout 900 python tools/gpu_fuzz_prover.py 160 41001 ) > $O/fuzz_prover.txt 2>&1; tail -2 $O/fuzz_prover.txt
( time timeout 400 python tools/gpu_fuzz_prover.py 12 41002 big ) > $O/fuzz_prover_big.txt 2>&1; tail -2 $O/fuzz_prover_big.txt
( time timeout 600 python tools/gpu_fuzz_prims.py 4000 41003 ) > $O/fuzz_prims.txt 2>&1; tail -2 $O/fuzz_prims.txt
( time timeout 300 python tools/gpu_fuzz_params.py 600 41004 ) > $O/fuzz_params.txt 2>&1; tail -2 $O/fuzz_params.txt
( time timeout 200 python tools/gpu_thread_stress.py 60 ) > $O/thread_stress.txt 2>&1; tail -2 $O/thread_stress.txt
( time timeout 300 python tools/gpu_leak_check.py 300 ) > $O/leak.txt 2>&1; tail -2 $O/leak.txt
( time CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_GRAPH_MAX_BATCH=0 timeout 600 python tools/gpu_fuzz_prover.py 80 41005 ) > $O/fuzz_prover_one_context_no_graphs.txt 2>&1; tail -2 $O/fuzz_prover_one_context_no_graphs.txt
