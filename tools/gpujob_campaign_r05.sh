#!/bin/bash
# Round-5 fuzz / stress campaign on the final binary (seeds differ from the bounded slices under -m gpu): default
# configuration (two contexts on the one device, hipGraph replay of small batches, grid reductions, host-side inversion),
# then the same prover fuzz with one context and graphs off.  The prover fuzz now draws the input form (tables /
# coefficients) for keys and wires and proves every batch three times (capture + replay of small batches).
O=gpurun_out/campaign_r05; mkdir -p $O
( time timeout 900 python tools/gpu_fuzz_prover.py 160 51001 ) > $O/fuzz_prover.txt 2>&1; tail -2 $O/fuzz_prover.txt
( time timeout 400 python tools/gpu_fuzz_prover.py 12 51002 big ) > $O/fuzz_prover_big.txt 2>&1; tail -2 $O/fuzz_prover_big.txt
( time timeout 600 python tools/gpu_fuzz_prims.py 4000 51003 ) > $O/fuzz_prims.txt 2>&1; tail -2 $O/fuzz_prims.txt
( time timeout 300 python tools/gpu_fuzz_params.py 600 51004 ) > $O/fuzz_params.txt 2>&1; tail -2 $O/fuzz_params.txt
( time timeout 200 python tools/gpu_thread_stress.py 60 ) > $O/thread_stress.txt 2>&1; tail -2 $O/thread_stress.txt
( time timeout 300 python tools/gpu_leak_check.py 300 ) > $O/leak.txt 2>&1; tail -2 $O/leak.txt
( time CAPGPU_CONTEXTS_PER_DEVICE=1 CAPGPU_GRAPH_MAX_BATCH=0 timeout 600 python tools/gpu_fuzz_prover.py 80 51005 ) > $O/fuzz_prover_one_context_no_graphs.txt 2>&1; tail -2 $O/fuzz_prover_one_context_no_graphs.txt
# round 5's switches the other way round: wire commitments from coefficients, the separate sort launches, no side stream
( time CAPGPU_WIRE_COMMIT=coeffs CAPGPU_MSM_CHAINED=0 CAPGPU_R1_OVERLAP_MAX=0 timeout 600 python tools/gpu_fuzz_prover.py 80 51006 ) > $O/fuzz_prover_round4_schedule.txt 2>&1; tail -2 $O/fuzz_prover_round4_schedule.txt
( time CAPGPU_R1_OVERLAP_MAX=64 timeout 600 python tools/gpu_fuzz_prover.py 80 51007 ) > $O/fuzz_prover_overlap_all.txt 2>&1; tail -2 $O/fuzz_prover_overlap_all.txt
( time CAPGPU_MSM_CHAINED=1 timeout 400 python tools/gpu_fuzz_prims.py 2500 51008 ) > $O/fuzz_prims2.txt 2>&1; tail -2 $O/fuzz_prims2.txt
