#!/bin/bash
# second campaign: other seeds, longer big-shape prover fuzz, one context per device this time
O=gpurun_out/campaign2_$1; mkdir -p $O
( time timeout 900 python tools/gpu_fuzz_prover.py 30 777 big ) > $O/fuzz_prover_big.txt 2>&1; tail -2 $O/fuzz_prover_big.txt
( time timeout 600 python tools/gpu_fuzz_prover.py 200 779 ) > $O/fuzz_prover.txt 2>&1; tail -2 $O/fuzz_prover.txt
( time timeout 600 python tools/gpu_fuzz_prims.py 6000 778 ) > $O/fuzz_prims.txt 2>&1; tail -2 $O/fuzz_prims.txt
( time timeout 300 python tools/gpu_thread_stress.py 120 ) > $O/thread_stress.txt 2>&1; tail -2 $O/thread_stress.txt
