#!/bin/bash
# round 5: what could a register-resident radix-16 / radix-256 round save at most?  The shipped passes against a TIMING-ONLY build whose
# radix-4 rounds are chained in registers (same multiplications, twiddle loads and additions; no LDS traffic, no barriers between rounds;
# wrong results): -DCAP_NTT_EXPERIMENT_NO_LDS_ROUNDS, tools/libcapgpu_nttexp.so
OUT=gpurun_out/r05_nttexp
mkdir -p $OUT
for i in 1 2; do
python tools/gpu_ntt_time.py shipped >> $OUT/ntt.jsonl 2>> $OUT/err.txt
CAPGPU_LIBRARY=$PWD/tools/libcapgpu_nttexp.so python tools/gpu_ntt_time.py no_lds_rounds >> $OUT/ntt.jsonl 2>> $OUT/err.txt
done
cat $OUT/ntt.jsonl; tail -3 $OUT/err.txt
