#!/bin/bash
# Failure handling of the library-communicator bootstrap, on a 1-GPU box: two ranks on device 0 cannot form an RCCL
# communicator (duplicate GPU), so the bootstrap must fail or time out on every rank, be reported in the JSON line
# (library_comm_error) and leave the replica headline and the torch-path MSM leg intact.
mkdir -p gpurun_out
export CAPGPU_BENCH_DEVICE=0 CAPGPU_BENCH_FORCE_LIB_COMM=1 CAPGPU_BENCH_COMM_TIMEOUT=40
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29527 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --dist-backend gloo --msm-log-n 18 > gpurun_out/bench_w2_comm.json 2> gpurun_out/bench_w2_comm.err
echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_w2_comm.json") if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value", round(d["value"],1), "library_comm_error:", d.get("library_comm_error")); print(d.get("msm"))
PY
tail -3 gpurun_out/bench_w2_comm.err
