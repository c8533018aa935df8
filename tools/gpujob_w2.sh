#!/bin/bash
# world-size-2 exercise of bench.py on a 1-GPU box: both ranks on device 0, gloo for the collectives
mkdir -p gpurun_out
export CAPGPU_BENCH_DEVICE=0
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 2 --warmup 1 --batch 8 --dist-backend gloo > gpurun_out/bench_w2.json 2> gpurun_out/bench_w2.err
echo "rc=$?"
python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/bench_w2.json") if l.startswith("{")][-1])
print("n_gpus", d["n_gpus"], "value", round(d["value"],1), "scaling", d["scaling"]); print(d.get("msm"))
PY
tail -5 gpurun_out/bench_w2.err
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --steps 2 --warmup 1 --workload mixed64 --no-msm --dist-backend gloo > gpurun_out/bench_w2_mixed.json 2> gpurun_out/bench_w2_mixed.err
echo "rc=$?"; tail -c 600 gpurun_out/bench_w2_mixed.json; tail -3 gpurun_out/bench_w2_mixed.err
