set -x
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu > gpurun_out/pytest_gpu_r01.txt 2>&1; echo "pytest rc=$?" >> gpurun_out/pytest_gpu_r01.txt
tail -15 gpurun_out/pytest_gpu_r01.txt
python __graft_entry__.py smoke > gpurun_out/smoke_r01.txt 2>&1; tail -3 gpurun_out/smoke_r01.txt
python bench.py --steps 4 --warmup 1 > gpurun_out/bench_r01.json 2> gpurun_out/bench_r01.err; tail -c 3000 gpurun_out/bench_r01.json; tail -5 gpurun_out/bench_r01.err
