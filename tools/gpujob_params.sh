#!/bin/bash
# on-disk parameter formats: timing at CRS size, then the whole GPU suite and the default bench
tag=$1
mkdir -p gpurun_out
python tools/gpu_time_params.py > gpurun_out/params_timing_$tag.json 2> gpurun_out/params_timing_$tag.err; cat gpurun_out/params_timing_$tag.json; tail -3 gpurun_out/params_timing_$tag.err
bash tools/gpujob_main.sh $tag
