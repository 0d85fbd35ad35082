#!/bin/bash
set -e
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r01_e.json 2> gpurun_out/bench_r01_e.err || (tail -5 gpurun_out/bench_r01_e.err; exit 1)
cat gpurun_out/bench_r01_e.json
rm -rf gpurun_out/prof gpurun_out/prof3d
./tools/profile.sh gpurun_out/prof > /dev/null
./tools/profile.sh gpurun_out/prof3d "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --metric L2_3D" > /dev/null
