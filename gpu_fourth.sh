#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
python bench.py > gpurun_out/bench_r01_d.json 2> gpurun_out/bench_r01_d.err || (tail -5 gpurun_out/bench_r01_d.err; exit 1)
cat gpurun_out/bench_r01_d.json
rm -rf gpurun_out/prof gpurun_out/prof3d
./tools/profile.sh gpurun_out/prof > /dev/null
./tools/profile.sh gpurun_out/prof3d "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --metric L2_3D" > /dev/null
