#!/bin/bash
set -e
mkdir -p gpurun_out
python __graft_entry__.py smoke 2>&1 | tail -3
python bench.py > gpurun_out/bench_r01_a.json 2> gpurun_out/bench_r01_a.err || (tail -20 gpurun_out/bench_r01_a.err; exit 1)
cat gpurun_out/bench_r01_a.json
./tools/profile.sh gpurun_out/prof
