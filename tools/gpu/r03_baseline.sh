#!/bin/bash
# Round-3 state check: GPU parity suite, default bench line, kernel stats of the headline step and of the block pipeline,
# C4 / C5 sizes.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/r03_pytest.log 2>&1 || (tail -30 gpurun_out/r03_pytest.log; exit 1)
tail -3 gpurun_out/r03_pytest.log
timeout -k 10 400 python bench.py > gpurun_out/r03_bench.json 2> gpurun_out/r03_bench.err || (tail -5 gpurun_out/r03_bench.err; exit 1)
tail -c 1500 gpurun_out/r03_bench.json; echo
./tools/gpu/kernel_stats.sh > gpurun_out/r03_kstats.txt 2>&1 || (tail -5 gpurun_out/r03_kstats.txt; exit 1)
./tools/gpu/kernel_stats_blocks.sh > gpurun_out/r03_kstats_blocks.txt 2>&1 || (tail -5 gpurun_out/r03_kstats_blocks.txt; exit 1)
timeout -k 10 300 python tools/bench_sizes.py > gpurun_out/r03_sizes.txt 2>&1 || (tail -5 gpurun_out/r03_sizes.txt; exit 1)
cat gpurun_out/r03_sizes.txt
