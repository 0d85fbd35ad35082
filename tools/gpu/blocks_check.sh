#!/bin/bash
# block pipeline: parity tests, then stage times and the kernel table
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_blocks_gpu.py tests/test_full_size_gpu.py -x -q -k "blocks or block_pipeline" > gpurun_out/blocks_pytest.log 2>&1 || (tail -40 gpurun_out/blocks_pytest.log; exit 1)
tail -3 gpurun_out/blocks_pytest.log
./tools/gpu/kernel_stats_blocks.sh > gpurun_out/blocks_kstats.txt 2>&1 || (tail -20 gpurun_out/blocks_kstats.txt; exit 1)
grep -v "^[EW]2026" gpurun_out/blocks_kstats.txt
