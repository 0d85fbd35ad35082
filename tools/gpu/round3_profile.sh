#!/bin/bash
# Round-3 evidence run (GPU box, through gpurun): default bench line, rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE
# passes for the L1_2D headline and for L2_3D, kernel table of the block pipeline, C4 / C5 sizes, the block pipeline
# through bench.py (single device, the sharded per-rank program at one rank, the same under RCCL), SQ counters.
set -e
export TMPDIR=/tmp
TAG=${1:-r03}
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_$TAG.json 2> gpurun_out/bench_$TAG.err || (tail -5 gpurun_out/bench_$TAG.err; exit 1)
tail -c 400 gpurun_out/bench_$TAG.json; echo
rm -rf gpurun_out/prof gpurun_out/prof3d
./tools/profile.sh gpurun_out/prof > /dev/null
./tools/profile.sh gpurun_out/prof3d "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --metric L2_3D" > /dev/null
./tools/gpu/kernel_stats_blocks.sh > gpurun_out/${TAG}_blocks_kernels.txt 2>&1
timeout -k 10 300 python tools/bench_sizes.py > gpurun_out/${TAG}_sizes.txt 2>&1
./tools/gpu/blocks_bench.sh > gpurun_out/${TAG}_blocks_bench.txt 2>&1
./tools/gpu/sq_counters.sh > /dev/null 2>&1
echo profiles collected
