#!/bin/bash
# Round-2 evidence run (GPU box, through gpurun): default bench line, rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE
# passes for the L1_2D headline and for L2_3D.  Summaries go to profiles/ via tools/summarize_prof.py afterwards.
set -e
export TMPDIR=/tmp
mkdir -p gpurun_out
python bench.py > gpurun_out/bench_r02.json 2> gpurun_out/bench_r02.err || (tail -5 gpurun_out/bench_r02.err; exit 1)
tail -c 600 gpurun_out/bench_r02.json; echo
rm -rf gpurun_out/prof gpurun_out/prof3d
./tools/profile.sh gpurun_out/prof > /dev/null
./tools/profile.sh gpurun_out/prof3d "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --metric L2_3D" > /dev/null
echo profiles collected
