#!/bin/bash
# A/B of environment switches on the block pipeline's stage times: tools/gpu/ab_blocks.sh "VAR=1" "VAR2=3 VAR3=4" ...
set -e
mkdir -p gpurun_out
for v in "" "$@"; do
  echo "== $v"
  env $v timeout -k 10 300 python tools/bench_blocks.py 10000000 6 2>&1 | grep -E "iter [345]|^cluster"
done
