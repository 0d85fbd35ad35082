#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 600 python tools/profile_exact.py > gpurun_out/profile_exact.txt 2>&1 || (tail -20 gpurun_out/profile_exact.txt; exit 1)
head -60 gpurun_out/profile_exact.txt
