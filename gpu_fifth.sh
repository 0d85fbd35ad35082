#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_slabs_gpu.py -x -q 2>&1 | tail -25
