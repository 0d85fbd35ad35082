#!/bin/bash
# parity subset + 2-D / 3-D bench phases
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dbscan_gpu.py tests/test_blocks_gpu.py tests/test_slabs_gpu.py tests/test_host_mirror_gpu.py -x -q 2>&1 | tail -3
for m in L1_2D L2_3D; do
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --metric $m > gpurun_out/q_$m.json 2> gpurun_out/q_$m.err
python - <<PY
import json
d=json.load(open("gpurun_out/q_$m.json")); print("$m", round(d["ms_per_step"],4), d["phase_ms"])
PY
done
