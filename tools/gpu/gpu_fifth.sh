#!/bin/bash
set -e
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_dbscan_gpu.py tests/test_blocks_gpu.py tests/test_slabs_gpu.py tests/test_host_mirror_gpu.py -x -q 2>&1 | tail -3
for v in 0 1; do
VCP_SORT=$v timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 > gpurun_out/b_2d$v.json 2> gpurun_out/b_2d$v.err
python - <<PY
import json
d=json.load(open("gpurun_out/b_2d$v.json")); print("2D sort=$v", d["ms_per_step"], d["phase_ms"])
PY
done
timeout -k 10 300 python bench.py --no-cpu-baseline --no-extras --steps 10 --metric L2_3D > gpurun_out/b_3d.json 2> gpurun_out/b_3d.err
python - <<PY
import json
d=json.load(open("gpurun_out/b_3d.json")); print("3D", d["ms_per_step"], d["phase_ms"])
PY
