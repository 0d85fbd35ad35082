#!/bin/bash
set -e
mkdir -p gpurun_out
python -m pytest tests/test_host_mirror_gpu.py -x -q -m gpu 2>&1 | tail -3
python - <<'PY' 2>&1 | tee gpurun_out/pcie_inclusive.txt
import time, numpy as np
from vtkcloudpoint_amd import _native as N, synth
ctx = N.Context(0)
d = synth.config_cloud(10_000_000, seed=4)
for it in range(3):
    t0 = time.time(); r = ctx.dbscan(d["motor"], d["eps_l1"], d["min_pts"], N.L1_2D); dt = time.time() - t0
print("host-buffer vcp_dbscan 10M L1_2D (H2D 160 MB + D2H 60 MB, pageable): %.2f ms -> %.1f Mpts/s, clusters %d" % (dt*1e3, 10/dt, r["cf"]))
t0 = time.time(); b = ctx.dbscan_blocks(d["motor"], 0.07, 7, 200); dt = time.time() - t0
print("host-buffer vcp_dbscan_blocks 10M (eps .07, minPts 7, 200/block): %.2f ms, rows %d cols %d kept %d total %d" % (dt*1e3, b["rows"], b["cols"], b["kept"], b["cluster_amount"]))
t0 = time.time(); b = ctx.dbscan_blocks(d["motor"], 0.07, 7, 200); dt = time.time() - t0
print("  second call: %.2f ms" % (dt*1e3))
PY
./tools/profile.sh gpurun_out/prof
./tools/profile.sh gpurun_out/prof3d "--steps 5 --warmup 2 --no-cpu-baseline --no-extras --metric L2_3D"
