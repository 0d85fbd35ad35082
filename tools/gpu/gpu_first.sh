#!/bin/bash
# first GPU contact: parity tests + a rough timing
set -e
mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -15
python - <<'PY' 2>&1 | tee gpurun_out/first_timing.txt
import time, numpy as np, torch
from vtkcloudpoint_amd import _native as N, synth
ctx = N.Context(0)
ctx.timing_enable(True)
for n in (1_000_000, 10_000_000):
    d = synth.config_cloud(n)
    for name, arr, eps, metric in (("L1_2D", d["motor"], d["eps_l1"], N.L1_2D), ("L2_3D", d["xyz"], d["eps_l2"], N.L2_3D)):
        t = torch.from_numpy(arr).cuda()
        lab = torch.zeros(n, dtype=torch.int32, device="cuda")
        core = torch.zeros(n, dtype=torch.uint8, device="cuda")
        cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
        torch.cuda.synchronize()
        for it in range(3):
            t0 = time.time()
            cf, ev = ctx.dbscan_dev(t.data_ptr(), n, arr.shape[1], eps, d["min_pts"], metric, 0, None, lab.data_ptr(), core.data_ptr(), cls.data_ptr())
            dt = time.time() - t0
        print(n, name, "clusters", cf, "ms %.3f" % (dt * 1e3), "Mpts/s %.1f" % (n / dt / 1e6))
        print("   ", [(k, round(v, 3)) for k, v in ctx.timing()])
PY
python - <<'PY' 2>&1 | tee gpurun_out/first_icp.txt
import time, numpy as np, torch
from vtkcloudpoint_amd import _native as N, synth
ctx = N.Context(0)
for jit in (0.05, 0.0):
    d = synth.config_icp(nd=1_000_000, nm=100, jitter=jit)
    m = torch.from_numpy(d["model"]).cuda(); x = torch.from_numpy(d["data"]).cuda(); torch.cuda.synchronize()
    for it in range(3):
        t0 = time.time(); r = ctx.icp_dev(m.data_ptr(), 100, x.data_ptr(), 1_000_000, 0.0 if jit else 1e-4, 50, N.STOP_SSE_DELTA if jit else N.STOP_RMSE); dt = time.time() - t0
    print("jitter", jit, "iters", r["iters"], "rmse %.3e" % r["rmse"], "ms %.3f" % (dt*1e3), "iters/s %.0f" % (r["iters"]/dt))
PY
