"""Device-resident timing of the block-partitioned pipeline (getClusterFromMotor + StartCode + CompleteWork3,
FrmMain.cs:1214-1544) at the reference's defaults: per-stage wall time and the library's phase timers."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cloud = synth.config_cloud(n, seed=4)
ctx = N.Context(0)
ctx.timing_enable(True)
d = torch.from_numpy(cloud["motor"]).cuda()
local = torch.zeros(n, dtype=torch.int32, device="cuda")
labels = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 4):
    t0 = time.perf_counter()
    info = ctx.blocks_begin(None, 0.07, 7, 200, 3, device_ptr=d.data_ptr(), n=n)
    t1 = time.perf_counter()
    tim_b = ctx.timing()
    ev = ctx.blocks_cluster_dev(0, info["nblocks"], local.data_ptr())
    t2 = time.perf_counter()
    tim_c = ctx.timing()
    out = ctx.blocks_finish_dev(local.data_ptr(), ev, labels.data_ptr())
    t3 = time.perf_counter()
    tim_f = ctx.timing()
    print("iter %d: begin %.2f ms, cluster %.2f ms, finish %.2f ms, total %.2f ms; blocks %d kept %d clusters %d"
          % (it, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, (t3 - t0) * 1e3, info["nblocks"], out["kept"],
             out["cluster_amount"]))
print("begin  ", [(k, round(v, 3)) for k, v in tim_b])
print("cluster", [(k, round(v, 3)) for k, v in tim_c])
print("finish ", [(k, round(v, 3)) for k, v in tim_f])
