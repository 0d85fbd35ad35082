"""C3: ICP of 1 M data points against a 100-point model, 50 rounds (device-resident), for kernel statistics."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
c = synth.config_icp()
dm = torch.from_numpy(c["model"]).cuda()
dd = torch.from_numpy(c["data"]).cuda()
torch.cuda.synchronize()
best = 1e9
for _ in range(5):
    t = time.perf_counter()
    r = ctx.icp_dev(dm.data_ptr(), len(c["model"]), dd.data_ptr(), len(c["data"]), 0.0, 50, N.STOP_SSE_DELTA)
    best = min(best, time.perf_counter() - t)
print("1M x 100, %d rounds: %.3f ms = %.0f rounds/s" % (r["iters"], best * 1e3, r["iters"] / best), flush=True)
