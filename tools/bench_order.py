"""Input-order sensitivity: the C4 cloud in the generator's shuffled order (bench.py's workload) vs a raster
order (rows of 1 unit in y, x ascending inside a row), the order a scanning instrument delivers."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
ctx.timing_enable(True)
c4 = synth.config_cloud(10_000_000, seed=4)
m = c4["motor"]
raster = m[np.lexsort((m[:, 0], np.floor(m[:, 1])))]
for tag, arr in (("shuffled", m), ("raster", np.ascontiguousarray(raster))):
    n = len(arr)
    d = torch.from_numpy(arr).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(8):
        t = time.perf_counter()
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, c4["eps_l1"], c4["min_pts"], N.L1_2D, 0, None, lab.data_ptr(),
                                core.data_ptr(), cls.data_ptr())
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("%s: %.2f ms = %.0f Mpoints/s, %d clusters; %s" % (tag, best * 1e3, n / best / 1e6, cf,
                                                          [(k, round(v, 3)) for k, v in ctx.timing()]), flush=True)
