"""Side measurements for DESIGN.md section 5: the 50 M-point C5 cloud (resident) and the 10 M-point C4 cloud through
the host-buffer entry point (pageable H2D + D2H over PCIe included)."""
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


def resident(cloud, eps, min_pts, tag):
    n = len(cloud)
    d = torch.from_numpy(cloud).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(6):
        t = time.perf_counter()
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, eps, min_pts, N.L1_2D, 0, None, lab.data_ptr(), core.data_ptr(),
                                cls.data_ptr())
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("%s: n=%d resident %.2f ms = %.0f Mpoints/s, %d clusters; phases %s"
          % (tag, n, best * 1e3, n / best / 1e6, cf, [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)


c4 = synth.config_cloud(10_000_000, seed=4)
resident(c4["motor"], c4["eps_l1"], c4["min_pts"], "C4")
best = None
for _ in range(4):
    t = time.perf_counter()
    r = ctx.dbscan(c4["motor"], c4["eps_l1"], c4["min_pts"], N.L1_2D)
    e = time.perf_counter() - t
    best = e if best is None else min(best, e)
print("C4 host-buffer entry point (H2D 160 MB + D2H 60 MB, pageable): %.2f ms = %.0f Mpoints/s"
      % (best * 1e3, len(c4["motor"]) / best / 1e6), flush=True)
del c4
c5 = synth.config_c5()
resident(c5["motor"], c5["eps_l1"], c5["min_pts"], "C5")
