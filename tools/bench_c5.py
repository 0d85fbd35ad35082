"""The 50 M-point C5 cloud, resident: wall time and phases (A/B of environment switches: tools/gpu/ab_c5.sh)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
ctx.timing_enable(True)
c5 = synth.config_c5()
cloud = c5["motor"]
n = len(cloud)
d = torch.from_numpy(cloud).cuda()
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()
best = None
for _ in range(5):
    t = time.perf_counter()
    cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, c5["eps_l1"], c5["min_pts"], N.L1_2D, 0, None, lab.data_ptr())
    e = time.perf_counter() - t
    best = e if best is None else min(best, e)
print("C5 %s: %.2f ms, %d clusters; %s" % (os.environ.get("VCP_CELL_BUDGET", "default"), best * 1e3, cf,
                                           [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)
