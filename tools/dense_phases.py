"""Per-phase times of the 10 M-point C4 cloud at denser settings (eps 0.3 / 0.7: 1.8 / 9.8 expected background neighbours)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

n = 10_000_000
c = synth.config_cloud(n, seed=4)
ctx = N.Context(0)
ctx.timing_enable(True)
d = torch.from_numpy(c["motor"]).cuda()
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
core = torch.zeros(n, dtype=torch.uint8, device="cuda")
cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
for eps in [float(x) for x in (sys.argv[1:] or ["0.1", "0.3", "0.7"])]:
    for _ in range(3):
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, eps, 10, N.L1_2D, 0, None, lab.data_ptr(), core.data_ptr(), cls.data_ptr())
    t = ctx.timing()
    print("eps %.1f: %d clusters, %.2f ms: %s" % (eps, cf, sum(v for _, v in t), [(k, round(v, 3)) for k, v in t]), flush=True)
