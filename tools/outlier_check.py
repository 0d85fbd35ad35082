"""Robust grid range: the C4 cloud with a few points 10^9 away must cost about the same as without them."""
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
for n in (1_000_000, 10_000_000):
    c = synth.config_cloud(n, seed=4)
    m = c["motor"].copy()
    ref = None
    for tag in ("clean", "outliers"):
        if tag == "outliers":
            m[123] = (1e9, -3e8)
            m[4567] = (-7e8, 2e9)
            m[99999] = (5e5, 5e5)
        d = torch.from_numpy(m).cuda()
        lab = torch.zeros(n, dtype=torch.int32, device="cuda")
        torch.cuda.synchronize()
        best = None
        for _ in range(3):
            t = time.perf_counter()
            cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, 0.1, 10, N.L1_2D, 0, None, lab.data_ptr())
            e = time.perf_counter() - t
            best = e if best is None else min(best, e)
        print("n=%d %s: %.2f ms, %d clusters" % (n, tag, best * 1e3, cf), flush=True)
