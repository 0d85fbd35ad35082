"""What the exact re-test through the index gather costs when EVERY neighbour pair sits on the threshold (integer
lattice, eps = the step) against a real-valued cloud of the same size and density."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402

ctx = N.Context(0)
n = 4_000_000
rng = np.random.default_rng(1)
side = 2000  # one point per lattice site on average
for name, c in (("lattice", rng.integers(0, side, (n, 2)).astype(np.float64)),
                ("real   ", rng.random((n, 2)) * side)):
    d = torch.from_numpy(c).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, 1.0, 4, N.L1_2D, 0, None, lab.data_ptr())
        best = min(best, time.perf_counter() - t)
    print("%s n=%d eps=1 minPts=4: %.2f ms, %d clusters" % (name, n, best * 1e3, cf), flush=True)
