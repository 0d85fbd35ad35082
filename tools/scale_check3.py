"""Path coverage at sizes no test reaches: (1) 140 M points in a dense uniform cloud -- the partition build applies
(1e8 cells) but n > 2^27, so the OUTPUT is the gather form fed by the partition's `pos`; (2) the same cloud through the
sort-based build (VCP_BUILD_SORT=1, the round-1 path).  The two label arrays must be identical."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 140_000_000
ext = 1000.0
xy = np.empty((n, 2))
for a in range(2):
    xy[:, a] = synth.snap(ext * synth.uniform01(901 + a, 0, n))
ctx = N.Context(0)
ctx.timing_enable(True)
d = torch.from_numpy(xy).cuda()
del xy
out = {}
for tag, env in (("partition", None), ("sort", "1")):
    if env:
        os.environ["VCP_BUILD_SORT"] = env
    else:
        os.environ.pop("VCP_BUILD_SORT", None)
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    best = None
    for _ in range(2):
        t = time.perf_counter()
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, 0.1, 8, N.L1_2D, 0, None, lab.data_ptr(), core.data_ptr())
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    out[tag] = (lab, core, cf, ev)
    print("%s build: n=%d %.1f ms = %.0f Mpoints/s, %d clusters; phases %s"
          % (tag, n, best * 1e3, n / best / 1e6, cf, [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)
a, b = out["partition"], out["sort"]
print("labels equal %s, core flags equal %s, cf %d == %d, evals equal %s"
      % (bool(torch.equal(a[0], b[0])), bool(torch.equal(a[1], b[1])), a[2], b[2], a[3] == b[3]), flush=True)
