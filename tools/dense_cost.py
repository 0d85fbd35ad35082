"""Dense regime: every point within eps of every other one (plus a few far outliers so that the degenerate O(n) path
does not apply) -- what the union scan costs with the chunk summary (dbscan.hip: k_chunkroot)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402

ctx = N.Context(0)
ctx.timing_enable(True)
rng = np.random.default_rng(2)
for n in (100_000, 445_000):
    c = rng.normal(0, 1.0, (n, 3))
    c[rng.integers(0, n, n // 1000)] *= 1e6
    for eps, mp in ((233.0, 2), (0.5, 10)):
        best = 1e9
        for _ in range(2):
            t = time.perf_counter()
            g = ctx.dbscan(c, eps, mp, N.L2_3D)
            best = min(best, time.perf_counter() - t)
        print("n=%d eps=%g minPts=%d: %.1f ms, %d clusters; phases %s"
              % (n, eps, mp, best * 1e3, g["cf"], [(k, round(v, 2)) for k, v in ctx.timing() if v >= 0.5]), flush=True)
