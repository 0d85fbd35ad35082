"""ICP of K centroids against K truths (the reference's real use of ICP, MainForm.ICP: cluster centroids vs the
truth list): K x K brute-force nearest neighbour per round."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
for K in (4000, 27380):
    rng = np.random.default_rng(K)
    cen = np.round(rng.uniform(0, 215.0, (K, 3)) * 1024) / 1024
    Rt = synth.rotation_about((1.0, 1.0, 1.0), 0.2)
    truth = cen @ Rt.T + np.array([0.3, -0.2, 0.1])
    best = None
    for _ in range(4):
        t = time.perf_counter()
        r = ctx.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("K=%d: %.2f ms, %d rounds, rmse %.2e" % (K, best * 1e3, r["iters"], r["rmse"]), flush=True)
    M = np.eye(4)
    best = None
    for _ in range(6):
        t = time.perf_counter()
        m = ctx.match(cen, truth, M, 0.5)
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("K=%d: match %.3f ms" % (K, best * 1e3), flush=True)
