"""refreshClusList (truth-guided assignment) at scale: 10 M raw points against T truths (host-buffer API)."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
ctx.timing_enable(True)
c = synth.config_cloud(10_000_000, seed=4)
m = c["motor"]
for T in (200, 4000):
    rng = np.random.default_rng(T)
    truths = np.round(rng.uniform(0, c["motor_extent"], (T, 2)) * 1024) / 1024
    ids = np.arange(1, T + 1, dtype=np.int32)
    best = None
    for _ in range(3):
        t = time.perf_counter()
        out, outl = ctx.assign_truths(m, truths, ids, 2.0)
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("T=%d: %.1f ms (host buffers, 10 M points), assigned %d, phases %s" % (T, best * 1e3, int((out > 0).sum()),
                                                                            [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)
