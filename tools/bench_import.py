"""AddFolder row work (filter, spherical conversion, duplicate removal) at 10 M rows, host buffers."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402

ctx = N.Context(0)
ctx.timing_enable(True)
rng = np.random.default_rng(1)
n = 10_000_000
rows = np.c_[rng.random(n) * 40, rng.random(n) * 40, rng.random(n) * 1100]
rows[5_000_000:6_000_000] = rows[:1_000_000]
for dedupe in (True, False):
    best = None
    for _ in range(3):
        t = time.perf_counter()
        r = ctx.import_convert(rows, 1.5, -0.5, 2, 1, dedupe)
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    print("dedupe=%s: %.1f ms (H2D 240 MB + D2H 250 MB included), kept %d, duplicates %d, phases %s"
          % (dedupe, best * 1e3, r["kept"], r["duplicates"], [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)
