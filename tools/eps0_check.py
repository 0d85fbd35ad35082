import sys, time, numpy as np
sys.path.insert(0, '/root/repo')
from vtkcloudpoint_amd import _native as N
ctx = N.Context(0); ctx.timing_enable(True)
rng = np.random.default_rng(3)
n = 695874
for scale in (1e3, 1e6, 1e9):
    c = rng.normal(0, 1.0, (n, 3))
    c[rng.integers(0, n, n // 1000)] *= scale
    for eps in (0.0, 1e-3):
        t = time.time(); g = ctx.dbscan(c, eps, 3, 0); e = time.time() - t
        print("scale %g eps %g: %.3f s cf=%d phases %s" % (scale, eps, e, g["cf"], [(k, round(v, 2)) for k, v in ctx.timing()]), flush=True)
