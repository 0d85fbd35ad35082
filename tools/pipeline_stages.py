"""Wall time of every stage of the sharded pipeline's per-rank program at ONE rank (10 M points, reference defaults):
where the 1.4 ms over the single-device entry points go."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import distributed as D  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

n = 10_000_000
c = synth.config_cloud(n, seed=4)
ctx = N.Context(0)
d = torch.from_numpy(c["motor"]).cuda()
labels = torch.zeros(n, dtype=torch.int32, device="cuda")
torch.cuda.synchronize()


class Timed:
    def __init__(self, c):
        self.c, self.t = c, {}

    def __getattr__(self, name):
        f = getattr(self.c, name)

        def g(*a, **k):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = f(*a, **k)
            torch.cuda.synchronize()
            self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
            return r
        return g


for it in range(4):
    tc = Timed(ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = D.sharded_pipeline(tc, d.data_ptr(), n, 0.07, 7, 200, 3, device="cuda", labels=labels)
    torch.cuda.synchronize()
    tot = time.perf_counter() - t0
print("total %.3f ms; library calls: %s; outside them: %.3f ms" % (
    tot * 1e3, {k: round(v * 1e3, 3) for k, v in tc.t.items()}, (tot - sum(tc.t.values())) * 1e3))
