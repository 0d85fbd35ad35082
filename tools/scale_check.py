"""One-off scale check: the C4 recipe at 100 M and 200 M points on one GPU (index widths, workspace sizes)."""
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
for n in (100_000_000, 200_000_000):
    t0 = time.time()
    c = synth.config_cloud(n, seed=4)
    m = c["motor"]
    del c
    print("generated %d points in %.0f s" % (n, time.time() - t0), flush=True)
    d = torch.from_numpy(m).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    best = None
    for _ in range(3):
        t = time.perf_counter()
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, 2, 0.1, 10, N.L1_2D, 0, None, lab.data_ptr(), core.data_ptr(), cls.data_ptr())
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    lmax = int(lab.max().item())
    ncore = int(core.sum(dtype=torch.int64).item())
    nlab = int((lab > 0).sum().item())
    # every labelled point is classed, every core point is labelled, ids are dense 1..cf
    ok = lmax == cf and bool(((lab > 0) == (cls > 0)).all().item()) and bool((lab[core > 0] > 0).all().item())
    uniq = int(torch.unique(lab).numel()) - 1
    print("n=%d: %.1f ms = %.0f Mpoints/s, clusters %d (max label %d, distinct %d), core %d, labelled %d, invariants %s, mem %.1f GB"
          % (n, best * 1e3, n / best / 1e6, cf, lmax, uniq, ncore, nlab, ok, torch.cuda.mem_get_info()[0] / 2**30), flush=True)
    print("   ", [(k, round(v, 2)) for k, v in ctx.timing()], flush=True)
    del d, lab, core, cls, m
    torch.cuda.empty_cache()
