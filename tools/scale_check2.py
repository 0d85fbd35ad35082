"""Scale checks of the other entry points: L2_3D at 100 M, block pipeline at 50 M, exact slabs 4 x 10 M against the
monolithic call on 40 M, centroids of 274 k clusters over 100 M points."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import distributed as D  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)


def timed(f, reps=2):
    best, out = None, None
    for _ in range(reps):
        t = time.perf_counter()
        out = f()
        e = time.perf_counter() - t
        best = e if best is None else min(best, e)
    return best, out


# 1. L2_3D at 100 M + centroids over the same labels
n = 100_000_000
c = synth.config_cloud(n, seed=4)
d3 = torch.from_numpy(c["xyz"]).cuda()
lab = torch.zeros(n, dtype=torch.int32, device="cuda")
t, (cf, ev) = timed(lambda: ctx.dbscan_dev(d3.data_ptr(), n, 3, c["eps_l2"], c["min_pts"], N.L2_3D, 0, None, lab.data_ptr()))
print("L2_3D n=%d: %.1f ms, clusters %d, max label %d" % (n, t * 1e3, cf, int(lab.max().item())), flush=True)
c3 = torch.zeros(cf, 3, dtype=torch.float64, device="cuda")
cnt = torch.zeros(cf, dtype=torch.int64, device="cuda")
t, _ = timed(lambda: ctx.centroids_dev(d3.data_ptr(), None, lab.data_ptr(), n, cf, c3.data_ptr(), None, cnt.data_ptr()))
ok = int(cnt.sum().item()) == int((lab > 0).sum().item()) and bool(torch.isfinite(c3).all().item())
print("centroids of %d clusters over %d points: %.1f ms, counts consistent %s" % (cf, n, t * 1e3, ok), flush=True)
del d3, lab, c3, cnt
torch.cuda.empty_cache()

# 2. block pipeline at 50 M (the first 50 M motor points)
m = torch.from_numpy(np.ascontiguousarray(c["motor"][:50_000_000])).cuda()
nb = m.shape[0]
local = torch.zeros(nb, dtype=torch.int32, device="cuda")
labels = torch.zeros(nb, dtype=torch.int32, device="cuda")


def blocks():
    info = ctx.blocks_begin(None, 0.07, 7, 200, 3, device_ptr=m.data_ptr(), n=nb)
    evb = ctx.blocks_cluster_dev(0, info["nblocks"], local.data_ptr())
    fin = ctx.blocks_finish_dev(local.data_ptr(), evb, labels.data_ptr())
    return info, fin


t, (info, fin) = timed(blocks)
print("block pipeline n=%d: %.1f ms, %d blocks, kept %d, clusters %d, max label %d"
      % (nb, t * 1e3, info["nblocks"], fin["kept"], fin["cluster_amount"], int(labels.max().item())), flush=True)
del m, local, labels
torch.cuda.empty_cache()

# 3. exact slabs: 4 x 10 M against the monolithic call on the same 40 M points
pts = np.ascontiguousarray(c["motor"][:40_000_000])
pts = pts[np.argsort(pts[:, 0], kind="stable")]
full = torch.from_numpy(pts).cuda()
lab = torch.zeros(len(pts), dtype=torch.int32, device="cuda")
t, (cf, ev) = timed(lambda: ctx.dbscan_dev(full.data_ptr(), len(pts), 2, 0.1, 10, N.L1_2D, 0, None, lab.data_ptr()), 1)
ctxs = [N.Context(0) for _ in range(4)]
parts = [full[k * 10_000_000:(k + 1) * 10_000_000] for k in range(4)]
t2, res = timed(lambda: D.exact_slabs_local(ctxs, parts, 0.1, 10, N.L1_2D), 1)
same = all(torch.equal(r["labels"], lab[k * 10_000_000:(k + 1) * 10_000_000]) for k, r in enumerate(res))
print("exact slabs 4 x 10 M vs monolithic 40 M: labels equal %s, cf %d == %d, evals equal %s, halos %s (mono %.1f ms, slabs in-process %.1f ms)"
      % (same, res[0]["cf"], cf, res[0]["dist_evals"] == ev, [r["halo"] for r in res], t * 1e3, t2 * 1e3), flush=True)
