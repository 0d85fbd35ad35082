"""Race stress: the union-find tolerates stale reads by design; the labels must nevertheless be identical from run
to run.  Repeats the 10 M-point calls and compares every output with the first."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtkcloudpoint_amd import _native as N  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402

ctx = N.Context(0)
n = 10_000_000
c = synth.config_cloud(n, seed=4)
for tag, arr, eps, metric, reps in (("L1_2D", c["motor"], c["eps_l1"], N.L1_2D, 40), ("L2_3D", c["xyz"], c["eps_l2"], N.L2_3D, 15)):
    d = torch.from_numpy(arr).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    ref = None
    bad = 0
    for r in range(reps):
        cf, ev = ctx.dbscan_dev(d.data_ptr(), n, arr.shape[1], eps, c["min_pts"], metric, 0, None, lab.data_ptr(), core.data_ptr())
        cur = (lab.clone(), core.clone(), cf, ev)
        if ref is None:
            ref = cur
        elif not (torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and cur[2:] == ref[2:]):
            bad += 1
    print("%s: %d runs, %d differ from the first (clusters %d)" % (tag, reps, bad, ref[2]), flush=True)
