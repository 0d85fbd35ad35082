"""Race tolerance of the union-find, re-checked by the driver every round.  The component build reads parent words
through possibly stale caches by design (csrc/dbscan.hip: a stale value is an older ancestor link, device-scope CAS is
the only arbiter of roots), and the cell-order partition ranks points with LDS atomics, so the internal order may
differ from run to run; labels, core flags, cluster count and the iritatorNum counter must not.  Three calls per
configuration on device-resident inputs, every output compared with the first (one pass, not a loop until it fails)."""
import pytest
import torch

from vtkcloudpoint_amd import _native as N
from vtkcloudpoint_amd import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,which", [(10_000_000, "L1_2D"), (4_000_000, "L2_3D")])
def test_repeated_calls_identical(vcp_ctx, n, which):
    c = synth.config_cloud(n, seed=4 if which == "L1_2D" else 6)
    arr, eps, metric = ((c["motor"], c["eps_l1"], N.L1_2D) if which == "L1_2D" else (c["xyz"], c["eps_l2"], N.L2_3D))
    d = torch.from_numpy(arr).cuda()
    lab = torch.zeros(n, dtype=torch.int32, device="cuda")
    core = torch.zeros(n, dtype=torch.uint8, device="cuda")
    cls = torch.zeros(n, dtype=torch.uint8, device="cuda")
    torch.cuda.synchronize()
    ref = None
    for _ in range(3):
        cf, ev = vcp_ctx.dbscan_dev(d.data_ptr(), n, arr.shape[1], eps, c["min_pts"], metric, 0, None, lab.data_ptr(),
                                    core.data_ptr(), cls.data_ptr())
        cur = (lab.clone(), core.clone(), cls.clone(), cf, ev)
        if ref is None:
            ref = cur
            assert cf > 0
        else:
            assert torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1]) and torch.equal(cur[2], ref[2])
            assert cur[3:] == ref[3:]
