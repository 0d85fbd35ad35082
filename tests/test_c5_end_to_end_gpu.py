"""BASELINE.json config 5 on the GPU: the 50 M-point scan-like cloud end to end -- DBSCAN -> per-cluster centroids ->
ICP of the centroids against a rotated + shifted copy ("truth") -> matching -- every stage compared with the CPU
oracle on the same inputs.  Reference path: FrmMain.cs:1214-1291 + :1442-1533 (clustering + GetClusList,
BC/Tools.cs:162-195), BC/ICP.cs:18-285, FrmMain.cs:3572-3618 (matching).

The clustering is the monolithic DBImproved.dbscan on (motor_x, motor_y) (BC/DBImproved.cs:91-114); the literal O(n^2)
port cannot run at this size, the oracle's order-free formulation (proved equal to the literal one on small inputs,
tests/test_oracle_dbscan.py) takes ~90 s.  Bars: labels / core flags / counters bit-exact, centroids 1e-12 relative,
ICP R, t, RMSE within 1e-5 (north_star), matching bit-exact."""
import numpy as np
import pytest

from vtkcloudpoint_amd import _native as N
from vtkcloudpoint_amd import synth

pytestmark = pytest.mark.gpu


def test_c5_50m_dbscan_centroids_icp_match(vcp_ctx, oracle):
    d = synth.config_c5()
    n = len(d["motor"])
    assert n == 50_000_000
    eps, mp = d["eps_l1"], d["min_pts"]

    # 1. clustering
    g = vcp_ctx.dbscan(d["motor"], eps, mp, N.L1_2D)
    o = oracle.dbscan(d["motor"], eps, mp, oracle.L1_2D)
    assert np.array_equal(g["labels"], o["labels"])
    assert np.array_equal(g["is_core"], o["is_key"]) and np.array_equal(g["is_classed"], o["classed"])
    assert g["cf"] == o["cf"] and g["evals"] == o["evals"]
    K = g["cf"]
    assert K > 1000  # every object blob yields at least its core

    # 2. centroids (Tools.GetClusList): 3-D and 2-D means, counts
    g3, g2, gcnt = vcp_ctx.centroids(d["xyz"], d["motor"], g["labels"], K)
    o3, o2, ocnt = oracle.centroids(d["xyz"], d["motor"], o["labels"], K)
    assert np.array_equal(gcnt, ocnt) and int(gcnt.min()) >= 1
    assert np.allclose(g3, o3, rtol=1e-12, atol=1e-12)
    assert np.allclose(g2, o2, rtol=1e-12, atol=1e-12)

    # 3. ICP of the centroids (data) against a rotated + shifted copy (model = "truth"), ICP.go_hell_ICP's intended
    #    arithmetic; the first round's correspondences are already the right ones at this rotation
    cen = np.ascontiguousarray(g3)
    Rt = synth.rotation_about((1.0, 1.0, 1.0), 0.0005)
    shift = np.array([0.003, -0.002, 0.001])
    truth = np.ascontiguousarray(cen @ Rt.T + shift)
    gi = vcp_ctx.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
    oi = oracle.icp(truth, cen, 1e-9, 100, oracle.STOP_SSE_DELTA)
    assert gi["iters"] == oi["iters"]
    assert np.abs(gi["R"] - oi["R"]).max() < 1e-5 and np.abs(gi["T"] - oi["T"]).max() < 1e-5
    assert abs(gi["rmse"] - oi["rmse"]) < 1e-5
    if oi["rmse"] < 1e-6:  # the oracle found the true correspondences (centroids of fringe fragments can lie closer
        # together than the displacement): then so must the GPU, and the transform is the one applied
        assert np.abs(gi["R"] - Rt).max() < 1e-5 and np.abs(gi["T"] - shift).max() < 1e-5 and gi["rmse"] < 1e-4

    # 4. matching (calMatchedCoords + RecorrectMatchingPtsByDistance) with the recovered transform
    M = np.eye(4)
    M[:3, :3] = gi["R"]
    M[:3, 3] = gi["T"]
    gm = vcp_ctx.match(cen, truth, M, 0.5)
    om = oracle.match(cen, truth, M, 0.5)
    for k in ("is_matched", "nearest", "nearest_dist", "matched_xyz"):
        assert np.array_equal(gm[k], om[k]), k
    assert gm["count"] == om["count"]
