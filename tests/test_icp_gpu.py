"""GPU parity: libvcp.so ICP (through the C-ABI) vs the CPU oracle.  Tolerance 1e-5 on R, t, RMSE
(BASELINE.json north_star); nearest-neighbour indices are bit-exact."""
import numpy as np
import pytest

from vtkcloudpoint_amd import _native as N
from vtkcloudpoint_amd import synth

pytestmark = pytest.mark.gpu
TOL = 1e-5


def test_nn_and_sums_bit_exact_indices(vcp_ctx, oracle):
    d = synth.config_icp(nd=20000, nm=100, jitter=0.05)
    sums, nn = vcp_ctx.icp_sums(d["model"], d["data"])
    assert np.array_equal(nn, oracle.find_closest(d["model"], d["data"]))
    ref = oracle.icp_sums(d["model"], d["data"])
    assert np.allclose(sums, ref, rtol=1e-12, atol=1e-9)
    # with a transform: P = R data + T first (TransPoint), same op order as the C#
    R = synth.rotation_about((0, 1, 0), 3.0)
    T = np.array([0.1, 0.2, -0.3])
    sums, nn = vcp_ctx.icp_sums(d["model"], d["data"], R, T)
    P = oracle.trans_point(d["data"], R, T)
    assert np.array_equal(nn, oracle.find_closest(d["model"], P))
    assert np.allclose(sums, oracle.icp_sums(d["model"], P), rtol=1e-12, atol=1e-9)


def test_nn_tie_lowest_index(vcp_ctx, oracle):
    model = np.array([[0.0, 0, 0], [2.0, 0, 0], [0.0, 2, 0], [2.0, 0, 0]])
    data = np.array([[1.0, 0, 0], [1.0, 1.0, 0], [2.0, 0.0, 0.0], [5.0, 5.0, 5.0]])
    _, nn = vcp_ctx.icp_sums(model, data)
    assert nn.tolist() == [0, 0, 1, 1]
    assert np.array_equal(nn, oracle.find_closest(model, data))


@pytest.mark.parametrize("stop", [N.STOP_SSE_DELTA, N.STOP_RMSE])
def test_icp_noise_free_recovers_transform(vcp_ctx, oracle, stop):
    d = synth.config_icp(nd=50000, nm=100, jitter=0.0)
    g = vcp_ctx.icp(d["model"], d["data"], 1e-4, 100, stop)
    o = oracle.icp(d["model"], d["data"], 1e-4, 100, stop)
    assert g["iters"] == o["iters"]
    assert np.abs(g["R"] - o["R"]).max() < TOL and np.abs(g["T"] - o["T"]).max() < TOL
    assert abs(g["rmse"] - o["rmse"]) < TOL
    assert np.abs(g["R"] - d["R_true"]).max() < TOL and np.abs(g["T"] - d["T_true"]).max() < TOL
    assert g["rmse"] < 1e-4


def test_icp_c3_50_rounds(vcp_ctx, oracle):
    """C3: 1M data points vs 100-pt model, 50 rounds fixed (tol 0 never stops early)."""
    d = synth.config_icp(nd=1_000_000, nm=100, jitter=0.05)
    g = vcp_ctx.icp(d["model"], d["data"], 0.0, 50, N.STOP_SSE_DELTA)
    o = oracle.icp(d["model"], d["data"], 0.0, 50, N.STOP_SSE_DELTA)
    assert g["iters"] == 50 and o["iters"] == 50
    assert np.abs(g["R"] - o["R"]).max() < TOL and np.abs(g["T"] - o["T"]).max() < TOL
    assert abs(g["rmse"] - o["rmse"]) < TOL and abs(g["sse"] - o["sse"]) < 1e-6 * o["sse"]


def test_icp_errors(vcp_ctx):
    with pytest.raises(N.VcpError) as e:
        vcp_ctx.icp(np.zeros((0, 3)), np.zeros((5, 3)))
    assert e.value.code == -2
    # empty data: nothing to match, one round, R/T untouched (BaseClass/ICP.cs: loop body divides by 0)
    g = vcp_ctx.icp(np.zeros((3, 3)), np.zeros((0, 3)))
    assert g["iters"] == 1


def test_vtklike_icp_configuration(vcp_ctx, oracle):
    """MainForm.ICP()'s configuration (FrmMain.cs:851-862): 2-D inputs (x, y, 0), centroid start, landmark
    subsampling, 100 fixed rounds.  Checked against the oracle's restatement of the header-documented behaviour
    (VTK itself is a closed binary here: unpinned) and against the known planar transform."""
    rng = np.random.default_rng(5)
    truth = np.c_[rng.random((150, 2)) * 400, np.zeros(150)]
    t = np.deg2rad(1.5)
    Rz = np.array([[np.cos(t), -np.sin(t), 0], [np.sin(t), np.cos(t), 0], [0, 0, 1]])
    cent = (truth[rng.integers(0, 150, 900)] - [200, 200, 0]) @ Rz.T + [203, 198.5, 0]  # centroids = moved truths
    for ml in (200, 5000):
        g = vcp_ctx.icp_vtklike(cent, truth, 100, ml, True)
        o = oracle.icp_vtklike(cent, truth, 100, ml, True)
        assert g["iters"] == o["iters"] == 100
        assert np.abs(g["M"] - o["M"]).max() < TOL and abs(g["mean_dist"] - o["mean_dist"]) < TOL
        moved = cent @ g["M"][:3, :3].T + g["M"][:3, 3]
        nn = oracle.find_closest(truth, moved)
        assert np.abs(moved - truth[nn]).max() < 1e-6 and g["mean_dist"] < 1e-6
        assert abs(g["M"][2, 2] - 1) < 1e-9 and np.abs(g["M"][2, :2]).max() < 1e-9  # planar input: rotation about z


@pytest.mark.parametrize("nd", [3000, 70000])
def test_large_model_goes_through_lds_tiles(vcp_ctx, oracle, nd):
    """Model beyond the scalar-cache path (nm > 512): LDS tiles, and one-wave workgroups for the small data set.
    Coordinates on a coarse lattice produce many exact distance ties (lowest model index must win) and a large
    absolute offset makes the binary32 screening ambiguous for most points (second sweep)."""
    rng = np.random.default_rng(nd)
    model = rng.integers(0, 24, size=(2500, 3)).astype(np.float64) * 0.5 + 1000.0   # duplicates included
    data = rng.integers(0, 48, size=(nd, 3)).astype(np.float64) * 0.25 + 1000.0
    sums, nn = vcp_ctx.icp_sums(model, data)
    assert np.array_equal(nn, oracle.find_closest(model, data))
    assert np.allclose(sums, oracle.icp_sums(model, data), rtol=1e-12, atol=1e-6)
    # K x K form of MainForm.ICP: centroids against a rotated + shifted copy
    cen = np.round(rng.uniform(0, 200.0, (2000, 3)) * 1024) / 1024
    truth = cen @ synth.rotation_about((1.0, 1.0, 1.0), 0.2).T + np.array([0.3, -0.2, 0.1])
    g = vcp_ctx.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
    o = oracle.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
    assert g["iters"] == o["iters"]
    assert np.abs(g["R"] - o["R"]).max() < TOL and np.abs(g["T"] - o["T"]).max() < TOL and g["rmse"] < 1e-9


def test_grid_nn_30k_x_30k_tie_heavy(vcp_ctx, oracle):
    """Models beyond 512 points are binned and searched through the grid (csrc/nngrid.hpp): the index must still be the
    one FindClosestPointSet's sequential strict-`<` scan returns (BaseClass/ICP.cs:224-250).  30 k x 30 k on a coarse
    lattice: thousands of exact distance ties and duplicated model points; plus queries far outside the model (ring
    search runs out and falls back to the whole set), planar and collinear models (degenerate grid axes)."""
    rng = np.random.default_rng(30)
    model = rng.integers(0, 60, size=(30_000, 3)).astype(np.float64) * 0.5 + 500.0
    data = rng.integers(0, 120, size=(30_000, 3)).astype(np.float64) * 0.25 + 500.0
    data[:200] += 1000.0  # far outside the model's box
    data[200:300] -= 750.0
    sums, nn = vcp_ctx.icp_sums(model, data)
    assert np.array_equal(nn, oracle.find_closest(model, data))
    assert np.allclose(sums, oracle.icp_sums(model, data), rtol=1e-12, atol=1e-3)
    for flat in (1, 2):  # planar (z constant), collinear (y and z constant)
        m2 = model[:5000].copy()
        m2[:, 3 - flat:] = 7.0
        d2 = data[:5000].copy()
        _, nn2 = vcp_ctx.icp_sums(m2, d2)
        assert np.array_equal(nn2, oracle.find_closest(m2, d2))
    same = np.repeat(model[:1], 1000, axis=0)  # every model point identical: one cell, index 0 wins everywhere
    _, nn3 = vcp_ctx.icp_sums(same, data[:2000])
    assert not nn3.any()
    # real-valued clouds (no ties), clustered: dense and empty cells side by side
    model = np.concatenate([rng.normal(0, 1, (20_000, 3)), rng.normal(40, 0.05, (10_000, 3))])
    data = np.concatenate([rng.normal(0, 3, (15_000, 3)), rng.normal(40, 0.2, (15_000, 3))])
    _, nn4 = vcp_ctx.icp_sums(model, data)
    assert np.array_equal(nn4, oracle.find_closest(model, data))
    # the whole K x K ICP through the grid
    cen = np.round(rng.uniform(0, 300.0, (30_000, 3)) * 1024) / 1024
    truth = cen @ synth.rotation_about((1.0, 1.0, 1.0), 0.05).T + np.array([0.03, -0.02, 0.01])
    g = vcp_ctx.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
    o = oracle.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
    assert g["iters"] == o["iters"]
    assert np.abs(g["R"] - o["R"]).max() < TOL and np.abs(g["T"] - o["T"]).max() < TOL and abs(g["rmse"] - o["rmse"]) < TOL


def test_non_finite_model_keeps_full_scan(vcp_ctx, oracle):
    rng = np.random.default_rng(4)
    model = rng.random((3000, 3)) * 10
    model[17, 1] = np.inf
    data = rng.random((5000, 3)) * 10
    _, nn = vcp_ctx.icp_sums(model, data)
    assert np.array_equal(nn, oracle.find_closest(model, data))
