"""Exact DBSCAN over several GPUs (SURVEY.md 8e mode 2) on one GPU: every "rank" is its own vcp context, the
exchange of distributed.exact_slabs is simulated in-process (exact_slabs_local).  The staged HIP engine
(vcp_slab_begin / _comps / _finish) must reproduce ONE DBImproved.dbscan over the whole list bit for bit:
against the oracle at small sizes, against the monolithic HIP call at 4 M points."""
import numpy as np
import pytest
import torch

from vtkcloudpoint_amd import _native as N
from vtkcloudpoint_amd import distributed as D
from vtkcloudpoint_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctxs():
    cs = [N.Context(0) for _ in range(4)]
    yield cs
    for c in cs:
        c.close()


def _run(ctxs, pts, cuts, eps, min_pts, metric, cf_in=0):
    parts = [torch.from_numpy(np.ascontiguousarray(pts[a:b])).cuda() for a, b in zip(cuts, cuts[1:])]
    res = D.exact_slabs_local(ctxs[:len(parts)], parts, eps, min_pts, metric, cf_in)
    lab = np.concatenate([r["labels"].cpu().numpy() for r in res])
    core = np.concatenate([r["is_core"].cpu().numpy() for r in res])
    cls = np.concatenate([r["is_classed"].cpu().numpy() for r in res])
    return res, lab, core, cls


@pytest.mark.parametrize("metric,dim,eps", [(N.L1_2D, 2, 0.25), (N.L2_2D, 2, 0.25), (N.L2_3D, 3, 0.5)])
@pytest.mark.parametrize("world", [2, 4])
def test_slabs_vs_oracle(ctxs, oracle, metric, dim, eps, world):
    rng = np.random.default_rng(40 + world + metric)
    n = 30_000
    pts = rng.uniform(0, 30, (n, dim))
    k = n // 2
    c = rng.uniform(2, 28, (12, dim))
    pts[:k] = c[rng.integers(0, 12, k)] + rng.normal(0, 0.5, (k, dim))
    pts = np.round(pts * 32) / 32  # exact d == eps ties
    pts = pts[np.argsort(pts[:, 0], kind="stable")]
    cuts = [n * r // world for r in range(world + 1)]
    res, lab, core, cls = _run(ctxs, pts, cuts, eps, 5, metric, cf_in=11)
    ref = oracle.dbscan(pts, eps, 5, metric=metric, cf_in=11)
    assert np.array_equal(lab, ref["labels"])
    assert np.array_equal(core, ref["is_key"]) and np.array_equal(cls, ref["classed"])
    for r in res:
        assert r["cf"] == ref["cf"] and r["dist_evals"] == ref["evals"]
    assert max(r["halo"] for r in res) < n // 4 and res[0]["boundary_pairs"] > 0


def test_arbitrary_ownership_and_empty_rank(ctxs, oracle):
    rng = np.random.default_rng(50)
    pts = np.round(rng.uniform(0, 8, (5000, 2)) * 16) / 16
    res, lab, core, cls = _run(ctxs, pts, [0, 0, 2000, 4999, 5000], 0.25, 4, N.L1_2D)
    ref = oracle.dbscan(pts, 0.25, 4)
    assert np.array_equal(lab, ref["labels"]) and np.array_equal(core, ref["is_key"])
    assert res[2]["cf"] == ref["cf"] and res[2]["dist_evals"] == ref["evals"]


def test_chain_through_all_slabs(ctxs):
    x = np.arange(0, 4000) * 0.125
    pts = np.stack([x, np.zeros_like(x)], axis=1)
    res, lab, core, cls = _run(ctxs, pts, [0, 1000, 2000, 3000, 4000], 0.25, 3, N.L1_2D)
    assert res[0]["cf"] == 1 and np.all(lab == 1) and core.all()


def test_4m_slabs_equal_monolithic_gpu(ctxs, vcp_ctx):
    """C4-style cloud at 4 M points cut into 4 x-slabs vs the single-GPU call on the whole list."""
    d = synth.config_cloud(4_000_000, seed=44)
    pts = d["motor"]
    pts = np.ascontiguousarray(pts[np.argsort(pts[:, 0], kind="stable")])
    n = len(pts)
    mono = vcp_ctx.dbscan(pts, d["eps_l1"], d["min_pts"], N.L1_2D)
    cuts = [n * r // 4 for r in range(5)]
    res, lab, core, cls = _run(ctxs, pts, cuts, d["eps_l1"], d["min_pts"], N.L1_2D)
    assert np.array_equal(lab, mono["labels"])
    assert np.array_equal(core, mono["is_core"]) and np.array_equal(cls, mono["is_classed"])
    assert res[0]["cf"] == mono["cf"] and res[0]["dist_evals"] == mono["evals"]
    assert mono["cf"] > 1000 and max(r["halo"] for r in res) < n // 50


def test_staged_engine_errors(vcp_ctx):
    c = vcp_ctx
    lab = torch.zeros(8, dtype=torch.int32, device="cuda")
    c.dbscan(np.zeros((4, 2)), 0.1, 2)  # any engine call drops a previous staged state
    with pytest.raises(N.VcpError):
        c.slab_finish([], [], [], [], 0, 4, lab.data_ptr())
    pts = torch.tensor([[0.0, 0.0], [0.05, 0.0], [0.1, 0.0], [5.0, 5.0]], dtype=torch.float64, device="cuda")
    ordv = torch.arange(4, dtype=torch.int32, device="cuda")
    rep = torch.zeros(4, dtype=torch.int32, device="cuda")
    assert c.slab_begin(pts.data_ptr(), 4, 2, N.L1_2D, 0.1, 2, None, ordv.data_ptr(), rep.data_ptr()) == 1
    assert rep.cpu().tolist() == [0, 0, 0, -1]
    assert c.slab_comps().tolist() == [0]
    with pytest.raises(N.VcpError):  # the map does not cover the local component
        c.slab_finish([3], [0], [1], [3], 0, 4, lab.data_ptr())
    with pytest.raises(N.VcpError):  # eps must be finite and >= 0 for a staged call
        c.slab_begin(pts.data_ptr(), 4, 2, N.L1_2D, -1.0, 2, None, ordv.data_ptr(), rep.data_ptr())
