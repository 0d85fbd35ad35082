"""Exact DBSCAN over several ranks (SURVEY.md 8e mode 2) on CPU: distributed.exact_slabs' per-rank program,
driven in-process with the oracle's staged backend, must reproduce ONE DBImproved.dbscan over the
rank-major concatenation -- labels, isKeyPoint, cf and iritatorNum -- for any way of cutting the cloud."""
import numpy as np
import pytest
import torch

from vtkcloudpoint_amd import distributed as D


def _cloud(rng, n, dim, snap, blobs=True):
    pts = rng.uniform(0, 10, (n, dim))
    if blobs:
        k = n // 2
        c = rng.uniform(1, 9, (6, dim))
        pts[:k] = c[rng.integers(0, 6, k)] + rng.normal(0, 0.3, (k, dim))
    pts = np.round(pts * snap) / snap  # multiples of 1/snap: exact d == eps ties are common
    return pts[rng.permutation(n)]


def _check(oracle, pts, cuts, eps, min_pts, metric, cf_in=0, literal=False):
    parts = [torch.from_numpy(np.ascontiguousarray(pts[a:b])) for a, b in zip(cuts, cuts[1:])]
    res = D.exact_slabs_local([oracle.StagedSlab() for _ in parts], parts, eps, min_pts, metric, cf_in)
    ref = oracle.dbscan(pts, eps, min_pts, metric=metric, cf_in=cf_in, literal=literal)
    lab = np.concatenate([r["labels"].numpy() for r in res])
    core = np.concatenate([r["is_core"].numpy() for r in res])
    cls = np.concatenate([r["is_classed"].numpy() for r in res])
    assert np.array_equal(lab, ref["labels"])
    assert np.array_equal(core, ref["is_key"])
    assert np.array_equal(cls, ref["classed"])
    for r in res:
        assert r["cf"] == ref["cf"] and r["dist_evals"] == ref["evals"]
    return res


@pytest.mark.parametrize("world", [1, 2, 3, 8])
@pytest.mark.parametrize("metric,dim", [(0, 2), (1, 2), (2, 3)])
def test_x_slabs_equal_monolithic(oracle, world, metric, dim):
    rng = np.random.default_rng(100 + world * 10 + metric)
    n = 3000
    pts = _cloud(rng, n, dim, 32.0)
    pts = pts[np.argsort(pts[:, 0], kind="stable")]
    cuts = [n * r // world for r in range(world + 1)]
    res = _check(oracle, pts, cuts, 0.25 if dim == 2 else 0.5, 5, metric, cf_in=7)
    if world > 1:
        assert max(r["halo"] for r in res) < n // 2  # slabs: the exchange is a boundary strip, not the cloud
        assert res[0]["boundary_pairs"] > 0


def test_against_literal_transcription(oracle):
    rng = np.random.default_rng(5)
    pts = _cloud(rng, 1500, 2, 16.0)
    pts = pts[np.argsort(pts[:, 0], kind="stable")]
    _check(oracle, pts, [0, 400, 900, 1500], 0.25, 4, 0, literal=True)


def test_arbitrary_distribution_and_empty_ranks(oracle):
    """Ownership need not be spatial (then everything is halo), ranks may be empty or tiny."""
    rng = np.random.default_rng(6)
    pts = _cloud(rng, 1200, 2, 32.0)
    _check(oracle, pts, [0, 0, 500, 500, 1199, 1200], 0.3, 4, 0)


def test_cluster_spanning_every_slab(oracle):
    """One chain of core points along x crosses all boundaries: a single global cluster, seed on rank 0."""
    x = np.arange(0, 400) * 0.125
    pts = np.stack([x, np.zeros_like(x)], axis=1)
    res = _check(oracle, pts, [0, 100, 200, 300, 400], 0.25, 3, 0)
    assert res[0]["cf"] == 1


def test_seed_owned_by_a_later_rank(oracle):
    """The global list order is rank-major, not spatial: rank 0 (first in the list) owns the right half here,
    rank 1 the left half, with an overlapping strip, so x order and list order disagree."""
    rng = np.random.default_rng(8)
    a = _cloud(rng, 800, 2, 32.0)
    a[:, 0] += 5.0          # rank 0 owns the right half
    b = _cloud(rng, 800, 2, 32.0)
    b[:, 0] -= 4.5          # rank 1 the left half, overlapping by a strip
    pts = np.concatenate([a, b])
    _check(oracle, pts, [0, 800, 1600], 0.3, 4, 0)


def test_eps_zero_and_min_pts_one(oracle):
    rng = np.random.default_rng(9)
    pts = np.round(rng.uniform(0, 4, (600, 2)) * 4) / 4  # many exact duplicates
    pts = pts[np.argsort(pts[:, 0], kind="stable")]
    _check(oracle, pts, [0, 200, 400, 600], 0.0, 2, 0)
    _check(oracle, pts, [0, 200, 400, 600], 0.25, 1, 0)


def test_rejects_bad_input(oracle):
    pts = torch.zeros((4, 2), dtype=torch.float64)
    with pytest.raises(ValueError):
        D.exact_slabs_local([oracle.StagedSlab()], [pts], -1.0, 3)
    bad = pts.clone()
    bad[0, 0] = float("nan")
    with pytest.raises(ValueError):
        D.exact_slabs_local([oracle.StagedSlab()], [bad], 0.1, 3)
