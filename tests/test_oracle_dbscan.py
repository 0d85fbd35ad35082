"""The oracle pinned on CPU: hand-derived known answers, literal == order-free formulation on random inputs
with exact ties, scikit-learn's DBSCAN for the core set / core partition, and the regression fixtures."""
import json
import os

import numpy as np
import pytest

from vtkcloudpoint_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def _cases():
    with open(os.path.join(HERE, "golden", "micro_cases.json")) as f:
        return json.load(f)["cases"]


def _coords(c):
    if "xy" in c:
        return np.array(c["xy"], np.float64)
    return np.stack([np.array(c["x"], np.float64), np.zeros(len(c["x"]))], 1)


@pytest.mark.parametrize("case", _cases(), ids=lambda c: c["name"])
@pytest.mark.parametrize("literal", [True, False], ids=["literal", "canonical"])
def test_hand_derived_known_answers(oracle, case, literal):
    coords = _coords(case)
    cls = np.array(case["classed_in"], np.uint8) if "classed_in" in case else None
    lab = np.array(case["labels_in"], np.int32) if "labels_in" in case else None
    r = oracle.dbscan(coords, case["eps"], case["min_pts"], case["metric"], case["cf_in"], cls, lab, literal=literal)
    assert r["labels"].tolist() == case["labels"]
    assert r["classed"].tolist() == case["classed"]
    assert r["is_key"].tolist() == case["is_key"]
    assert r["cf"] == case["cf"] and r["evals"] == case["evals"]


def test_literal_equals_canonical_random_with_ties(oracle):
    rng = np.random.default_rng(0)
    for trial in range(400):
        n = int(rng.integers(1, 120))
        metric = int(rng.integers(0, 3))
        c = rng.integers(0, 12, size=(n, 3)).astype(np.float64) * 0.25  # quantised: many d == eps ties
        eps = float(rng.choice([0.25, 0.5, 0.75, 1.0, 0.0]))
        mp = int(rng.integers(1, 7))
        cf = int(rng.integers(0, 5))
        cls = (rng.random(n) < 0.15).astype(np.uint8) if trial % 3 == 0 else None
        lab0 = (rng.integers(1, 4, n) * cls).astype(np.int32) if cls is not None else None
        a = oracle.dbscan(c, eps, mp, metric, cf, cls, lab0, literal=True)
        b = oracle.dbscan(c, eps, mp, metric, cf, cls, lab0, literal=False)
        for k in ("labels", "classed", "is_key"):
            assert np.array_equal(a[k], b[k]), (trial, k)
        assert a["cf"] == b["cf"] and a["evals"] == b["evals"], trial


def test_degenerate_parameters(oracle):
    rng = np.random.default_rng(3)
    c = rng.integers(0, 6, size=(30, 2)).astype(np.float64)
    for eps, mp in [(-1.0, 0), (-1.0, 3), (0.5, 0), (0.5, -2), (float("nan"), 0), (float("inf"), 3)]:
        a = oracle.dbscan(c, eps, mp, 0, 2, literal=True)
        b = oracle.dbscan(c, eps, mp, 0, 2)
        assert all(np.array_equal(a[k], b[k]) for k in ("labels", "classed", "is_key")), (eps, mp)
        assert a["cf"] == b["cf"] and a["evals"] == b["evals"]
    # eps < 0 with minPts <= 0: every point its own cluster, isClassed stays false (expandCluster on an
    # empty neighbour list only sets p.clusterId, DBImproved.cs:58)
    a = oracle.dbscan(c, -1.0, 0, 0, 2, literal=True)
    assert a["labels"].tolist() == list(range(3, 33)) and not a["classed"].any()


def test_dead_dedupe_scan_changes_nothing(oracle):
    d = synth.make_cloud(600, 5, 2, 200, 10.0, 0.5, motor_sigma=0.3, motor_bg_density=2.0)
    a = oracle.dbscan(d["motor"], 0.2, 4, literal=True, dedupe=False)
    b = oracle.dbscan(d["motor"], 0.2, 4, literal=True, dedupe=True)
    assert np.array_equal(a["labels"], b["labels"]) and a["evals"] == b["evals"]


def test_sklearn_core_set_and_core_partition(oracle):
    sk = pytest.importorskip("sklearn.cluster")
    d = synth.config_c1()
    for eps, mp in ((0.5, 10), (0.25, 6)):
        r = oracle.dbscan(d["motor"], eps, mp)
        m = sk.DBSCAN(eps=eps, min_samples=mp, metric="manhattan").fit(d["motor"])
        core = np.zeros(len(d["motor"]), bool)
        core[m.core_sample_indices_] = True
        assert np.array_equal(core, r["is_key"].astype(bool))
        # same partition of the core points (border assignment legitimately differs: highest id wins here)
        a, b = r["labels"][core], m.labels_[core]
        pairs = set(zip(a.tolist(), b.tolist()))
        assert len(pairs) == len(set(a.tolist())) == len(set(b.tolist()))
        # noise = points within eps of no core point, on both sides
        assert np.array_equal(r["labels"] == 0, m.labels_ == -1)


def test_regression_fixture_c1(oracle):
    g = np.load(os.path.join(HERE, "golden", "c1_dbscan.npz"))
    import hashlib
    d = synth.config_c1()
    assert hashlib.sha256(d["motor"].tobytes()).hexdigest() == str(g["motor_sha"])
    assert hashlib.sha256(d["xyz"].tobytes()).hexdigest() == str(g["xyz_sha"])
    for lit in (True, False):
        l1 = oracle.dbscan(d["motor"], d["eps_l1"], d["min_pts"], oracle.L1_2D, literal=lit)
        assert np.array_equal(l1["labels"], g["l1_labels"]) and np.array_equal(l1["is_key"], g["l1_key"])
        assert l1["cf"] == int(g["l1_cf"]) and l1["evals"] == int(g["l1_evals"])
        l2 = oracle.dbscan(d["xyz"], d["eps_l2"], d["min_pts"], oracle.L2_3D, literal=lit)
        assert np.array_equal(l2["labels"], g["l2_labels"]) and l2["evals"] == int(g["l2_evals"])


def test_db_dead_class_literal(oracle):
    """BaseClass/DB.cs: signed metric dx+dy (asymmetric) and the ifShown filter; hand-checked tiny case."""
    # getDisP(p, q) = (px-qx)+(py-qy) <= e  <=>  s_q >= s_p - e with s = x+y
    c = np.array([[0.0, 0.0], [1.0, 0.0], [2.0, 0.0], [9.0, 0.0]])
    r = oracle.db_literal(c, 0.5, 3)
    # s = 0,1,2,9: p0 sees everyone (4 >= 3): cluster 1 pulls in all four points
    assert r["labels"].tolist() == [1, 1, 1, 1] and r["cluster_amount"] == 1 and r["points_amount"] == 4
    shown = np.array([1, 1, 0, 1], np.uint8)
    r = oracle.db_literal(c, 0.5, 3, shown)
    assert r["labels"].tolist() == [1, 1, 0, 1] and r["points_amount"] == 3


def test_literal_equals_canonical_with_non_finite_points_and_min_pts_le_0(oracle):
    """A point with a NaN / infinite coordinate is nobody's neighbour, not even its own.  With minPts <= 0 it still seeds
    a cluster (0 >= minPts, DBImproved.cs:105), takes the id (:58), but is never popped from its empty list: not classed
    (:63-65) and queried once, not twice.  The order-free formulation must say the same."""
    rng = np.random.default_rng(3)
    for t in range(300):
        n = int(rng.integers(5, 60))
        c = rng.integers(0, 6, size=(n, 2)).astype(np.float64)
        k = rng.integers(0, n, 3)
        c[k[0]] = np.nan
        c[k[1], 0] = np.inf
        mp = int(rng.integers(-1, 3))
        cls = (rng.random(n) < 0.3).astype(np.uint8)
        lab = (cls * 7).astype(np.int32)
        a = oracle.dbscan(c, 1.0, mp, 0, 2, cls, lab, literal=True)
        b = oracle.dbscan(c, 1.0, mp, 0, 2, cls, lab, literal=False)
        for key in ("labels", "classed", "is_key"):
            assert np.array_equal(a[key], b[key]), (t, key)
        assert a["cf"] == b["cf"] and a["evals"] == b["evals"], t
