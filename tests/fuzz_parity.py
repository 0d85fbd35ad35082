"""Randomised parity sweep on the GPU box (drives the oracle, hence under tests/): as a script an open-ended sweep, and
through run() the bounded, fixed-seed form that tests/test_fuzz_gpu.py runs under `pytest -m gpu` (one pass, per-case
GPU time bound).  libvcp (through the C-ABI) against the order-free CPU oracle on clouds no
fixed test has -- sizes from one point to a few million, uniform / clustered / lattice / duplicate-heavy / collinear
shapes, far outliers, non-finite coordinates, every metric, isClassed inputs, cf presets -- plus the block pipeline.
usage: python tests/fuzz_parity.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import binding as O  # noqa: E402  (test tooling: the oracle is the checker)
from vtkcloudpoint_amd import _native as N  # noqa: E402

rng = None   # set by run()
ctx = None
done = {"dbscan": 0, "blocks": 0}
QUIET = [False]


class Mismatch(AssertionError):
    pass


def fail(msg):
    print(msg, flush=True)
    raise Mismatch(msg)


def say(*a, **k):
    if not QUIET[0]:
        print(*a, **k)


KIND = [0]
NN_LOG = [4.6]   # log10 of the largest model / data set of a nearest-neighbour case


def cloud(n, dim):
    kind = int(rng.integers(0, 7))
    KIND[0] = kind
    ext = float(rng.choice([1.0, 10.0, 300.0, 1e4]))
    if kind == 0:
        c = rng.uniform(0, ext, (n, dim))
    elif kind == 1:  # blobs on a sparse background
        k = max(1, n // int(rng.integers(50, 5000)))
        cen = rng.uniform(0, ext, (k, dim))
        c = cen[rng.integers(0, k, n)] + rng.normal(0, ext * float(rng.choice([1e-3, 1e-2])), (n, dim))
        bg = rng.random(n) < 0.3
        c[bg] = rng.uniform(0, ext, (int(bg.sum()), dim))
    elif kind == 2:  # lattice: exact ties, duplicates
        side = max(2, int(rng.integers(2, 2000)))
        c = rng.integers(0, side, (n, dim)).astype(np.float64) * float(rng.choice([0.25, 1.0, 3.0]))
    elif kind == 3:  # heavy duplicates
        k = max(1, n // int(rng.integers(2, 200)))
        c = rng.uniform(0, ext, (k, dim))[rng.integers(0, k, n)]
    elif kind == 4:  # collinear / degenerate axis
        c = rng.uniform(0, ext, (n, dim))
        c[:, int(rng.integers(0, dim))] = float(rng.uniform(-5, 5))
    elif kind == 5:  # far outliers coarsen / trim the grid
        c = rng.normal(0, 1.0, (n, dim))
        m = max(1, n // 1000)
        c[rng.integers(0, n, m)] *= float(rng.choice([1e3, 1e6, 1e9]))
    else:  # quantised to 2^-10 like the benchmark clouds, off the origin
        c = np.round(rng.uniform(0, ext, (n, dim)) * 1024) / 1024 + float(rng.choice([0.0, 12345.5, -7e5]))
    if rng.random() < 0.1 and n > 3:
        bad = rng.integers(0, n, max(1, n // 500))
        c[bad, int(rng.integers(0, dim))] = float(rng.choice([np.nan, np.inf, -np.inf]))
    return np.ascontiguousarray(c)


def nn_case():
    """nearest neighbour of ICP (squared distances, lowest index on ties) and of the matching (sqrt distances) through
    every form: scalar cache (<= 512 model points), grid, and the full scan of non-finite models"""
    nm = int(10 ** rng.uniform(0, NN_LOG[0]))
    nd = int(min(10 ** rng.uniform(0, NN_LOG[0]), 2e8 / max(nm, 1)))
    nd = max(nd, 1)
    kind = int(rng.integers(0, 5))
    if kind == 0:
        model, data = rng.uniform(0, 50, (nm, 3)), rng.uniform(-10, 60, (nd, 3))
    elif kind == 1:  # lattices: thousands of exact ties
        model = rng.integers(0, 12, (nm, 3)).astype(np.float64) * 0.5
        data = rng.integers(-4, 30, (nd, 3)).astype(np.float64) * 0.25
    elif kind == 2:  # clustered model (centroids of fragments), data near it
        k = max(1, nm // 100)
        cen = rng.uniform(0, 200, (k, 3))
        model = cen[rng.integers(0, k, nm)] + rng.normal(0, 0.5, (nm, 3))
        data = model[rng.integers(0, nm, nd)] + rng.normal(0, 0.05, (nd, 3))
    elif kind == 3:  # planar / collinear
        model, data = rng.uniform(0, 50, (nm, 3)), rng.uniform(0, 50, (nd, 3))
        model[:, 2] = 3.0
        if rng.random() < 0.5:
            model[:, 1] = -1.0
    else:  # data far away from the model
        model, data = rng.uniform(0, 5, (nm, 3)), rng.uniform(0, 5, (nd, 3)) + float(rng.choice([1e2, 1e4, -1e3]))
    if rng.random() < 0.1:
        model[int(rng.integers(0, nm)), int(rng.integers(0, 3))] = float(rng.choice([np.nan, np.inf]))
    sums, nn = ctx.icp_sums(model, data)
    want = O.find_closest(model, data)
    if not np.array_equal(nn, want):
        np.savez("gpurun_out/fuzz_fail_nn.npz", model=model, data=data)
        fail("MISMATCH nn nm=%d nd=%d kind=%d" % (nm, nd, kind))
    if np.isfinite(model).all():
        M = np.eye(4)
        M[:3, 3] = rng.normal(0, 0.1, 3)
        md = float(rng.choice([0.1, 1.0, 10.0]))
        gm = ctx.match(data[:20000], model, M, md)
        om = O.match(data[:20000], model, M.reshape(16), md)
        if not (np.array_equal(gm["nearest"], om["nearest"]) and np.array_equal(gm["is_matched"], om["is_matched"])
                and np.array_equal(gm["matched_xyz"], om["matched_xyz"]) and gm["count"] == om["count"]):
            np.savez("gpurun_out/fuzz_fail_match.npz", model=model, data=data, M=M, md=md)
            fail("MISMATCH match K=%d T=%d kind=%d" % (min(nd, 20000), nm, kind))
    done["nn"] = done.get("nn", 0) + 1


def tools_case():
    """centroids (plain and ptsCount-weighted), centroid merge, and the keyed block pipeline (partition on X,Y, DBSCAN on
    motor) on random clouds"""
    n = int(10 ** rng.uniform(0.5, 5.3))
    K = int(rng.integers(1, max(2, min(n, 3000) + 1)))
    K = min(K, n)
    xyz = rng.uniform(-100, 100, (n, 3)) if rng.random() < 0.7 else rng.integers(-9, 9, (n, 3)).astype(np.float64)
    motor = rng.uniform(-10, 10, (n, 2))
    lab = rng.integers(0, K + 1, n).astype(np.int32)
    g3, g2, gc = ctx.centroids(xyz, motor, lab, K)
    o3, o2, oc = O.centroids(xyz, motor, lab, K)
    ok = np.array_equal(gc, oc)
    full = oc > 0
    ok = ok and np.allclose(g3[full], o3[full], rtol=1e-12, atol=1e-9) and np.allclose(g2[full], o2[full], rtol=1e-12, atol=1e-10)
    ok = ok and np.isnan(g3[~full]).all()
    grp = rng.integers(0, K + 1, n).astype(np.int32)  # 1..K, 0 = in no list
    grp[:K] = np.arange(1, K + 1)  # no empty list: clusList[i].li[0] throws in the C# (VCP_ERR_INDEX here, covered by the tests)
    cid = rng.integers(0, 3, n).astype(np.int32)
    pc = rng.integers(1, 6, n).astype(np.int32)
    ign = bool(rng.integers(0, 2))
    w3, wi = ctx.centroids_weighted(xyz, grp, cid, pc, K, ign)
    v3, vi = O.fixed_centroids(xyz, grp, cid, pc, K, ign)
    has = vi > 0
    ok = ok and np.array_equal(wi, vi) and np.allclose(w3[has], v3[has], rtol=1e-12, atol=1e-9)
    cen = np.round(rng.uniform(0, 30, (min(K, 2000), 2)) * 8) / 8
    ids = np.arange(1, len(cen) + 1, dtype=np.int32)
    thr = float(rng.choice([0.25, 0.5, 1.0]))
    gm, gmc = ctx.merge_centroids(cen, ids, thr)
    om, omc = O.merge_ids(cen, ids, thr)
    ok = ok and np.array_equal(gm, om) and gmc == omc
    if n >= 2:
        key = np.round(rng.uniform(0, 20, (n, 2)) * 64) / 64
        mot = np.round(rng.uniform(0, 20, (n, 2)) * 64) / 64 if rng.random() < 0.5 else key + 0.0
        eps, mp, pic = float(rng.choice([0.125, 0.3, 0.7])), int(rng.integers(1, 8)), int(rng.choice([5, 50, 400]))
        try:
            ob = O.block_pipeline(mot, eps, mp, pic, 3, key_xy=key)
        except O.OracleError:
            ob = None
        if ob is not None:
            gb = ctx.dbscan_blocks(mot, eps, mp, pic, 3, key_xy=key)  # (20 x 20 extent: never near the block-count limit)
            ok = ok and np.array_equal(gb["labels"], ob["labels"]) and np.array_equal(gb["order"], ob["order"]) \
                and gb["cluster_amount"] == ob["cluster_amount"] and gb["evals"] == ob["evals"]
    if not ok:
        fail("MISMATCH tools n=%d K=%d" % (n, K))
    done["tools"] = done.get("tools", 0) + 1


def db_case():
    """the dead v1.0 class DB (BaseClass/DB.cs: signed dx + dy, ifShown mask): the sort-and-scan form where its relation is
    provably 1-D, the pair-by-pair form otherwise (no binary grid, e < 0 / NaN, non-finite coordinates) -- labels, isClassed,
    isKeyPoint, clusterAmount, iritatorNum against the literal transcription"""
    n = int(10 ** rng.uniform(0, 3.3))
    c = cloud(n, 2)
    kind = int(rng.integers(0, 4))
    if kind == 0:
        c = np.round(c * 4.0) / 4.0
    elif kind == 1:
        c = np.round(c * 3.0) / 3.0
    span = float(np.nanmax(np.abs(np.where(np.isfinite(c), c, 0.0)))) if n else 1.0
    eps = float(rng.choice([0.0, 0.25, 1.0 / 3.0, 1.0, -0.5, np.nan, np.inf, 0.01 * span, 0.2 * span]))
    mp = int(rng.integers(-1, 9))
    shown = None if rng.random() < 0.4 else (rng.random(n) < 0.8).astype(np.uint8)
    cls = None if rng.random() < 0.5 else (rng.random(n) < 0.3).astype(np.uint8)
    lab0 = None if cls is None else (cls * rng.integers(1, 5, n)).astype(np.int32)
    o = O.db_literal(c, eps, mp, shown, cls, lab0)
    g = ctx.dbscan(c, eps, mp, N.SIGNED_SUM_2D, 0, cls, lab0, in_mask=shown)
    if not (np.array_equal(g["labels"], o["labels"]) and np.array_equal(g["is_classed"], o["classed"])
            and np.array_equal(g["is_core"], o["is_key"]) and g["cf"] == o["cluster_amount"] and g["evals"] == o["evals"]):
        np.savez("gpurun_out/fuzz_fail_db.npz", c=c, eps=eps, mp=mp, shown=np.zeros(0) if shown is None else shown,
                 cls=np.zeros(0) if cls is None else cls, lab0=np.zeros(0) if lab0 is None else lab0)
        fail("MISMATCH DB n=%d eps=%r mp=%d kind=%d" % (n, eps, mp, kind))
    done["db"] = done.get("db", 0) + 1


def run(budget=300.0, seed=12345, max_log_n=6.3, gpu_bound=None, device=0, quiet=False, nn_log=4.6, with_db=True):
    """One sweep of `budget` seconds from `seed`.  max_log_n: log10 of the largest DBSCAN cloud; gpu_bound: optional
    function n -> seconds, the GPU time a DBSCAN call on n points may take (the sweep found two cliffs that way: eps = 0
    with far outliers, a cloud inside one eps-ball).  Raises Mismatch on the first disagreement."""
    global rng, ctx
    rng = np.random.default_rng(seed)
    own = ctx is None
    if own:
        ctx = N.Context(device)
    QUIET[0] = quiet
    NN_LOG[0] = nn_log
    for k in list(done):
        done[k] = 0
    os.makedirs("gpurun_out", exist_ok=True)
    t0 = time.time()
    while time.time() - t0 < budget:
        u = rng.random()
        if u < 0.2:
            nn_case()
            continue
        if u < 0.3:
            tools_case()
            continue
        if with_db and u < 0.36:
            db_case()
            continue
        n = int(10 ** rng.uniform(0, max_log_n))
        metric = int(rng.integers(0, 3))
        dim = 3 if metric == 2 else int(rng.integers(2, 4))
        c = cloud(n, dim)
        if KIND[0] == 5 and n > 50000:  # the CPU oracle grids the full bounding box: with far outliers its cells hold the whole
            c = np.ascontiguousarray(c[:50000])  # bulk and it turns quadratic -- small clouds only for this shape
            n = len(c)
        fin = c[np.isfinite(c).all(axis=1)]
        gd = 3 if metric == 2 else 2
        # robust extent per axis (outliers and degenerate axes must not fool the density estimate: the CPU oracle is
        # quadratic in the neighbourhood size); eps so that a typical point has from none to a few dozen neighbours
        if len(fin) > 10:
            q = np.percentile(fin[:, :gd], [2, 98], axis=0)
            spans = np.maximum(q[1] - q[0], 0.0)
        else:
            spans = np.ones(gd)
        live = spans[spans > 0]
        vol = float(np.prod(live)) if len(live) else 1.0
        deff = max(len(live), 1)
        eps = float((rng.uniform(0.2, 30) * vol / max(n, 1)) ** (1.0 / deff))
        if rng.random() < 0.15 and n <= 20000:  # round thresholds (exact ties on the lattices); small clouds only: on a
            eps = float(rng.choice([0.0, 0.25, 1.0, 3.0]))  # cloud that fits inside eps the CPU oracle is quadratic
        if eps == 0.0 and n > 30000:  # the CPU oracle's grid degenerates at eps = 0 (quadratic): small clouds only
            c = np.ascontiguousarray(c[:30000])
            n = len(c)
        mp = int(rng.choice([1, 2, 3, 5, 7, 10, 16, 17, 40]))
        cf = int(rng.integers(0, 5))
        if rng.random() < 0.25:
            cls = (rng.random(n) < 0.1).astype(np.uint8)
            lab0 = (rng.integers(1, 4, n) * cls).astype(np.int32)
        else:
            cls, lab0 = None, None
        say("case n=%d dim=%d metric=%d eps=%.6g mp=%d cf=%d cls=%d kind=%d" % (n, dim, metric, eps, mp, cf, cls is not None, KIND[0]),
            end="", flush=True)
        t1 = time.time()
        g = ctx.dbscan(c, eps, mp, metric, cf, cls, lab0)
        t2 = time.time()
        say(" gpu %.3fs" % (t2 - t1), end="", flush=True)
        if gpu_bound is not None and done["dbscan"] > 0 and t2 - t1 > gpu_bound(n):  # (the first call allocates)
            fail("SLOW dbscan n=%d dim=%d metric=%d eps=%r mp=%d kind=%d: %.3f s on the GPU" % (n, dim, metric, eps, mp, KIND[0], t2 - t1))
        o = O.dbscan(c, eps, mp, metric, cf, cls, lab0)
        say(" cpu %.2fs" % (time.time() - t2), flush=True)
        ok = (np.array_equal(g["labels"], o["labels"]) and np.array_equal(g["is_classed"], o["classed"])
              and np.array_equal(g["is_core"], o["is_key"]) and g["cf"] == o["cf"] and g["evals"] == o["evals"])
        if not ok:
            np.savez("gpurun_out/fuzz_fail_dbscan.npz", c=c, eps=eps, mp=mp, metric=metric, cf=cf,
                     cls=np.zeros(0) if cls is None else cls, lab0=np.zeros(0) if lab0 is None else lab0)
            fail("MISMATCH dbscan n=%d dim=%d metric=%d eps=%r mp=%d cf=%d cls=%s" % (n, dim, metric, eps, mp, cf, cls is not None))
        done["dbscan"] += 1
        if rng.random() < 0.3 and n >= 2 and np.isfinite(c).all():
            m2 = np.ascontiguousarray(c[:, :2])
            pic = int(rng.choice([1, 3, 20, 200, 5000]))
            try:
                ob = O.block_pipeline(m2, eps, mp, pic, 3)
            except O.OracleError as e:
                try:
                    ctx.dbscan_blocks(m2, eps, mp, pic, 3)
                    fail("MISSING ERROR blocks n=%d oracle code %d" % (n, e.code))
                except N.VcpError:
                    continue
            try:
                gb = ctx.dbscan_blocks(m2, eps, mp, pic, 3)
            except N.VcpError as e:
                if e.code == -5:  # more than 2^26 - 4 blocks: the library's documented limit (the C# would need a 4 GB array)
                    continue
                raise
            okb = (np.array_equal(gb["labels"], ob["labels"]) and np.array_equal(gb["order"], ob["order"])
                   and np.array_equal(gb["block_of"], ob["block_of"]) and gb["kept"] == ob["kept"]
                   and gb["cluster_amount"] == ob["cluster_amount"] and gb["evals"] == ob["evals"])
            if not okb:
                np.savez("gpurun_out/fuzz_fail_blocks.npz", m=m2, eps=eps, mp=mp, pic=pic)
                fail("MISMATCH blocks n=%d eps=%r mp=%d pic=%d" % (n, eps, mp, pic))
            done["blocks"] += 1
        if (done["dbscan"] % 50) == 0:
            say("%.0f s: %d dbscan, %d block pipelines agree" % (time.time() - t0, done["dbscan"], done["blocks"]), flush=True)
    msg = ("OK: %d dbscan calls, %d block pipelines, %d nearest-neighbour / matching cases, %d calls of the dead class DB "
           "bit-exact and %d centroid / merge / keyed-pipeline cases against the oracle (seed %d)"
           % (done["dbscan"], done["blocks"], done.get("nn", 0), done.get("db", 0), done.get("tools", 0), seed))
    print(msg, flush=True)
    if own:
        ctx.close()
        ctx = None
    return dict(done)


if __name__ == "__main__":
    try:
        run(float(sys.argv[1]) if len(sys.argv) > 1 else 300.0, int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
    except Mismatch:
        sys.exit(1)

