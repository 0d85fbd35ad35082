"""A SECOND, independent transcription of the reference's DBImproved (BaseClass/DBImproved.cs:14-114) -- pure Python,
written from the C# text without looking at oracle/vcp_oracle.cpp -- against the oracle's literal C++ transcription.
The reference cannot be compiled or run here (parity unpinned by the reference); two transcriptions
made independently that agree on every output, including the iritatorNum count and the growth of the neighbour list
with duplicates (the boxed-reference comparison at :76 is never true), is the strongest pin available."""
import numpy as np
import pytest


class P:  # Point3D: the members the path touches (BaseClass/DataModel.cs:102-160)
    __slots__ = ("motor_x", "motor_y", "clusterId", "isClassed", "isKeyPoint", "index")

    def __init__(self, x, y, cid=0, classed=False):
        self.motor_x, self.motor_y, self.clusterId, self.isClassed, self.isKeyPoint = x, y, cid, classed, False


class DBImprovedPy:
    iritatorNum = 0  # static

    def __init__(self, cf=0):
        self.clusterAmount = 0
        self.pointsAmount = 0
        self.cf = cf

    @staticmethod
    def getDisP(p1, p2):  # :14-25
        dx = p1.motor_x - p2.motor_x
        dy = p1.motor_y - p2.motor_y
        DBImprovedPy.iritatorNum += 1
        return abs(dx) + abs(dy)

    @staticmethod
    def isKeyPoint(lst, p, e, minPts):  # :33-54
        count = 0
        tmp = []
        for i in range(len(lst)):
            if DBImprovedPy.getDisP(p, lst[i]) <= e:
                count += 1
                tmp.append([i])  # a BOXED int: a fresh object per Add
        if count >= minPts:
            p.isKeyPoint = True
        return tmp

    @staticmethod
    def expandCluster(p, nei, c, e, minPts, lst):  # :56-90
        p.clusterId = c
        i = 0
        while i < len(nei):  # nei.Count is re-read every trip
            dpp = lst[nei[i][0]]
            if not dpp.isClassed:
                dpp.isClassed = True
                tmp = DBImprovedPy.isKeyPoint(lst, dpp, e, minPts)
                if len(tmp) >= minPts:
                    for k in range(len(tmp)):
                        flag = False
                        for j in range(len(nei)):
                            if nei[j] is tmp[k]:  # object == object on boxed ints: reference equality
                                flag = True
                                break
                        if not flag:
                            nei.append(tmp[k])
            dpp.clusterId = c
            i += 1

    def dbscan(self, lst, e, minPts):  # :91-114
        for i in range(len(lst)):
            dpp = lst[i]
            self.pointsAmount += 1
            if dpp.isClassed:
                continue
            tmp = DBImprovedPy.isKeyPoint(lst, dpp, e, minPts)
            if len(tmp) >= minPts:
                self.cf += 1
                DBImprovedPy.expandCluster(dpp, tmp, self.cf, e, minPts, lst)
        self.clusterAmount = self.cf


def run_py(c, eps, mp, cf_in, cls, lab0):
    pts = [P(float(c[i, 0]), float(c[i, 1]), 0 if lab0 is None else int(lab0[i]), False if cls is None else bool(cls[i]))
           for i in range(len(c))]
    DBImprovedPy.iritatorNum = 0
    db = DBImprovedPy(cf_in)
    db.dbscan(pts, eps, mp)
    return dict(labels=np.array([p.clusterId for p in pts], np.int32), classed=np.array([p.isClassed for p in pts], np.uint8),
                is_key=np.array([p.isKeyPoint for p in pts], np.uint8), cf=db.clusterAmount,
                evals=DBImprovedPy.iritatorNum, points=db.pointsAmount)


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_two_transcriptions_agree(oracle, seed):
    rng = np.random.default_rng(1000 + seed)
    for trial in range(120):
        n = int(rng.integers(0, 45))
        if trial % 4 == 0:
            c = rng.integers(0, 7, size=(n, 2)).astype(np.float64) * 0.5      # lattice: exact d == eps ties, duplicates
        else:
            c = np.round(rng.normal(0, 1.2, size=(n, 2)) * 8) / 8
        if trial % 11 == 0 and n:
            c[rng.integers(0, n)] = np.nan                                       # a point that is not its own neighbour
        eps = float(rng.choice([0.5, 1.0, 1.5, 0.0, 0.25]))
        mp = int(rng.integers(0, 6))
        cf = int(rng.integers(0, 4))
        if trial % 3 == 0 and n:
            cls = (rng.random(n) < 0.2).astype(np.uint8)
            lab0 = (rng.integers(1, 5, n) * cls).astype(np.int32)
        else:
            cls, lab0 = None, None
        a = run_py(c, eps, mp, cf, cls, lab0)
        o = oracle.dbscan(c, eps, mp, oracle.L1_2D, cf, cls, lab0, literal=True, dedupe=bool(trial % 2))
        tag = "seed %d trial %d" % (seed, trial)
        assert np.array_equal(a["labels"], o["labels"]), tag
        assert np.array_equal(a["classed"], o["classed"]), tag
        assert np.array_equal(a["is_key"], o["is_key"]), tag
        assert a["cf"] == o["cf"] and a["evals"] == o["evals"], tag


# ---- the block pipeline: getClusterFromMotor (FrmMain.cs:1214-1291), DoWork3 (:1340-1361), StartCode (:2782-2794),
# CompleteWork3 (:1442-1520)
class OutOfRange(Exception):
    pass


def block_pipeline_py(motor, eps, min_pts, pts_in_cell):
    """Second transcription, with the declared deviations of DESIGN.md section 2 applied (stable sorts; a block-0 point is
    not filed under a rectangle as well; clusterSum summed in block order)."""
    n = len(motor)
    raw = [P(float(motor[i, 0]), float(motor[i, 1])) for i in range(n)]
    for i, p in enumerate(raw):
        p.index = i
    x_Min = min(p.motor_x for p in raw)
    y_Min = min(p.motor_y for p in raw)
    x_Max = max(p.motor_x for p in raw)
    y_Max = max(p.motor_y for p in raw)
    raw.sort(key=lambda p: max(p.motor_x - x_Min, p.motor_y - y_Min))  # :1229-1251 (stable)
    cell = raw[:pts_in_cell]
    cell_x = max(p.motor_x for p in cell) - x_Min
    cell_y = max(p.motor_y for p in cell) - y_Min
    with np.errstate(divide="ignore", invalid="ignore"):
        fr = np.float64(y_Max - y_Min) / np.float64(cell_y)
        fc = np.float64(x_Max - x_Min) / np.float64(cell_x)
    if not (np.isfinite(fr) and np.isfinite(fc)):
        raise ZeroDivisionError
    rows, cols = int(fr) + 1, int(fc) + 1
    first = set(id(p) for p in cell)

    def by_scale(min_x, min_y, max_x, max_y):  # Tools.getListByScale2, Tools.cs:510-513
        return [p for p in raw if id(p) not in first and p.motor_x > min_x and p.motor_y > min_y
                and p.motor_x <= max_x and p.motor_y <= max_y]

    cells = [None] * (rows * cols)
    cells[0] = cell
    index = 0
    for p in range(rows):
        for q in range(cols):
            if index == 0:
                index += 1
                continue
            lo_x, lo_y = x_Min + q * cell_x, y_Min + p * cell_y
            if p == rows - 1 and q != cols - 1:
                cells[index] = by_scale(lo_x, lo_y, x_Min + (q + 1) * cell_x, y_Max)
            elif p != rows - 1 and q == cols - 1:
                cells[index] = by_scale(lo_x, lo_y, x_Max, y_Min + (p + 1) * cell_y)
            elif p == rows - 1 and q == cols - 1:
                cells[index] = by_scale(lo_x, lo_y, x_Max, y_Max)
            else:
                cells[index] = by_scale(lo_x, lo_y, x_Min + (q + 1) * cell_x, y_Min + (p + 1) * cell_y)
            index += 1
    block_of = np.full(n, -1, np.int32)
    for b, c in enumerate(cells):
        for p in c:
            block_of[p.index] = b
    # DoWork3 (:1340-1361: clusterSum = 1 before the pool threads add to it), StartCode per cell
    DBImprovedPy.iritatorNum = 0
    clusterSum = 1
    for c in cells:
        db = DBImprovedPy()
        db.dbscan(c, eps, min_pts)
        clusterSum += db.clusterAmount
    # CompleteWork3
    idNow, clusLen, delSum = 0, 0, 0
    merge = []
    for c in cells:
        if len(c) == 0:
            continue
        c.sort(key=lambda p: p.clusterId)  # :1449 (stable)
        idLast = c[0].clusterId
        if idLast != 0:
            idNow += 1
            clusLen = 1
        else:
            clusLen = 0
        for pt in c:
            cid = pt.clusterId
            if cid == 0:
                merge.append(pt)
            else:
                if cid != idLast:
                    if clusLen <= 3 and idLast != 0:
                        delSum += 1
                        for k in range(clusLen):
                            at = len(merge) - 1 - k
                            if at < 0:
                                raise OutOfRange()  # List<T> indexer: ArgumentOutOfRangeException
                            merge[at].clusterId = 0
                    else:
                        idNow += 1
                    clusLen = 1
                else:
                    clusLen += 1
                pt.clusterId = idNow
                merge.append(pt)
                idLast = cid
    dbb = DBImprovedPy()
    dbb.clusterAmount = clusterSum - delSum
    dbb.cf = clusterSum - delSum - 1
    kept = dbb.cf
    zero = [p for p in merge if p.clusterId == 0]
    merge = [p for p in merge if p.clusterId != 0]
    for p in zero:
        p.isClassed = False
    dbb.dbscan(zero, eps, min_pts)
    merge += zero
    labels = np.zeros(n, np.int32)
    for p in merge:
        labels[p.index] = p.clusterId
    return dict(labels=labels, block_of=block_of, order=np.array([p.index for p in merge], np.int64), rows=rows, cols=cols,
                kept=kept, del_sum=delSum, cluster_amount=dbb.clusterAmount, evals=DBImprovedPy.iritatorNum)


def test_block_pipeline_two_transcriptions_agree(oracle):
    rng = np.random.default_rng(77)
    done = thrown = 0
    for trial in range(150):
        n = int(rng.integers(1, 140))
        if trial % 3 == 0:
            m = rng.integers(0, 9, size=(n, 2)).astype(np.float64) * 0.5
        else:
            m = np.round(np.concatenate([rng.normal(0, 0.4, (n // 2, 2)), rng.uniform(-3, 3, (n - n // 2, 2))]) * 16) / 16
        eps = float(rng.choice([0.25, 0.5, 1.0]))
        mp = int(rng.integers(1, 5))
        pic = int(rng.integers(1, 25))
        try:
            a = block_pipeline_py(m, eps, mp, pic)
        except (ZeroDivisionError, OutOfRange):
            with pytest.raises(Exception):
                oracle.block_pipeline(m, eps, mp, pic, 3, canonical=False, brute=True)
            thrown += 1
            continue
        o = oracle.block_pipeline(m, eps, mp, pic, 3, canonical=False, brute=True)
        tag = "trial %d" % trial
        for k in ("rows", "cols", "kept", "del_sum", "cluster_amount", "evals"):
            assert a[k] == o[k], (tag, k, a[k], o[k])
        assert np.array_equal(a["block_of"], o["block_of"]), tag
        assert np.array_equal(a["labels"], o["labels"]), tag
        assert np.array_equal(a["order"], o["order"]), tag
        done += 1
    assert done > 100 and thrown > 0


# ---- ICP.go_hell_ICP (BaseClass/ICP.cs:18-181): loop structure, composition order and stop rule as written; the Horn
# solve in its INTENDED arithmetic through numpy's symmetric eigen solver instead of Jacobi sweeps
def icp_py(model, data, e, max_round=200):
    R = np.zeros((3, 3))  # the caller's matrices: untouched when round 1 already stops
    T = np.zeros(3)
    pre_d = d = 0.0
    rnd = 0
    P = data.copy()
    while True:
        pre_d = d
        # FindClosestPointSet :224-250: strict `<`, seeded with model[0] -> lowest index among equal distances
        dd = ((P[:, None, :] - model[None, :, :]) ** 2)
        dist = dd[:, :, 0] + dd[:, :, 1] + dd[:, :, 2]
        Y = model[np.argmin(dist, axis=1)]
        muP, muY = P.mean(axis=0), Y.mean(axis=0)
        cov = (P.T @ Y) / len(P) - np.outer(muP, muY)
        A = cov - cov.T
        delta = np.array([A[1, 2], A[2, 0], A[0, 1]])
        tr = np.trace(cov)
        Q = np.zeros((4, 4))
        Q[0, 0] = tr
        Q[0, 1:] = delta
        Q[1:, 0] = delta
        Q[1:, 1:] = cov + cov.T - tr * np.eye(3)
        w, V = np.linalg.eigh(Q)
        q = V[:, np.argmax(w)]
        R1 = np.array([  # CalculateRotation :274-285
            [q[0] * q[0] + q[1] * q[1] - q[2] * q[2] - q[3] * q[3], 2.0 * (q[1] * q[2] - q[0] * q[3]), 2.0 * (q[1] * q[3] + q[0] * q[2])],
            [2.0 * (q[1] * q[2] + q[0] * q[3]), q[0] * q[0] - q[1] * q[1] + q[2] * q[2] - q[3] * q[3], 2.0 * (q[2] * q[3] - q[0] * q[1])],
            [2.0 * (q[1] * q[3] - q[0] * q[2]), 2.0 * (q[2] * q[3] + q[0] * q[1]), q[0] * q[0] - q[1] * q[1] - q[2] * q[2] + q[3] * q[3]]])
        T1 = muY - R1 @ muP
        d = float((((P - Y) ** 2).sum(axis=1)).sum())
        rnd += 1
        if abs(d - pre_d) >= e:
            if rnd == 1:
                R, T = R1.copy(), T1.copy()
            else:
                R, T = R1 @ R, R1 @ T + T1
            P = data @ R.T + T  # TransPoint :195-219, always from the original data
        if not abs(d - pre_d) >= e or rnd >= max_round:
            return dict(R=R, T=T, sse=d, iters=rnd)


def test_icp_loop_two_transcriptions_agree(oracle):
    rng = np.random.default_rng(5)
    for trial in range(40):
        nm = int(rng.integers(4, 40))
        nd = int(rng.integers(10, 300))
        model = rng.uniform(0, 20, (nm, 3))
        base = model[rng.integers(0, nm, nd)] + rng.normal(0, 0.05, (nd, 3))
        ang = np.deg2rad(rng.uniform(0, 6))
        ax = rng.normal(size=3)
        ax /= np.linalg.norm(ax)
        K = np.array([[0, -ax[2], ax[1]], [ax[2], 0, -ax[0]], [-ax[1], ax[0], 0]])
        Rt = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * (K @ K)
        data = base @ Rt.T + rng.uniform(-0.3, 0.3, 3)
        tol = float(rng.choice([1e-3, 1e-6, 1e-9]))
        a = icp_py(model, data, tol)
        o = oracle.icp(model, data, tol, 200, oracle.STOP_SSE_DELTA)
        assert a["iters"] == o["iters"], trial
        assert np.allclose(a["R"], o["R"], atol=1e-9) and np.allclose(a["T"], o["T"], atol=1e-8), trial
        assert abs(a["sse"] - o["sse"]) <= 1e-9 * max(1.0, abs(o["sse"])), trial


# ---- calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618, getDisP :829-835), GetClusList (Tools.cs:162-195)
def test_matching_and_centroids_python_loops(oracle):
    import math
    rng = np.random.default_rng(31)
    for trial in range(25):
        K, T = int(rng.integers(1, 40)), int(rng.integers(1, 40))
        cen = rng.integers(0, 8, (K, 3)).astype(np.float64) * 0.5 if trial % 2 else rng.uniform(0, 5, (K, 3))
        tru = rng.integers(0, 8, (T, 3)).astype(np.float64) * 0.5 if trial % 2 else rng.uniform(0, 5, (T, 3))
        M = np.eye(4)
        M[:3, :3] += rng.normal(0, 0.05, (3, 3))
        M[:3, 3] = rng.normal(0, 0.2, 3)
        md = float(rng.choice([0.3, 1.0, 2.5]))
        mx = np.zeros((K, 3))
        is_m = np.zeros(K, np.uint8)
        near = np.zeros(K, np.int32)
        cnt = 0
        for j in range(K):
            for r in range(3):  # :3575-3583, left to right
                mx[j, r] = cen[j, 0] * M[r, 0] + cen[j, 1] * M[r, 1] + cen[j, 2] * M[r, 2] + M[r, 3]

            def dis(i):
                dx, dy, dz = tru[i, 0] - mx[j, 0], tru[i, 1] - mx[j, 1], tru[i, 2] - mx[j, 2]
                return math.sqrt(dx * dx + dy * dy + dz * dz)
            best, c2t = 0, dis(0)
            for i in range(T):
                ddd = dis(i)
                if ddd < c2t:
                    c2t, best = ddd, i
            near[j] = best
            if c2t < md:
                is_m[j] = 1
                cnt += 1
        o = oracle.match(cen, tru, M.reshape(16), md)
        assert np.array_equal(o["matched_xyz"], mx) and np.array_equal(o["nearest"], near)
        assert np.array_equal(o["is_matched"], is_m) and o["count"] == cnt
    for trial in range(25):  # GetClusList: LINQ Average = sequential binary64 sum / count, in list order
        n, K = int(rng.integers(1, 200)), int(rng.integers(1, 9))
        xyz = rng.uniform(-50, 50, (n, 3))
        motor = rng.uniform(-5, 5, (n, 2))
        lab = rng.integers(0, K + 1, n).astype(np.int32)
        c3, c2, counts = oracle.centroids(xyz, motor, lab, K)
        for k in range(1, K + 1):
            li = [i for i in range(n) if lab[i] == k]
            assert counts[k - 1] == len(li)
            if not li:
                continue
            for a in range(3):
                sm = 0.0
                for i in li:
                    sm += xyz[i, a]
                assert c3[k - 1, a] == sm / len(li)
            for a in range(2):
                sm = 0.0
                for i in li:
                    sm += motor[i, a]
                assert c2[k - 1, a] == sm / len(li)


# ---- Tools.MergeIDByDistance (Tools.cs:580-621) + refreshCensAndClusByDictionary (:521-572)
def test_merge_and_refresh_python_transcription(oracle):
    rng = np.random.default_rng(41)
    for trial in range(40):
        K = int(rng.integers(1, 25))
        n = int(rng.integers(K, 300))
        lab = rng.integers(0, K + 1, n).astype(np.int32)
        lab[:K] = np.arange(1, K + 1)  # every cluster has a member (clusList position == id - 1, as the C# assumes)
        xyz = np.round(rng.uniform(0, 6, (n, 3)) * 4) / 4
        motor = rng.uniform(-5, 5, (n, 2))
        thr = float(rng.choice([0.25, 0.6, 1.5]))
        # centroids as GetClusList leaves them (sequential sums)
        cen = []
        for k in range(1, K + 1):
            li = [i for i in range(n) if lab[i] == k]
            c = [0.0, 0.0, 0.0]
            for a in range(3):
                sm = 0.0
                for i in li:
                    sm += xyz[i, a]
                c[a] = sm / len(li)
            cen.append(c)
        cen = np.array(cen)
        # MergeIDByDistance
        pts = [P(cen[k, 0], cen[k, 1]) for k in range(K)]   # motor_x = X, motor_y = Y, clusterId = 0
        before = list(range(1, K + 1))                        # IDBeforeMerge
        DBImprovedPy().dbscan(pts, thr, 2)
        dick, seen = {}, set()
        for a, p in enumerate(pts):
            if p.clusterId != 0:
                if before[a] not in seen:
                    seen.add(before[a])
                    for b, q in enumerate(pts):
                        if q.clusterId == p.clusterId and before[b] != before[a]:
                            seen.add(before[b])
                            dick[before[b]] = before[a]
            else:
                seen.add(before[a])
        map_to, mc = oracle.merge_ids(cen[:, :2], np.arange(1, K + 1, dtype=np.int32), thr)
        assert mc == len(dick)
        assert np.array_equal(map_to, np.array([dick.get(k, 0) for k in range(1, K + 1)], np.int32)), trial
        # refreshCensAndClusByDictionary: lists per id in rawData order, merged lists appended, removed, sorted, renumbered
        lists = {k: [i for i in range(n) if lab[i] == k] for k in range(1, K + 1)}
        order_ids = list(range(1, K + 1))
        for k in order_ids:
            if k in dick:
                lists[dick[k]].extend(lists[k])
        kept = sorted(k for k in order_ids if k not in dick)
        newlab = np.zeros(n, np.int32)
        c3, c2 = [], []
        for new_id, k in enumerate(kept, start=1):
            for i in lists[k]:
                newlab[i] = new_id
            row3, row2 = [], []
            for a in range(3):
                sm = 0.0
                for i in lists[k]:
                    sm += xyz[i, a]
                row3.append(sm / len(lists[k]))
            for a in range(2):
                sm = 0.0
                for i in lists[k]:
                    sm += motor[i, a]
                row2.append(sm / len(lists[k]))
            c3.append(row3)
            c2.append(row2)
        lo, ko, o3, o2, oc = oracle.refresh_by_dictionary(xyz, motor, lab, K, map_to)
        assert ko == len(kept) and np.array_equal(lo, newlab), trial
        assert np.array_equal(o3, np.array(c3).reshape(-1, 3)) and np.array_equal(o2, np.array(c2).reshape(-1, 2)), trial
