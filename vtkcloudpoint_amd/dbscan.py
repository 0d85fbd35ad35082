"""Drop-in mirrors of the reference's clustering classes over the C-ABI.

DBImproved: BaseClass/DBImproved.cs:8-116 (same public fields, same dbscan signature, same in-place mutation
of the caller's Point3D objects).  DB: BaseClass/DB.cs (dead v1.0 class, FrmMain.cs:38) -- surface only.
"""
import numpy as np

from . import _native
from .datamodel import motor_array, xyz_array
from .runtime import default_context


class NotSupportedError(RuntimeError):
    pass


class DBImproved:
    iritatorNum = 0  # public static int iritatorNum (DBImproved.cs:12): distance evaluations so far

    # which coordinates / metric getDisP uses; the shipped C# is L1 on (motor_x, motor_y) (DBImproved.cs:16-21),
    # the Euclidean forms are its commented-out alternatives (:20, :24)
    metric = _native.L1_2D

    def __init__(self, ctx=None):
        self.clusterAmount = 0  # DBImproved.cs:10
        self.pointsAmount = 0   # :11
        self.cf = 0             # :13
        self._ctx = ctx

    @staticmethod
    def getDisP(p1, p2):
        """DBImproved.cs:14-25 (host-side, one pair; the bulk path runs on the GPU)."""
        dx = p1.motor_x - p2.motor_x
        dy = p1.motor_y - p2.motor_y
        DBImproved.iritatorNum += 1
        return abs(dx) + abs(dy)

    def _coords(self, lst):
        if self.metric == _native.L2_3D:
            return xyz_array(lst)
        return motor_array(lst)

    def dbscan(self, lst, e, minPts):
        """DBImproved.cs:91-114: mutates clusterId / isClassed / isKeyPoint of the points in `lst`."""
        n = len(lst)
        if n == 0:  # the C# loop body never runs
            self.clusterAmount = self.cf
            return
        ctx = self._ctx or default_context()
        classed = np.fromiter((1 if p.isClassed else 0 for p in lst), np.uint8, n)
        any_classed = bool(classed.any())
        labels = np.fromiter((p.clusterId for p in lst), np.int32, n) if any_classed else None
        r = ctx.dbscan(self._coords(lst), float(e), int(minPts), self.metric, int(self.cf),
                       classed if any_classed else None, labels)
        lab, core, cls = r["labels"], r["is_core"], r["is_classed"]
        for i, p in enumerate(lst):
            if any_classed:
                p.clusterId = int(lab[i])
            elif lab[i] != 0:
                p.clusterId = int(lab[i])  # untouched points keep whatever clusterId the caller left
            if cls[i]:
                p.isClassed = True
            if core[i]:
                p.isKeyPoint = True
        self.pointsAmount += n                      # :99
        self.cf = r["cf"]
        self.clusterAmount = self.cf                # :112
        DBImproved.iritatorNum += r["evals"]


class DB:
    """BaseClass/DB.cs:9-116, the v1.0 class (dead in the reference: its only use is commented out at FrmMain.cs:38).
    Signed metric dx + dy on (X, Y) (:21), ifShown filter (:40,:63,:98), cluster ids from 1.  dbscan runs on the GPU
    (csrc/dbdead.hip, the 1-D structure of the signed relation); isKeyPoint / expandCluster stay host-side statics
    with the C#'s own control flow, as the class exposes them."""
    iritatorNum = 0

    def __init__(self, ctx=None):
        self.clusterAmount = 0
        self.pointsAmount = 0
        self._ctx = ctx

    @staticmethod
    def getDisP(p1, p2):
        dx = p1.X - p2.X
        dy = p1.Y - p2.Y
        DB.iritatorNum += 1
        return dx + dy  # DB.cs:21

    @staticmethod
    def isKeyPoint(lst, p, e, minPts):
        """DB.cs:33-55 (host-side: O(n) per call, for source compatibility only)."""
        tmp = [i for i, p2 in enumerate(lst) if p2.ifShown and DB.getDisP(p, p2) <= e]
        if len(tmp) >= minPts:
            p.isKeyPoint = True
        return tmp

    @staticmethod
    def expandCluster(p, nei, c, e, minPts, lst):
        """DB.cs:57-91, statement by statement (host-side; the never-matching boxed-reference dedupe scan :72-84 is
        dropped: it only costs time)."""
        p.clusterId = c
        t = 0
        while t < len(nei):
            dpp = lst[nei[t]]
            t += 1
            if not dpp.ifShown:
                continue
            if not dpp.isClassed:
                dpp.isClassed = True
                tmp = DB.isKeyPoint(lst, dpp, e, minPts)
                if len(tmp) >= minPts:
                    nei.extend(tmp)
            dpp.clusterId = c

    def dbscan(self, lst, e, minPts):
        """DB.cs:92-115: mutates clusterId / isClassed / isKeyPoint of the shown points of `lst`."""
        n = len(lst)
        if n == 0:
            self.clusterAmount = 0
            return
        ctx = self._ctx or default_context()
        shown = np.fromiter((1 if p.ifShown else 0 for p in lst), np.uint8, n)
        classed = np.fromiter((1 if p.isClassed else 0 for p in lst), np.uint8, n)
        labels = np.fromiter((p.clusterId for p in lst), np.int32, n)
        r = ctx.dbscan(xyz_array(lst)[:, :2].copy(), float(e), int(minPts), _native.SIGNED_SUM_2D, 0, classed, labels,
                       in_mask=shown)
        lab, core, cls = r["labels"], r["is_core"], r["is_classed"]
        for i, p in enumerate(lst):
            p.clusterId = int(lab[i])
            if cls[i]:
                p.isClassed = True
            if core[i]:
                p.isKeyPoint = True
        self.pointsAmount += int(shown.sum())       # :98-101
        self.clusterAmount = r["cf"]                # :113
        DB.iritatorNum += r["evals"]
