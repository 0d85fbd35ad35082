"""Drop-in mirrors of the Tools statics on the path (BaseClass/Tools.cs) and of the MainForm methods that
drive it (FrmMain.cs), over the C-ABI.  Same names, argument meaning and in-place mutation as the C#."""
import numpy as np

from .datamodel import ClusObj, Point3D, motor_array, xyz_array
from .runtime import default_context


class Tools:
    @staticmethod
    def getListByScale(rawData, min_x, min_y, max_x, max_y):
        """Tools.cs:507-509 (host-side filter; the bulk partition runs inside vcp_dbscan_blocks)."""
        return [p for p in rawData if p.X > min_x and p.Y > min_y and p.X <= max_x and p.Y <= max_y]

    @staticmethod
    def getListByScale2(rawData, min_x, min_y, max_x, max_y):
        """Tools.cs:510-513."""
        return [p for p in rawData
                if p.motor_x > min_x and p.motor_y > min_y and p.motor_x <= max_x and p.motor_y <= max_y]

    @staticmethod
    def GetClusList(rawData, centers, centers2D, clusList, idList=None, ctx=None):
        """Tools.cs:162-195: bucket points by clusterId-1 into clusList[*].li; append one 3-D and one 2-D
        centroid per non-empty cluster (means computed on the GPU)."""
        ctx = ctx or default_context()
        K = len(clusList)
        labels = np.fromiter((p.clusterId for p in rawData), np.int32, len(rawData))
        for p in rawData:
            if p.clusterId != 0:
                clusList[p.clusterId - 1].li.append(p)  # IndexError like the C# when the id is out of range
        if K == 0 or len(rawData) == 0:
            return
        c3, c2, counts = ctx.centroids(xyz_array(rawData), motor_array(rawData), labels, K)
        for k, obj in enumerate(clusList):
            if counts[k] == 0:
                continue  # :191
            centers.append(Point3D(c3[k, 0], c3[k, 1], c3[k, 2], obj.clusId, True))
            centers2D.append(Point3D(c2[k, 0], c2[k, 1], 0, obj.clusId, True))

    @staticmethod
    def getClusterCenter(clus, rawData, centers, clusList, idList=None, ctx=None):
        """Tools.cs:118-155: remap ids through idList (or identity), bucket, centroids of X,Y,Z."""
        ctx = ctx or default_context()
        idTmp = {j: i + 1 for i, j in enumerate(idList)} if idList is not None else {t + 1: t + 1 for t in range(clus)}
        for p in rawData:
            if p.clusterId != 0:
                p.clusterId = idTmp[p.clusterId]  # KeyError like the C# dictionary
                clusList[p.clusterId - 1].li.append(p)
        K = len(clusList)
        if K == 0 or len(rawData) == 0:
            return
        labels = np.fromiter((p.clusterId for p in rawData), np.int32, len(rawData))
        c3, _, counts = ctx.centroids(xyz_array(rawData), None, labels, K)
        for k, obj in enumerate(clusList):
            obj.clusId = k + 1
            if counts[k] == 0:
                continue
            obj.clusId = obj.li[0].clusterId
            centers.append(Point3D(c3[k, 0], c3[k, 1], c3[k, 2], obj.clusId, True))

    @staticmethod
    def MergeIDByDistance(centers, thre, ctx=None):
        """Tools.cs:580-621: returns {merged id -> id it is merged into}; mutates the centers like the C#
        (IDBeforeMerge, motor_x/motor_y := X/Y, clusterId := cluster of the centroid DBSCAN)."""
        ctx = ctx or default_context()
        for p in centers:
            p.IDBeforeMerge = p.clusterId
            p.motor_x, p.motor_y = p.X, p.Y
            p.clusterId = 0
        if not centers:
            return {}
        ids = np.fromiter((p.IDBeforeMerge for p in centers), np.int32, len(centers))
        cxy = motor_array(centers)
        # the centroid DBSCAN itself (labels are part of the C#'s visible side effects)
        r = ctx.dbscan(cxy, float(thre), 2)
        for p, l, c in zip(centers, r["labels"], r["is_classed"]):
            p.clusterId = int(l)
            p.isClassed = bool(c) or p.isClassed
        map_to, _ = ctx.merge_centroids(cxy, ids, float(thre))
        return {int(ids[k]): int(map_to[k]) for k in range(len(centers)) if map_to[k] != 0}

    @staticmethod
    def refreshCensAndClusByDictionary(dic, clusList, centers, centers2D, ctx=None):
        """Tools.cs:521-572.  `centers` / `centers2D` are `ref` lists in the C#: they are extended in place.
        clusList is rebuilt in place (merged entries removed, ids renumbered 1..K')."""
        ctx = ctx or default_context()
        K = len(clusList)
        map_by_id = np.zeros(K, np.int32)
        for ob in clusList:
            if ob.clusId in dic:
                map_by_id[ob.clusId - 1] = dic[ob.clusId]
        # move the points (list order: the target's own points, then merged clusters in clusList order)
        for ob in list(clusList):
            if ob.clusId in dic:
                clusList[dic[ob.clusId] - 1].li.extend(ob.li)
        pts = [p for ob in clusList if ob.clusId not in dic for p in ob.li]
        old = np.fromiter((ob.clusId for ob in clusList if ob.clusId not in dic for _ in ob.li), np.int32, len(pts))
        labels, nk, c3, c2, counts = ctx.refresh_by_dictionary(xyz_array(pts), motor_array(pts), old, K, map_by_id)
        clusList[:] = sorted((ob for ob in clusList if ob.clusId not in dic), key=lambda o: o.clusId)
        for new_id, ob in enumerate(clusList, 1):
            ob.clusId = new_id
            for p in ob.li:
                p.clusterId = new_id
        for k, ob in enumerate(clusList):
            centers.append(Point3D(c3[k, 0], c3[k, 1], c3[k, 2], ob.clusId, True))
            centers2D.append(Point3D(c2[k, 0], c2[k, 1], 0, ob.clusId, True))


class Point2D:
    """BaseClass/DataModel.cs:66-98."""

    def __init__(self, xx, yy, cluID=0):
        self.x, self.y, self.clusID = float(xx), float(yy), int(cluID)
        self.isFilter = False
        self.radius = 0.0


def getCircles(clusList, is3D, ctx=None):
    """Tools.getCircles (BaseClass/Tools.cs:394-409): one Point2D (centre, radius, clusID = position+1) per
    cluster with more than 3 points; Geometry.FindMinimalBoundingCircle runs on the GPU (vcp_mcc)."""
    ctx = ctx or default_context()
    K = len(clusList)
    pts, lab = [], []
    for j, ob in enumerate(clusList):
        for p in ob.li:
            pts.append((p.X, p.Y) if is3D else (p.motor_x, p.motor_y))
            lab.append(j + 1)
    if K == 0 or not pts:
        return []
    r = ctx.mcc(np.array(pts, np.float64), np.array(lab, np.int32), K)
    circles = []
    for j in range(K):
        if not r["valid"][j]:
            continue
        c = Point2D(r["centers"][j, 0], r["centers"][j, 1], j + 1)
        c.radius = float(r["radius"][j])
        circles.append(c)
    return circles


Tools.getCircles = staticmethod(getCircles)


class ClusterPipeline:
    """The MainForm state and methods on the path: getClusterFromMotor + DoWork3/StartCode + CompleteWork3
    (FrmMain.cs:1214-1291, :1340-1361, :2782-2794, :1432-1544) as one blocking call."""

    def __init__(self, rawData, ctx=None):
        self.rawData = rawData
        self.ctx = ctx or default_context()
        self.clusForMerge = []
        self.centers, self.centers2D, self.clusList = [], [], []
        self.clusterSum = 0
        self.info = None

    def getClusterFromMotor(self, tr, pts, ptsInCell, small_max=3):
        raw = self.rawData
        r = self.ctx.dbscan_blocks(motor_array(raw), float(tr), int(pts), int(ptsInCell), small_max)
        self.info = r
        for p, l in zip(raw, r["labels"]):
            p.clusterId = int(l)
            p.isClassed = l != 0
        self.clusForMerge = [raw[int(i)] for i in r["order"]]
        self.clusList = []
        for j in range(r["cluster_amount"]):  # FrmMain.cs:1527-1532
            obj = ClusObj()
            obj.clusId = j + 1
            self.clusList.append(obj)
        self.centers, self.centers2D = [], []
        Tools.GetClusList(self.clusForMerge, self.centers, self.centers2D, self.clusList, None, self.ctx)
        self.clusterSum = r["cluster_amount"]  # :1538
        return r


class Matcher:
    """calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618)."""

    def __init__(self, centers, truths, M, ctx=None):
        self.centers = centers
        self.truths = np.asarray(truths, np.float64).reshape(-1, 3)
        self.M = np.asarray(M, np.float64).reshape(4, 4)
        self.ctx = ctx or default_context()
        self.matchedID = []

    def RecorrectMatchingPtsByDistance(self, matchDistance):
        c = np.array([(p.tmp_X, p.tmp_Y, p.tmp_Z) for p in self.centers], np.float64).reshape(-1, 3)
        r = self.ctx.match(c, self.truths, self.M, float(matchDistance))
        self.matchedID = []
        for j, p in enumerate(self.centers):
            p.matched_X, p.matched_Y, p.matched_Z = (float(v) for v in r["matched_xyz"][j])
            p.isMatched = bool(r["is_matched"][j])
            if p.isMatched:
                p.matchNum = int(r["nearest"][j])  # index into the truth cloud (:3611)
                self.matchedID.append(p.matchNum)
        return r["count"]
