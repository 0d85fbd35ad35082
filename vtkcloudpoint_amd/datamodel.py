"""Host-side data model mirroring BaseClass/DataModel.cs: the fields of Point3D (:102-160) and ClusObj
(:14-64) that the clustering / matching path reads or writes.  These are plain mutable objects, like the
C# ones; the compute classes marshal lists of them into flat arrays for the C-ABI."""
import numpy as np


class Point3D:
    """BaseClass/DataModel.cs:102-160 (only the properties the path touches)."""
    __slots__ = ("IDBeforeMerge", "motor_x", "motor_y", "Distance", "X", "Y", "Z", "clusterId", "pathId",
                 "ifShown", "ptsCount", "isClassed", "isKeyPoint", "isMatched", "matchNum", "pointName",
                 "tmp_X", "tmp_Y", "tmp_Z", "matched_X", "matched_Y", "matched_Z", "isFilterByDistance")

    def __init__(self, xx=0.0, yy=0.0, zz=0.0, clusterId=0, isShown=False):
        self.X, self.Y, self.Z = float(xx), float(yy), float(zz)  # DataModel.cs:105-119
        self.clusterId = int(clusterId)
        self.ifShown = bool(isShown)
        self.IDBeforeMerge = 0
        self.motor_x = self.motor_y = self.Distance = 0.0
        self.pathId = self.ptsCount = self.matchNum = 0
        self.isClassed = self.isKeyPoint = self.isMatched = self.isFilterByDistance = False
        self.pointName = None
        self.tmp_X = self.tmp_Y = self.tmp_Z = 0.0
        self.matched_X = self.matched_Y = self.matched_Z = 0.0


class ClusObj:
    """BaseClass/DataModel.cs:14-64."""

    def __init__(self, clusName=None):
        self.li = []
        self.clusId = 0
        self.clusName = clusName
        self.visible = True
        self.ptsCount = 0


def points_from_arrays(motor=None, xyz=None):
    """Build a List<Point3D> from flat arrays (test / demo helper)."""
    n = len(motor) if motor is not None else len(xyz)
    out = []
    for i in range(n):
        p = Point3D()
        if motor is not None:
            p.motor_x, p.motor_y = float(motor[i][0]), float(motor[i][1])
        if xyz is not None:
            p.X, p.Y, p.Z = float(xyz[i][0]), float(xyz[i][1]), float(xyz[i][2])
        p.ifShown = True
        out.append(p)
    return out


def motor_array(lst):
    return np.array([(p.motor_x, p.motor_y) for p in lst], dtype=np.float64).reshape(-1, 2)


def xyz_array(lst):
    return np.array([(p.X, p.Y, p.Z) for p in lst], dtype=np.float64).reshape(-1, 3)
