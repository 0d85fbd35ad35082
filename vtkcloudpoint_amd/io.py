"""The reference's text formats on either side of the path (host side; the numeric work of an import runs on the
GPU through vcp_import_convert).

* scan files          one point per line, `motor_x<TAB>motor_y<TAB>Distance` -- what MainForm.AddFolder reads
                      (FrmMain.cs:1005-1009 via FileMap.ReadFile, BaseClass/FileMap.cs:16-33) and what
                      Tools.ExportPoints-style code writes back (BaseClass/Tools.cs:233).
* clustering export   `clusterId<TAB>motor_x<TAB>motor_y<TAB>Distance`, clusters in list order
                      (BaseClass/Tools.cs:366-392).

Numbers are written with C#'s "F<bit>" format.  Python's fixed-point formatting rounds the exact binary value
half-to-even while .NET Framework 3.5 formats from a 15-digit decimal expansion; the two differ only for values
whose 16th significant digit decides a tie -- not reproduced here (no .NET in this image to pin it).
"""
import numpy as np

from .datamodel import Point3D


def read_scan_text(path):
    """rows [n, 3] = (motor_x, motor_y, Distance) of a scan file.  Like `Convert.ToDouble(tmpxyz[k])`
    (FrmMain.cs:1006-1008) a line with fewer than three fields or a non-numeric field is an error; extra fields
    are ignored."""
    rows = []
    with open(path, "r", encoding="gb2312", errors="replace") as f:  # FileMap.cs:24 reads as GB2312
        for ln, line in enumerate(f.read().splitlines(), 1):
            parts = line.split("\t")
            if len(parts) < 3:
                raise ValueError("%s:%d: expected motor_x<TAB>motor_y<TAB>Distance" % (path, ln))
            try:
                rows.append((float(parts[0]), float(parts[1]), float(parts[2])))
            except ValueError:
                raise ValueError("%s:%d: not a number" % (path, ln))
    return np.array(rows, dtype=np.float64).reshape(-1, 3)


def write_scan_text(path, rawData, bit=6):
    """BaseClass/Tools.cs:233: one `motor_x<TAB>motor_y<TAB>Distance` line per point, "F<bit>"."""
    fmt = "%%.%df\t%%.%df\t%%.%df\n" % (bit, bit, bit)
    with open(path, "w", encoding="ascii", newline="\r\n") as f:  # StreamWriter.WriteLine on Windows
        for p in rawData:
            f.write(fmt % (p.motor_x, p.motor_y, p.Distance))


def write_clusters_text(path, clusList, bit=6):
    """BaseClass/Tools.cs:366-392: `clusterId<TAB>motor_x<TAB>motor_y<TAB>Distance`, cluster by cluster."""
    fmt = "%%d\t%%.%df\t%%.%df\t%%.%df\n" % (bit, bit, bit)
    with open(path, "w", encoding="ascii", newline="\r\n") as f:
        for clus in clusList:
            for p in clus.li:
                f.write(fmt % (p.clusterId, p.motor_x, p.motor_y, p.Distance))


def add_folder(files, x_angle=0.0, y_angle=0.0, xdir=2, ydir=1, typpe=1, ctx=None):
    """MainForm.AddFolder for scan points (FrmMain.cs:960-1100, typpe 1 = drop exact duplicates, 2 = keep them):
    reads the files in order, filters `Distance == 0 || Distance > 1000`, converts (motor_x, motor_y, Distance) to
    X, Y, Z and, for typpe 1, drops every point whose X, Y, Z equal those of a point already in the list -- over
    ALL files read so far, as `rawData` keeps growing.  Returns (rawData: list of Point3D, duplicatNum, pathList).
    The conversion and the duplicate search run on the GPU in one call."""
    if typpe not in (1, 2):
        raise NotImplementedError("typpe 3/4 (fixed points per file) are UI bookkeeping, not on the GPU path")
    from . import runtime
    ctx = ctx or runtime.default_context()
    rows, path_id = [], []
    for k, f in enumerate(files):
        r = read_scan_text(f)
        rows.append(r)
        path_id.append(np.full(len(r), k, np.int32))
    rows = np.concatenate(rows) if rows else np.zeros((0, 3))
    path_id = np.concatenate(path_id) if path_id else np.zeros(0, np.int32)
    res = ctx.import_convert(rows, x_angle, y_angle, xdir, ydir, dedupe=(typpe == 1))
    rawData = []
    for i in np.nonzero(res["state"] == 1)[0]:
        p = Point3D(res["xyz"][i, 0], res["xyz"][i, 1], res["xyz"][i, 2], 0, True)
        p.motor_x, p.motor_y, p.Distance = rows[i]
        p.pathId = int(path_id[i])
        rawData.append(p)
    return rawData, int(res["duplicates"]), list(files)
