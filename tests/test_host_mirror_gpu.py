"""The reference's class surface driven the way FrmMain drives it, on the GPU: DBImproved on
List<Point3D>, the block pipeline + GetClusList, MergeIDByDistance + refresh, go_hell_ICP, matching."""
import numpy as np
import pytest

from vtkcloudpoint_amd import synth
from vtkcloudpoint_amd.datamodel import ClusObj, Point3D, points_from_arrays
from vtkcloudpoint_amd.dbscan import DBImproved
from vtkcloudpoint_amd.icp import ICP, Matrix
from vtkcloudpoint_amd.tools import ClusterPipeline, Matcher, Tools

pytestmark = pytest.mark.gpu


def test_dbimproved_like_startcode_and_noise_pass(vcp_ctx, oracle):
    d = synth.config_c1()
    lst = points_from_arrays(d["motor"], d["xyz"])
    before = DBImproved.iritatorNum
    db = DBImproved(vcp_ctx)  # FrmMain.cs:2785-2786
    db.dbscan(lst, 0.5, 10)
    o = oracle.dbscan(d["motor"], 0.5, 10, literal=True)
    assert [p.clusterId for p in lst] == o["labels"].tolist()
    assert [p.isClassed for p in lst] == o["classed"].astype(bool).tolist()
    assert [p.isKeyPoint for p in lst] == o["is_key"].astype(bool).tolist()
    assert db.clusterAmount == o["cf"] == db.cf and db.pointsAmount == len(lst)
    assert DBImproved.iritatorNum - before == o["evals"]
    # FrmMain.cs:1507-1516: second DBImproved over the noise with cf preset, isClassed reset by the caller
    zero = [p for p in lst if p.clusterId == 0]
    for p in zero:
        p.isClassed = False
    dbb = DBImproved(vcp_ctx)
    dbb.cf = db.clusterAmount
    dbb.dbscan(zero, 1.0, 3)
    zc = np.array([(p.motor_x, p.motor_y) for p in zero])
    o2 = oracle.dbscan(zc, 1.0, 3, 0, o["cf"], literal=True)
    assert [p.clusterId for p in zero] == o2["labels"].tolist() and dbb.clusterAmount == o2["cf"]
    # a second call on already-classed points: seeds are skipped but still count (DBImproved.cs:101)
    db3 = DBImproved(vcp_ctx)
    db3.dbscan(lst, 0.5, 10)
    lab_in = np.array([pp for pp in o["labels"]], np.int32)
    lab_in[[i for i, p in enumerate(lst) if p in zero]] = o2["labels"]
    cls_in = np.array([1 if l else 0 for l in lab_in], np.uint8)
    o3 = oracle.dbscan(d["motor"], 0.5, 10, 0, 0, cls_in, lab_in, literal=True)
    assert [p.clusterId for p in lst] == o3["labels"].tolist() and db3.clusterAmount == o3["cf"]


def test_block_pipeline_then_centroids_merge(vcp_ctx, oracle):
    d = synth.config_cloud(100_000, seed=17)
    raw = points_from_arrays(d["motor"], d["xyz"])
    mf = ClusterPipeline(raw, vcp_ctx)
    r = mf.getClusterFromMotor(0.1, 10, 200)
    o = oracle.block_pipeline(d["motor"], 0.1, 10, 200, 3)
    assert [p.clusterId for p in raw] == o["labels"].tolist() and mf.clusterSum == o["cluster_amount"]
    assert [raw.index(p) for p in mf.clusForMerge[:50]] == o["order"][:50].tolist()
    c3, c2, cnt = oracle.centroids(d["xyz"], d["motor"], o["labels"], o["cluster_amount"], o["order"])
    ne = cnt > 0
    assert len(mf.centers) == int(ne.sum())
    got = np.array([(p.X, p.Y, p.Z) for p in mf.centers])
    assert np.allclose(got, c3[ne], rtol=1e-12, atol=1e-12)
    assert [p.clusterId for p in mf.centers] == (np.nonzero(ne)[0] + 1).tolist()
    # Clustering.MergeBtn_Click (Clustering.cs:141-154): clone, merge by distance, refresh
    tmpCenters = [Point3D(p.X, p.Y, p.Z, p.clusterId, True) for p in mf.centers]
    tmpClus = []
    for ob in mf.clusList:
        c = ClusObj()
        c.clusId, c.li = ob.clusId, list(ob.li)
        tmpClus.append(c)
    dic = Tools.MergeIDByDistance(tmpCenters, 2.0, vcp_ctx)
    ids = (np.nonzero(ne)[0] + 1).astype(np.int32)
    mo, _ = oracle.merge_ids(c3[ne][:, :2], ids, 2.0)
    assert dic == {int(i): int(m) for i, m in zip(ids, mo) if m}
    if all(len(c.li) for c in tmpClus):
        newC, newC2 = [], []
        Tools.refreshCensAndClusByDictionary(dic, tmpClus, newC, newC2, vcp_ctx)
        assert len(tmpClus) == len(mf.clusList) - len(dic) == len(newC)
        assert [c.clusId for c in tmpClus] == list(range(1, len(tmpClus) + 1))


def test_go_hell_icp_and_matching(vcp_ctx, oracle):
    c = synth.config_icp(nd=20000, nm=100, jitter=0.0)
    model = points_from_arrays(None, c["model"])
    data = points_from_arrays(None, c["data"])
    R, T = Matrix(3, 3), Matrix(3, 1)
    icp = ICP(vcp_ctx)
    icp.go_hell_ICP(model, data, R, T, 1e-4)  # FrmMain.cs:2685-2690
    assert np.abs(R.to_numpy() - c["R_true"]).max() < 1e-5
    assert np.abs(T.to_numpy().ravel() - c["T_true"]).max() < 1e-5
    Y = icp.FindClosestPointSet(model, data[:100])
    assert [model.index(y) for y in Y] == oracle.find_closest(c["model"], c["data"][:100]).tolist()
    # matching: centroids = transformed truths, M = the ICP result as a 4x4
    M = np.eye(4)
    M[:3, :3] = R.to_numpy()
    M[:3, 3] = T.to_numpy().ravel()
    cen = []
    for i in range(50):
        p = Point3D()
        p.tmp_X, p.tmp_Y, p.tmp_Z = c["data"][i]
        cen.append(p)
    mt = Matcher(cen, c["model"], M, vcp_ctx)
    assert mt.RecorrectMatchingPtsByDistance(0.01) == 50
    assert [p.matchNum for p in cen] == [i % 100 for i in range(50)]


def test_cpp_host_mirror_demo():
    """vtkcloudpoint_amd/host/cpp/vcp_host.hpp (the C++ mirror of the C# classes) against hand-derived answers."""
    import os
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "vtkcloudpoint_amd", "host", "cpp",
                       "host_demo")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "PASS host_demo" in out.stdout, out.stdout + out.stderr


def test_scan_files_to_cluster_export(vcp_ctx, oracle, tmp_path):
    """File -> import (filter, spherical conversion, duplicate removal over all files) -> DBImproved -> export, in
    the reference's text formats; the import is checked against the oracle's literal AddFolder restatement."""
    from vtkcloudpoint_amd import io as vio
    from vtkcloudpoint_amd.datamodel import ClusObj
    from vtkcloudpoint_amd.dbscan import DBImproved
    rng = np.random.default_rng(12)
    files, allrows = [], []
    for k in range(3):
        n = 4000
        rows = np.empty((n, 3))
        c = rng.uniform(-20, 20, (8, 2))
        rows[:, :2] = np.round((c[rng.integers(0, 8, n)] + rng.normal(0, 0.4, (n, 2))) * 64) / 64
        rows[:, 2] = np.round(rng.uniform(5, 40, n) * 16) / 16
        rows[::97, 2] = 0.0          # filtered: Distance == 0
        rows[5::113, 2] = 1500.0     # filtered: Distance > 1000
        rows[10:60] = rows[200:250]  # exact duplicates inside a file
        if k:
            rows[300:330] = allrows[0][400:430]  # and across files
        f = tmp_path / ("scan%d.txt" % k)
        with open(f, "w", newline="\r\n") as fh:
            for r in rows:
                fh.write("%r\t%r\t%r\n" % (float(r[0]), float(r[1]), float(r[2])))
        files.append(str(f))
        allrows.append(rows)
    raw, dup, paths = vio.add_folder(files, x_angle=1.0, y_angle=-0.5, typpe=1, ctx=vcp_ctx)
    ref = oracle.import_convert(np.concatenate(allrows), 1.0, -0.5, 2, 1, True, literal=True)
    assert dup == ref["duplicates"] and len(raw) == ref["kept"] and dup > 100
    keep = np.nonzero(ref["state"] == 1)[0]
    got = np.array([(p.X, p.Y, p.Z) for p in raw])
    assert np.allclose(got, ref["xyz"][keep], rtol=1e-12, atol=1e-12)
    assert [p.pathId for p in raw] == (keep // 4000).tolist() and paths == files
    db = DBImproved(vcp_ctx)
    db.dbscan(raw, 0.5, 8)
    o = oracle.dbscan(np.array([(p.motor_x, p.motor_y) for p in raw]), 0.5, 8)
    assert [p.clusterId for p in raw] == o["labels"].tolist() and db.clusterAmount == o["cf"] > 3
    clus = [ClusObj() for _ in range(db.clusterAmount)]
    for p in raw:
        if p.clusterId:
            clus[p.clusterId - 1].li.append(p)
    out = tmp_path / "clusters.txt"
    vio.write_clusters_text(str(out), clus, bit=3)
    lines = open(out).read().split("\n")
    assert len(lines) - 1 == int((o["labels"] > 0).sum()) and lines[0].startswith("1\t")
