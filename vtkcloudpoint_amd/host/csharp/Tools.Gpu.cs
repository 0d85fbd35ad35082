// Tools.Gpu.cs -- the Tools statics on the hot path, as a drop-in: add `partial` to the declaration in
// vtkPointCloud/BaseClass/Tools.cs:11 (`partial class Tools`), delete the six originals named below from that file
// (same names, same signatures, same in-place effects) and add this file to the project.  Nothing else in Tools.cs
// changes; the callers (FrmMain.cs:1533, Clustering.cs:125-181, SureDistanceFilter.cs:74, FrmMain.cs:1539-1540) stay
// as they are.
using System;
using System.Collections.Generic;
using System.Linq;

namespace vtkPointCloud
{
    partial class Tools
    {
        // Tools.GetClusList, Tools.cs:162-195
        public static void GetClusList(List<Point3D> rawData, List<Point3D> centers, List<Point3D> centers2D,
                                       List<ClusObj> clusList, List<int> idList)
        {
            int n = rawData.Count, K = clusList.Count;
            double[] xyz = new double[3 * n], mot = new double[2 * n];
            int[] lab = new int[n];
            for (int i = 0; i < n; i++)
            {
                Point3D p = rawData[i];
                xyz[3 * i] = p.X; xyz[3 * i + 1] = p.Y; xyz[3 * i + 2] = p.Z;
                mot[2 * i] = p.motor_x; mot[2 * i + 1] = p.motor_y;
                lab[i] = p.clusterId;
                if (p.clusterId != 0) clusList[p.clusterId - 1].li.Add(p);
            }
            if (K == 0 || n == 0) return;
            double[] c3 = new double[3 * K], c2 = new double[2 * K];
            long[] cnt = new long[K];
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_centroids(c.Ctx, xyz, mot, lab, n, K, c3, c2, cnt));
            for (int k = 0; k < K; k++)
            {
                if (cnt[k] == 0) continue;
                centers.Add(new Point3D(c3[3 * k], c3[3 * k + 1], c3[3 * k + 2], clusList[k].clusId, true));
                centers2D.Add(new Point3D(c2[2 * k], c2[2 * k + 1], 0, clusList[k].clusId, true));
            }
        }

        // Tools.MergeIDByDistance, Tools.cs:580-621
        public static Dictionary<int, int> MergeIDByDistance(List<Point3D> centers, double thre)
        {
            Dictionary<int, int> dick = new Dictionary<int, int>();
            int K = centers.Count;
            if (K == 0) return dick;
            double[] cxy = new double[2 * K];
            int[] ids = new int[K], mapTo = new int[K];
            for (int k = 0; k < K; k++)
            {
                Point3D p = centers[k];
                p.IDBeforeMerge = p.clusterId; p.motor_x = p.X; p.motor_y = p.Y; p.clusterId = 0;
                cxy[2 * k] = p.X; cxy[2 * k + 1] = p.Y; ids[k] = p.IDBeforeMerge;
            }
            int mergeCount;
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_merge_centroids(c.Ctx, cxy, ids, K, thre, mapTo, out mergeCount));
            for (int k = 0; k < K; k++) if (mapTo[k] != 0) dick.Add(ids[k], mapTo[k]);
            return dick;
        }

        // Tools.refreshCensAndClusByDictionary, Tools.cs:521-572 (caller Clustering.cs:125-181): the points of every list
        // whose id is a key of dic move to the END of list dic[id] (in clusList order), the merged lists go, the rest
        // is renumbered 1..K' by ascending id and every centroid is recomputed.  List surgery stays here (it is the
        // caller-visible structure); relabelling + renumbering + the K' x 5 means are one native call.
        static public void refreshCensAndClusByDictionary(Dictionary<int, int> dic, List<ClusObj> clusList,
                                                          ref List<Point3D> centers, ref List<Point3D> centers2D)
        {
            int K0 = clusList.Count;
            // like the C#, a target is addressed by POSITION dic[id] - 1 (clusList[k].clusId == k + 1 on entry, Tools.cs:531)
            foreach (ClusObj ob in clusList.ToArray())
                if (dic.ContainsKey(ob.clusId)) clusList[dic[ob.clusId] - 1].li.AddRange(ob.li);
            clusList.RemoveAll(delegate(ClusObj o) { return dic.ContainsKey(o.clusId); });
            clusList.Sort(delegate(ClusObj a, ClusObj b) { return a.clusId.CompareTo(b.clusId); });   // ids are distinct
            int n = 0, K = 0;
            foreach (ClusObj ob in clusList) { n += ob.li.Count; K = Math.Max(K, ob.clusId); }
            K = Math.Max(K, K0);
            double[] xyz = new double[3 * Math.Max(n, 1)], mot = new double[2 * Math.Max(n, 1)];
            int[] lab = new int[Math.Max(n, 1)], mapById = new int[Math.Max(K, 1)];   // the moves are done: identity map
            int t = 0;
            foreach (ClusObj ob in clusList)
                foreach (Point3D p in ob.li)
                {
                    xyz[3 * t] = p.X; xyz[3 * t + 1] = p.Y; xyz[3 * t + 2] = p.Z;
                    mot[2 * t] = p.motor_x; mot[2 * t + 1] = p.motor_y;
                    lab[t++] = ob.clusId;
                }
            int newK = 0;
            double[] c3 = new double[3 * Math.Max(K, 1)], c2 = new double[2 * Math.Max(K, 1)];
            long[] cnt = new long[Math.Max(K, 1)];
            if (n > 0)
                using (VcpNative.Lease c = VcpNative.Rent())
                    VcpNative.Check(c, VcpNative.vcp_refresh_by_dictionary(c.Ctx, xyz, mot, lab, n, K, mapById, out newK, c3, c2, cnt));
            // surviving ids in ascending order take 1..K' (:551-560).  A surviving list WITHOUT points keeps its number in
            // the C# too (idForMerge counts lists, the native call counts ids that still have points): number by position
            int idForMerge = 0, k = 0;
            t = 0;
            foreach (ClusObj ob in clusList)
            {
                ob.clusId = ++idForMerge;
                foreach (Point3D pp in ob.li) pp.clusterId = idForMerge;
                if (ob.li.Count == 0)
                {   // obj.li.Average on an empty list throws InvalidOperationException in the C# (:563)
                    throw new InvalidOperationException("Sequence contains no elements");
                }
                centers.Add(new Point3D(c3[3 * k], c3[3 * k + 1], c3[3 * k + 2], ob.clusId, true));
                centers2D.Add(new Point3D(c2[2 * k], c2[2 * k + 1], 0, ob.clusId, true));
                k++;
            }
            Console.WriteLine("keys中包含" + dic.Count + "质心有" + centers.Count);
        }

        // Tools.getFixedPtsCentroid, Tools.cs:78-111 (caller SureDistanceFilter.cs:74)
        static public List<Point3D> getFixedPtsCentroid(List<ClusObj> clusList, bool isIgnoreDuplication)
        {
            List<Point3D> scanCen = new List<Point3D>();
            int K = clusList.Count, n = 0;
            for (int i = 0; i < K; i++) n += clusList[i].li.Count;
            if (K == 0) return scanCen;
            double[] xyz = new double[3 * Math.Max(n, 1)];
            int[] group = new int[Math.Max(n, 1)], cid = new int[Math.Max(n, 1)], cnt = new int[Math.Max(n, 1)];
            int t = 0;
            for (int i = 0; i < K; i++)
                foreach (Point3D p in clusList[i].li)
                {
                    xyz[3 * t] = p.X; xyz[3 * t + 1] = p.Y; xyz[3 * t + 2] = p.Z;
                    group[t] = i + 1; cid[t] = p.clusterId; cnt[t] = p.ptsCount; t++;
                }
            double[] c3 = new double[3 * K];
            long[] inside = new long[K];
            // an empty list: VCP_ERR_INDEX, where the C# throws ArgumentOutOfRangeException at li[0] (:106)
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_centroids_weighted(c.Ctx, xyz, group, cid, cnt, n, K,
                isIgnoreDuplication ? 1 : 0, c3, inside));
            for (int i = 0; i < K; i++)
            {
                Point3D tmp = new Point3D();
                tmp.X = c3[3 * i]; tmp.Y = c3[3 * i + 1]; tmp.Z = c3[3 * i + 2];
                tmp.pointName = clusList[i].li[0].pointName;
                tmp.ifShown = true;
                scanCen.Add(tmp);
            }
            return scanCen;
        }

        // Tools.getCircles, Tools.cs:394-409 (Geometry.FindMinimalBoundingCircle, Geometry.cs:247-319)
        static public List<Point2D> getCircles(List<ClusObj> clusList, bool is3D)
        {
            List<Point2D> circles = new List<Point2D>();
            int K = clusList.Count, n = 0;
            for (int j = 0; j < K; j++) n += clusList[j].li.Count;
            if (K == 0 || n == 0) return circles;
            double[] xy = new double[2 * n];
            int[] lab = new int[n];
            int t = 0;
            for (int j = 0; j < K; j++)
                foreach (Point3D p in clusList[j].li)
                {   // FindMinimalBoundingCircle reads X,Y for the 3-D view, motor_x,motor_y for the 2-D one
                    xy[2 * t] = is3D ? p.X : p.motor_x; xy[2 * t + 1] = is3D ? p.Y : p.motor_y; lab[t] = j + 1; t++;
                }
            double[] cen = new double[2 * K], rad = new double[K];
            byte[] valid = new byte[K];
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_mcc(c.Ctx, xy, lab, null, n, n, K, cen, rad, valid, null));
            for (int j = 0; j < K; j++)
            {
                if (valid[j] == 0) continue;   // li.Count <= 3 (:400)
                Point2D c = new Point2D(cen[2 * j], cen[2 * j + 1]);
                c.radius = rad[j];
                c.clusID = j + 1;
                circles.Add(c);
            }
            return circles;
        }

        // MainForm.getClusterFromMotor + DoWork3 + the labelling half of CompleteWork3
        // (FrmMain.cs:1214-1291, :1340-1361, :1442-1520) as one blocking call; returns clusForMerge.
        // partitionOnXY: the twin getClusterFromList (:1136-1213), whose rectangles are cut on (X, Y).
        // VcpNative.Devices = { 0, 1, ..., 7 } spreads the per-block step over those GPUs (one process, vcp_dbscan_blocks_multi).
        public static List<Point3D> ClusterBlocks(List<Point3D> rawData, double tr, int pts, int ptsInCell,
                                                  bool partitionOnXY, out int clusterAmount)
        {
            int n = rawData.Count;
            double[] mot = new double[2 * n];
            double[] key = partitionOnXY ? new double[2 * n] : null;
            for (int i = 0; i < n; i++)
            {
                mot[2 * i] = rawData[i].motor_x; mot[2 * i + 1] = rawData[i].motor_y;
                if (partitionOnXY) { key[2 * i] = rawData[i].X; key[2 * i + 1] = rawData[i].Y; }
            }
            int[] lab = new int[n], blk = new int[n];
            long[] order = new long[Math.Max(n, 1)];
            long m, ev; int rows, cols, kept, del;
            if (VcpNative.Devices.Length > 1)
            {   // several GPUs: the per-block DBImproved calls (StartCode on pool threads, FrmMain.cs:1356-1359) are split over
                // VcpNative.Devices by contiguous block ranges; same results, bit for bit
                lock (VcpNative.MultiLock)
                    VcpNative.CheckMulti(VcpNative.vcp_dbscan_blocks_multi(VcpNative.Multi, key, mot, n, tr, pts, ptsInCell, 3,
                        lab, blk, order, out m, out rows, out cols, out kept, out del, out clusterAmount, out ev));
            }
            else
            {
                using (VcpNative.Lease c = VcpNative.Rent(VcpNative.Devices.Length == 1 ? VcpNative.Devices[0] : VcpNative.Device))
                    VcpNative.Check(c, VcpNative.vcp_dbscan_blocks_keyed(c.Ctx, key, mot, n, tr, pts, ptsInCell, 3, lab, blk,
                        order, out m, out rows, out cols, out kept, out del, out clusterAmount, out ev));
            }
            for (int i = 0; i < n; i++) { rawData[i].clusterId = lab[i]; rawData[i].isClassed = lab[i] != 0; }
            List<Point3D> clusForMerge = new List<Point3D>((int)m);
            for (long t = 0; t < m; t++) clusForMerge.Add(rawData[(int)order[t]]);
            return clusForMerge;
        }
    }
}
