// ToolsGpu.cs -- replacement bodies for the Tools statics on the path (vtkPointCloud/BaseClass/Tools.cs)
// and for MainForm's block pipeline.  Paste the bodies over the originals (same signatures).
using System;
using System.Collections.Generic;

namespace vtkPointCloud
{
    public static class ToolsGpu
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
            VcpNative.Check(VcpNative.vcp_centroids(VcpNative.Ctx, xyz, mot, lab, n, K, c3, c2, cnt));
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
            VcpNative.Check(VcpNative.vcp_merge_centroids(VcpNative.Ctx, cxy, ids, K, thre, mapTo, out mergeCount));
            for (int k = 0; k < K; k++) if (mapTo[k] != 0) dick.Add(ids[k], mapTo[k]);
            return dick;
        }

        // MainForm.getClusterFromMotor + DoWork3 + the labelling half of CompleteWork3
        // (FrmMain.cs:1214-1291, :1340-1361, :1442-1520) as one blocking call; returns clusForMerge.
        public static List<Point3D> ClusterBlocks(List<Point3D> rawData, double tr, int pts, int ptsInCell,
                                                  out int clusterAmount)
        {
            int n = rawData.Count;
            double[] mot = new double[2 * n];
            for (int i = 0; i < n; i++) { mot[2 * i] = rawData[i].motor_x; mot[2 * i + 1] = rawData[i].motor_y; }
            int[] lab = new int[n], blk = new int[n];
            long[] order = new long[Math.Max(n, 1)];
            long m, ev; int rows, cols, kept, del;
            VcpNative.Check(VcpNative.vcp_dbscan_blocks(VcpNative.Ctx, mot, n, tr, pts, ptsInCell, 3, lab, blk, order,
                out m, out rows, out cols, out kept, out del, out clusterAmount, out ev));
            for (int i = 0; i < n; i++) { rawData[i].clusterId = lab[i]; rawData[i].isClassed = lab[i] != 0; }
            List<Point3D> clusForMerge = new List<Point3D>((int)m);
            for (long t = 0; t < m; t++) clusForMerge.Add(rawData[(int)order[t]]);
            return clusForMerge;
        }
    }
}
