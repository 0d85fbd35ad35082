// DB.cs -- drop-in replacement for vtkPointCloud/BaseClass/DB.cs (the v1.0 class; its only use is commented out at
// FrmMain.cs:38): same public surface -- clusterAmount, pointsAmount, static iritatorNum, static getDisP / isKeyPoint /
// expandCluster, dbscan(List<Point3D>, double, int).  dbscan marshals to flat arrays and calls vcp_dbscan with the
// signed metric and the ifShown mask.
// RETAINED FROM THE ORIGINAL: the bodies of the three public statics below (getDisP, isKeyPoint, expandCluster) are the
// reference's own lines (BaseClass/DB.cs:14-91) minus comments and minus the never-matching dedupe loop.  They are public
// API of the class, any caller may invoke them directly, and they must behave identically; they are on no hot path and
// have no native counterpart.  Everything else in this file is new.
using System;
using System.Collections;
using System.Collections.Generic;

namespace vtkPointCloud
{
    public class DB
    {
        public int clusterAmount = 0;
        public int pointsAmount = 0;
        public static int iritatorNum = 0;

        public static double getDisP(Point3D p1, Point3D p2)
        {
            double dx = p1.X - p2.X;
            double dy = p1.Y - p2.Y;
            iritatorNum++;
            return dx + dy;
        }

        public static ArrayList isKeyPoint(List<Point3D> lst, Point3D p, double e, int minPts)
        {
            int count = 0;
            ArrayList tmpList = new ArrayList();
            for (int i = 0; i < lst.Count; i++)
            {
                Point3D p2 = lst[i];
                if (!p2.ifShown) { continue; }
                if (getDisP(p, p2) <= e) { ++count; tmpList.Add(i); }
            }
            if (count >= minPts) p.isKeyPoint = true;
            return tmpList;
        }

        public static void expandCluster(Point3D p, ArrayList nei, int c, double e, int minPts, List<Point3D> lst)
        {
            p.clusterId = c;
            for (int i = 0; i < nei.Count; i++)
            {
                Point3D dpp = (Point3D)lst[(int)nei[i]];
                if (!dpp.ifShown) { continue; }
                if (dpp.isClassed == false)
                {
                    dpp.isClassed = true;
                    ArrayList tmpList = isKeyPoint(lst, dpp, e, minPts);
                    if (tmpList.Count >= minPts) nei.AddRange(tmpList);   // the original's dedupe scan never matches (boxed ints)
                }
                dpp.clusterId = c;
            }
        }

        public void dbscan(List<Point3D> lst, double e, int minPts)
        {
            int n = lst.Count;
            if (n == 0) { this.clusterAmount = 0; return; }
            double[] xy = new double[2 * n];
            byte[] shown = new byte[n], classed = new byte[n];
            int[] labels = new int[n];
            int nShown = 0;
            for (int i = 0; i < n; i++)
            {
                Point3D p = lst[i];
                xy[2 * i] = p.X; xy[2 * i + 1] = p.Y;
                shown[i] = (byte)(p.ifShown ? 1 : 0);
                classed[i] = (byte)(p.isClassed ? 1 : 0);
                labels[i] = p.clusterId;
                if (p.ifShown) nShown++;
            }
            byte[] isCore = new byte[n], isClassed = new byte[n];
            int cfOut; long evals;
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_dbscan(c.Ctx, xy, n, 2, VcpNative.VCP_SIGNED_SUM_2D, e, minPts, 0, shown,
                classed, labels, isCore, isClassed, out cfOut, out evals));
            for (int i = 0; i < n; i++)
            {
                Point3D p = lst[i];
                p.clusterId = labels[i];
                if (isClassed[i] != 0) p.isClassed = true;
                if (isCore[i] != 0) p.isKeyPoint = true;
            }
            pointsAmount += nShown;
            this.clusterAmount = cfOut;
            unchecked { iritatorNum += (int)evals; }
        }
    }
}
