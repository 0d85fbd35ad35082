// DBImproved.cs -- drop-in replacement for vtkPointCloud/BaseClass/DBImproved.cs: same public surface
// (clusterAmount, pointsAmount, static iritatorNum, cf, static getDisP, dbscan(List<Point3D>, double, int)),
// same in-place mutation of the caller's Point3D objects; the body marshals to flat arrays and calls
// vcp_dbscan.  Callers stay unchanged: FrmMain.cs:1507-1516, :2785-2789, Tools.cs:591-592.
using System;
using System.Collections.Generic;

namespace vtkPointCloud
{
    public class DBImproved
    {
        public int clusterAmount = 0;
        public int pointsAmount = 0;
        public static int iritatorNum = 0;
        public int cf = 0;

        public static double getDisP(Point3D p1, Point3D p2)
        {
            double dx = p1.motor_x - p2.motor_x;
            double dy = p1.motor_y - p2.motor_y;
            iritatorNum++;
            return Math.Abs(dx) + Math.Abs(dy);
        }

        public void dbscan(List<Point3D> lst, double e, int minPts)
        {
            int n = lst.Count;
            if (n == 0) { this.clusterAmount = cf; return; }
            double[] xy = new double[2 * n];
            byte[] classed = new byte[n];
            int[] labels = new int[n];
            bool any = false;
            for (int i = 0; i < n; i++)
            {
                Point3D p = lst[i];
                xy[2 * i] = p.motor_x; xy[2 * i + 1] = p.motor_y;
                classed[i] = (byte)(p.isClassed ? 1 : 0);
                labels[i] = p.clusterId;
                any |= p.isClassed;
            }
            byte[] isCore = new byte[n], isClassed = new byte[n];
            int cfOut; long evals;
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_dbscan(c.Ctx, xy, n, 2, VcpNative.VCP_L1_2D, e, minPts, cf, null,
                any ? classed : null, labels, isCore, isClassed, out cfOut, out evals));
            for (int i = 0; i < n; i++)
            {
                Point3D p = lst[i];
                if (any || labels[i] != 0) p.clusterId = labels[i];
                if (isClassed[i] != 0) p.isClassed = true;
                if (isCore[i] != 0) p.isKeyPoint = true;
            }
            pointsAmount += n;
            cf = cfOut;
            this.clusterAmount = cf;
            unchecked { iritatorNum += (int)evals; }   // the C# counter is a 32-bit int and wraps the same way
        }
    }
}
