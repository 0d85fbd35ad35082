// ICP.cs -- drop-in replacement for vtkPointCloud/BaseClass/ICP.cs: go_hell_ICP keeps its signature and
// writes R (3x3 Matrix) and T (3x1 Matrix) in place (caller: FrmMain.cs:2685-2690).  The arithmetic is the
// intended Besl-McKay/Horn loop; the shipped body cannot run past round 1 (see SURVEY.md fact 4).
using System;
using System.Collections.Generic;

namespace vtkPointCloud
{
    class ICP
    {
        public int maxIter = 1000;

        static double[] Flatten(List<Point3D> l)
        {
            double[] a = new double[3 * l.Count];
            for (int i = 0; i < l.Count; i++) { a[3 * i] = l[i].X; a[3 * i + 1] = l[i].Y; a[3 * i + 2] = l[i].Z; }
            return a;
        }

        public void go_hell_ICP(List<Point3D> model, List<Point3D> data, Matrix R, Matrix T, double e)
        {
            if (data.Count == 0) return;
            double[] r = new double[9], t = new double[3];
            double sse, rmse; int iters;
            using (VcpNative.Lease c = VcpNative.Rent())
                VcpNative.Check(c, VcpNative.vcp_icp(c.Ctx, Flatten(model), model.Count, Flatten(data), data.Count, e,
                maxIter, VcpNative.VCP_STOP_SSE_DELTA, r, t, out sse, out rmse, out iters));
            if (iters == 1 && sse < e) return;
            for (int i = 0; i < 3; i++)
            {
                for (int j = 0; j < 3; j++) R[i, j] = r[3 * i + j];
                T[i, 0] = t[i];
            }
        }
    }
}
