// MainForm.Gpu.cs -- the MainForm members on the hot path, as a drop-in: MainForm is already `partial`
// (FrmMain.cs / FrmMain.Designer.cs); delete calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618),
// refreshClusList (:3437-3467) and ICP() (:841-907) from FrmMain.cs, replace the row loop of AddFolder (:991-1090) by
// the call shown at AddScanRows below, and add this file.  Field names are the reference's own (centers, trues,
// rawData, clusList, truePointCloud, truePointVertices, M, ren, vtkControl, matchedID, x_angle, y_angle, pathList,
// PtsInRegionTxt, toolStripStatusLabelCurrentPointCount, trueScale, centroidScale, scale, clock, clock_y, clock_x).
using System;
using System.Collections.Generic;
using System.Linq;
using System.Windows.Forms;

namespace vtkPointCloud
{
    public partial class MainForm
    {
        double[] gpuMatched;      // matched_X/Y/Z of every centroid, filled by calMatchedCoords
        int[] gpuNearest;
        double[] gpuNearestDist;

        static double[] Matrix16(vtk.vtkMatrix4x4 m)
        {
            double[] a = new double[16];
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) a[4 * r + c] = m.GetElement(r, c);
            return a;
        }

        double[] TruthArray()
        {
            int T = (int)truePointCloud.GetNumberOfPoints();
            double[] t = new double[3 * T];
            for (int i = 0; i < T; i++) { double[] p = truePointCloud.GetPoint(i); t[3 * i] = p[0]; t[3 * i + 1] = p[1]; t[3 * i + 2] = p[2]; }
            return t;
        }

        // FrmMain.cs:3572-3587.  The transform and the nearest-truth search are one native call; the distances are
        // kept for RecorrectMatchingPtsByDistance, which the UI calls again whenever the threshold changes.
        public void calMatchedCoords()
        {
            int K = centers.Count;
            if (K == 0) return;
            double[] c = new double[3 * K];
            for (int j = 0; j < K; j++) { c[3 * j] = centers[j].tmp_X; c[3 * j + 1] = centers[j].tmp_Y; c[3 * j + 2] = centers[j].tmp_Z; }
            gpuMatched = new double[3 * K];
            gpuNearest = new int[K];
            gpuNearestDist = new double[K];
            byte[] isM = new byte[K];
            int cnt;
            using (VcpNative.Lease lease = VcpNative.Rent())
                VcpNative.Check(lease, VcpNative.vcp_match(lease.Ctx, c, K, TruthArray(), (int)truePointCloud.GetNumberOfPoints(),
                Matrix16(M), double.PositiveInfinity, gpuMatched, isM, gpuNearest, gpuNearestDist, out cnt));
            for (int j = 0; j < K; j++)
            {
                centers[j].matched_X = gpuMatched[3 * j];
                centers[j].matched_Y = gpuMatched[3 * j + 1];
                centers[j].matched_Z = gpuMatched[3 * j + 2];
                centers[j].isMatched = false;
            }
        }

        // FrmMain.cs:3588-3618
        public void RecorrectMatchingPtsByDistance(double matchDistance, bool isShowUnmatchedCenterPts, bool isShowUnmatchedTruePts)
        {
            int countMatched = 0;
            matchedID = new List<int>();
            if (gpuNearest == null || gpuNearest.Length != centers.Count) calMatchedCoords();
            for (int j = 0; j < centers.Count; j++)
            {
                centers[j].isMatched = false;
                if (gpuNearestDist[j] < matchDistance)
                {
                    centers[j].isMatched = true;
                    centers[j].matchNum = gpuNearest[j];
                    matchedID.Add(gpuNearest[j]);
                    countMatched++;
                }
            }
            this.toolStripStatusLabelCurrentPointCount.Text = "总共" + centers.Count + "个聚类质心，总共" + truePointCloud.GetNumberOfPoints() + "个真值点，匹配" + countMatched + "个点";
            showMatchedLine(isShowUnmatchedCenterPts, isShowUnmatchedTruePts);
        }

        // FrmMain.cs:3437-3467: nearest truth within the radius per raw point (the LINQ query :3452-3456)
        private void refreshClusList()
        {
            double clusterRadius;
            if (!double.TryParse(this.PtsInRegionTxt.Text, out clusterRadius))
            {
                MessageBox.Show("输入的文件格式有误，请重新输入");
                return;
            }
            isStartDrawCircle = true;
            foreach (ClusObj oj in clusList) oj.li.Clear();
            int n = rawData.Count, T = trues.Count;
            double[] mot = new double[2 * n], txy = new double[2 * T];
            int[] tid = new int[T], ids = new int[n];
            for (int i = 0; i < n; i++) { mot[2 * i] = rawData[i].motor_x; mot[2 * i + 1] = rawData[i].motor_y; }
            for (int s = 0; s < T; s++) { txy[2 * s] = trues[s].tmp_X; txy[2 * s + 1] = trues[s].tmp_Y; tid[s] = trues[s].clusterId; }
            long yedian = 0;
            if (n > 0)
                using (VcpNative.Lease lease = VcpNative.Rent())
                    VcpNative.Check(lease, VcpNative.vcp_assign_truths(lease.Ctx, mot, n, txy, tid, T, clusterRadius, ids, out yedian));
            for (int i = 0; i < n; i++) if (ids[i] != 0) clusList[ids[i] - 1].li.Add(rawData[i]);
            this.toolStripStatusLabelCurrentPointCount.Text = String.Format("当前聚类个数：{0}，有效点个数： {1}，野点个数： {2}", (clusList.Count(i => i.li.Count != 0)), rawData.Count - yedian, yedian);
            addCircles();
        }

        // FrmMain.cs:841-907.  The reference hands the centroids (tmp_X, tmp_Y, 0) -- Tools.ArrayList2PolyData type 1,
        // Tools.cs:696-703 -- and the truth points to VTK's closed vtkIterativeClosestPointTransform with RigidBody,
        // 100 iterations, StartByMatchingCentroidsOn and everything else at its defaults (:851-858).  vcp_icp_vtklike runs
        // that configuration (VTK 5.0 header: every ns/200-th source point is a landmark, closest target point per
        // landmark, rigid-body landmark fit per round); parity against VTK itself is unpinned (closed binary).
        // M receives the accumulated matrix like icp.GetMatrix() (:862); the display part is the reference's own, with a
        // plain vtkTransform carrying M in place of the icp object.
        void ICP()
        {
            ren = new vtk.vtkRenderer();
            vtk.vtkPolyData SourcePolydata = Tools.ArrayList2PolyData(1, this.centers, this.trueScale, this.centroidScale,
                this.scale, this.clock, this.clock_y, this.clock_x);
            vtk.vtkPolyData TargetPolydata = new vtk.vtkPolyData();
            TargetPolydata.SetPoints(truePointCloud);
            TargetPolydata.SetVerts(truePointVertices);

            int ns = centers.Count;
            double[] src = new double[3 * Math.Max(ns, 1)];
            for (int i = 0; i < ns; i++) { src[3 * i] = centers[i].tmp_X; src[3 * i + 1] = centers[i].tmp_Y; src[3 * i + 2] = 0.0; }
            double[] tgt = TruthArray();
            double[] m16 = new double[16];
            double meanDist; int iters;
            using (VcpNative.Lease lease = VcpNative.Rent())
                VcpNative.Check(lease, VcpNative.vcp_icp_vtklike(lease.Ctx, src, ns, tgt, tgt.Length / 3, 100, 200, 1, m16,
                    out meanDist, out iters));
            M = new vtk.vtkMatrix4x4();
            for (int r = 0; r < 4; r++) for (int c = 0; c < 4; c++) M.SetElement(r, c, m16[4 * r + c]);
            Console.WriteLine("刚性变换矩阵为：" + M);

            vtk.vtkTransform solved = new vtk.vtkTransform();
            solved.SetMatrix(M);
            vtk.vtkTransformPolyDataFilter icpTransformFilter = new vtk.vtkTransformPolyDataFilter();
            icpTransformFilter.SetInput(SourcePolydata);
            icpTransformFilter.SetTransform(solved);
            icpTransformFilter.Update();
            vtk.vtkPolyDataMapper targetMapper = new vtk.vtkPolyDataMapper();
            targetMapper.SetInputConnection(TargetPolydata.GetProducerPort());
            vtk.vtkActor targetActor = new vtk.vtkActor();
            targetActor.SetMapper(targetMapper);
            targetActor.GetProperty().SetColor(0, 1, 0);
            targetActor.GetProperty().SetPointSize(4);
            vtk.vtkPolyDataMapper solutionMapper = new vtk.vtkPolyDataMapper();
            solutionMapper.SetInputConnection(icpTransformFilter.GetOutputPort());
            vtk.vtkActor solutionActor = new vtk.vtkActor();
            solutionActor.SetMapper(solutionMapper);
            solutionActor.GetProperty().SetColor(0, 0, 1);
            solutionActor.GetProperty().SetPointSize(3);
            ren.AddActor(targetActor);
            ren.AddActor(solutionActor);
            vtkControl.GetRenderWindow().AddRenderer(ren);
            SourcePolydata.FastDelete();
            TargetPolydata.FastDelete();
        }

        // The row loop of AddFolder for scan files (FrmMain.cs:991-1090, typpe 1 = remove duplicates, 2 = keep them).
        // AddFolder keeps its file / tree-view code; per file it now only PARSES (FileMap.ReadFile + Split('\t') +
        // Convert.ToDouble, :1005-1008, or the xls cells :996-1001) into rows = (motor_x, motor_y, Distance) triples and
        // collects them, then calls this once for all files of the folder:
        //     List<double> rows = new List<double>(); List<int> rowPath = new List<int>();
        //     foreach file: foreach parsed line: rows.Add(mx); rows.Add(my); rows.Add(dist); rowPath.Add(pathList.Count);
        //                   pathList.Add(file);
        //     duplicatNum += AddScanRows(rows.ToArray(), rowPath.ToArray(), typpe, xdir, ydir);
        // One call = one native conversion: the Distance filter (:1011), the spherical conversion (:1025-1062) and, for
        // typpe 1, the duplicate test against every EARLIER kept row (:1063-1068: rawData.FindAll over the whole list, so
        // duplicates across files count too) run on the GPU (hash table instead of the O(n^2) FindAll); rows already in
        // rawData from an earlier AddFolder are passed in front so that they take part in the test.  Fixed-point files
        // (typpe 3 / 4) keep the reference's loop: a handful of rows per file.
        int AddScanRows(double[] rows, int[] rowPath, int typpe, int xdir, int ydir)
        {
            int nNew = rowPath.Length;
            int nOld = (typpe == 1) ? rawData.Count : 0;   // earlier points take part in the duplicate test only
            int n = nOld + nNew;
            if (nNew == 0) return 0;
            double[] all = new double[3 * n];
            for (int i = 0; i < nOld; i++) { all[3 * i] = rawData[i].motor_x; all[3 * i + 1] = rawData[i].motor_y; all[3 * i + 2] = rawData[i].Distance; }
            Array.Copy(rows, 0, all, 3 * nOld, 3 * nNew);
            double[] xyz = new double[3 * n];
            byte[] state = new byte[n];
            long kept, dup;
            using (VcpNative.Lease lease = VcpNative.Rent())
                VcpNative.Check(lease, VcpNative.vcp_import_convert(lease.Ctx, all, n, this.x_angle, this.y_angle, xdir, ydir,
                    typpe == 1 ? 1 : 0, xyz, state, out kept, out dup));
            int duplicates = 0;
            for (int i = nOld; i < n; i++)
            {
                if (state[i] == 0) continue;                  // Distance == 0 or > 1000 (:1011)
                if (state[i] == 2) { duplicates++; continue; }  // equals an earlier kept row (:1065-1068)
                Point3D point = new Point3D();
                point.motor_x = all[3 * i]; point.motor_y = all[3 * i + 1]; point.Distance = all[3 * i + 2];
                point.pathId = rowPath[i - nOld];
                point.ifShown = true;
                point.isClassed = false;
                point.clusterId = 0;
                point.X = xyz[3 * i]; point.Y = xyz[3 * i + 1]; point.Z = xyz[3 * i + 2];
                rawData.Add(point);
            }
            return duplicates;
        }
    }
}
