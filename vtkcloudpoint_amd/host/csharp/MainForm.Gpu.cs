// MainForm.Gpu.cs -- the MainForm members on the hot path, as a drop-in: MainForm is already `partial`
// (FrmMain.cs / FrmMain.Designer.cs); delete calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618) and
// refreshClusList (:3437-3467) from FrmMain.cs and add this file.  Field names are the reference's own
// (centers, trues, rawData, clusList, truePointCloud, M, matchedID, PtsInRegionTxt, toolStripStatusLabelCurrentPointCount).
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
            VcpNative.Check(VcpNative.vcp_match(VcpNative.Ctx, c, K, TruthArray(), (int)truePointCloud.GetNumberOfPoints(),
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
            if (n > 0) VcpNative.Check(VcpNative.vcp_assign_truths(VcpNative.Ctx, mot, n, txy, tid, T, clusterRadius, ids, out yedian));
            for (int i = 0; i < n; i++) if (ids[i] != 0) clusList[ids[i] - 1].li.Add(rawData[i]);
            this.toolStripStatusLabelCurrentPointCount.Text = String.Format("当前聚类个数：{0}，有效点个数： {1}，野点个数： {2}", (clusList.Count(i => i.li.Count != 0)), rawData.Count - yedian, yedian);
            addCircles();
        }
    }
}
