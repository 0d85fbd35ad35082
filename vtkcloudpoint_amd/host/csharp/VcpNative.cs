// VcpNative.cs -- P/Invoke declarations for libvcp.so / vcp.dll (include/vcp.h).  Drop next to the classes
// in vtkPointCloud/BaseClass/.  NOT COMPILED IN THIS REPO'S IMAGE (no dotnet/mono/csc here): kept as the
// reference-side binding a maintainer adds; the executable mirrors are ../cpp/vcp_host.hpp and the Python
// package.  Target: .NET Framework 3.5+ (blittable arrays are pinned by the marshaller, no copies).
using System;
using System.Runtime.InteropServices;

namespace vtkPointCloud
{
    internal static class VcpNative
    {
        const string Lib = "vcp";   // libvcp.so on Linux (Mono/.NET), vcp.dll on Windows

        public const int VCP_L1_2D = 0, VCP_L2_2D = 1, VCP_L2_3D = 2, VCP_SIGNED_SUM_2D = 3;
        public const int VCP_STOP_SSE_DELTA = 0, VCP_STOP_RMSE = 1;

        [DllImport(Lib)] public static extern int vcp_create(int device_id, out IntPtr ctx);
        [DllImport(Lib)] public static extern void vcp_destroy(IntPtr ctx);
        [DllImport(Lib)] public static extern IntPtr vcp_last_error(IntPtr ctx);

        [DllImport(Lib)] public static extern int vcp_dbscan(IntPtr ctx, double[] coords, long n, int dim, int metric,
            double eps, int min_pts, int cf_in, byte[] in_mask, byte[] in_classed, int[] labels, byte[] is_core,
            byte[] is_classed, out int cf_out, out long dist_evals);

        [DllImport(Lib)] public static extern int vcp_dbscan_blocks(IntPtr ctx, double[] motor, long n, double eps,
            int min_pts, int pts_in_cell, int small_max, int[] labels, int[] block_of, long[] merge_order, out long m_out,
            out int rows, out int cols, out int kept, out int del_sum, out int cluster_amount, out long dist_evals);

        [DllImport(Lib)] public static extern int vcp_centroids(IntPtr ctx, double[] xyz, double[] motor, int[] labels,
            long n, int K, double[] c3, double[] c2, long[] counts);

        [DllImport(Lib)] public static extern int vcp_merge_centroids(IntPtr ctx, double[] cxy, int[] ids, int K,
            double thr, int[] map_to, out int merge_count);

        [DllImport(Lib)] public static extern int vcp_icp(IntPtr ctx, double[] model, long nm, double[] data, long nd,
            double tol, int max_iter, int stop_rule, double[] R, double[] T, out double sse, out double rmse, out int iters);

        [DllImport(Lib)] public static extern int vcp_match(IntPtr ctx, double[] centers, int K, double[] truths, int T,
            double[] M, double max_dist, double[] matched_xyz, byte[] is_matched, int[] nearest, double[] nearest_dist,
            out int count_matched);

        // one context per thread: StartCode runs on ThreadPool threads (FrmMain.cs:1358)
        [ThreadStatic] static IntPtr tlsCtx;
        public static IntPtr Ctx
        {
            get
            {
                if (tlsCtx == IntPtr.Zero)
                {
                    int rc = vcp_create(0, out tlsCtx);
                    if (rc != 0) throw new InvalidOperationException("vcp_create: " + Marshal.PtrToStringAnsi(vcp_last_error(IntPtr.Zero)));
                }
                return tlsCtx;
            }
        }
        public static void Check(int rc)
        {
            if (rc != 0) throw new InvalidOperationException("vcp error " + rc + ": " + Marshal.PtrToStringAnsi(vcp_last_error(tlsCtx)));
        }
    }
}
