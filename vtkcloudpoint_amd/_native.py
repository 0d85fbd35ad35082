"""ctypes binding of libvcp.so (include/vcp.h).  No CPU fallback: if the HIP library is missing or
no GPU is present every compute call raises."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvcp.so")

L1_2D, L2_2D, L2_3D, SIGNED_SUM_2D = 0, 1, 2, 3
STOP_SSE_DELTA, STOP_RMSE = 0, 1

STATUS = {
    0: "VCP_OK", -1: "VCP_ERR_ARG", -2: "VCP_ERR_EMPTY", -3: "VCP_ERR_DEGENERATE", -4: "VCP_ERR_INDEX",
    -5: "VCP_ERR_TOO_LARGE", -6: "VCP_ERR_NO_DEVICE", -7: "VCP_ERR_HIP", -8: "VCP_ERR_UNSUPPORTED",
    -9: "VCP_ERR_NOMEM",
}

# every symbol include/vcp.h declares (tests check that the library exports all of them)
SYMBOLS = [
    "vcp_create", "vcp_destroy", "vcp_last_error", "vcp_version", "vcp_set_stream", "vcp_dev_alloc",
    "vcp_dev_free", "vcp_h2d", "vcp_d2h", "vcp_timing_enable", "vcp_timing_count", "vcp_timing_get",
    "vcp_dbscan", "vcp_dbscan_dev", "vcp_dbscan_blocks", "vcp_blocks_begin", "vcp_blocks_begin_dev",
    "vcp_blocks_share", "vcp_blocks_cluster_dev", "vcp_blocks_finish_dev", "vcp_centroids", "vcp_centroids_dev",
    "vcp_merge_centroids", "vcp_refresh_by_dictionary", "vcp_icp", "vcp_icp_dev", "vcp_icp_sums",
    "vcp_match", "vcp_mcc", "vcp_assign_truths", "vcp_icp_vtklike", "vcp_import_convert",
    "vcp_slab_begin", "vcp_slab_comps", "vcp_slab_finish", "vcp_release_workspace", "vcp_selftest_scan_dev",
    "vcp_centroids_weighted", "vcp_dbscan_blocks_keyed", "vcp_blocks_begin_keyed", "vcp_blocks_begin_keyed_dev",
    "vcp_selftest_horn", "vcp_create_multi", "vcp_destroy_multi", "vcp_multi_last_error", "vcp_multi_count",
    "vcp_multi_ctx", "vcp_dbscan_blocks_multi", "vcp_blocks_share_plan", "vcp_blocks_plan_dev", "vcp_blocks_plan_cuts",
    "vcp_blocks_build_dev", "vcp_blocks_finish_local_dev", "vcp_blocks_finish_zero_dev", "vcp_blocks_finish_zcoords_dev",
    "vcp_blocks_finish_pairs_dev", "vcp_scatter_pairs_dev",
]


class VcpError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("%s (%d): %s" % (STATUS.get(code, "?"), code, msg))
        self.code = code


_lib = None


def lib():
    """Load libvcp.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libvcp.so not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                              "or `make -C vtkcloudpoint_amd/csrc`")
        # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 with the same soname as
        # /opt/rocm's.  Whichever loads first serves both; letting torch load first keeps torch.cuda and
        # torch.distributed (RCCL) working next to libvcp in the same process.
        try:
            import torch  # noqa: F401
        except Exception:
            pass
        _lib = C.CDLL(LIB_PATH)
        _lib.vcp_last_error.restype = C.c_char_p
        _lib.vcp_last_error.argtypes = [C.c_void_p]
        _lib.vcp_create.argtypes = [C.c_int, C.POINTER(C.c_void_p)]
        _lib.vcp_destroy.argtypes = [C.c_void_p]
        _lib.vcp_destroy.restype = None
        _lib.vcp_multi_last_error.restype = C.c_char_p
        _lib.vcp_multi_last_error.argtypes = [C.c_void_p]
        _lib.vcp_destroy_multi.argtypes = [C.c_void_p]
        _lib.vcp_destroy_multi.restype = None
        _lib.vcp_multi_ctx.restype = C.c_void_p
        _lib.vcp_multi_ctx.argtypes = [C.c_void_p, C.c_int]
    return _lib


def _ptr(a):
    if a is None:
        return None
    if isinstance(a, int):
        return C.c_void_p(a)
    return C.c_void_p(a.ctypes.data)


def _f64(a, cols=None):
    a = np.ascontiguousarray(a, dtype=np.float64)
    if cols is not None:
        a = a.reshape(-1, cols)
    return a


class Context:
    """One vcp_ctx: one GPU, one stream.  Not thread-safe; create one per thread."""

    def __init__(self, device=0):
        self._h = C.c_void_p()
        rc = lib().vcp_create(int(device), C.byref(self._h))
        if rc != 0:
            raise VcpError(rc, (lib().vcp_last_error(None) or b"").decode())

    def close(self):
        if self._h:
            lib().vcp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc):
        if rc != 0:
            raise VcpError(rc, (lib().vcp_last_error(self._h) or b"").decode())

    # -- plumbing ------------------------------------------------------------------------------
    def set_stream(self, stream_handle):
        self._chk(lib().vcp_set_stream(self._h, C.c_void_p(stream_handle)))

    def selftest_scan_dev(self, in_ptr, out_ptr, n, op=0):
        tot = C.c_uint32(0)
        self._chk(lib().vcp_selftest_scan_dev(self._h, C.c_void_p(in_ptr), C.c_void_p(out_ptr), C.c_int64(n), C.c_int(op),
                                              C.byref(tot)))
        return tot.value

    def release_workspace(self):
        """Free the device workspace kept between calls (it is re-allocated on demand)."""
        self._chk(lib().vcp_release_workspace(self._h))

    def timing_enable(self, on=True):
        self._chk(lib().vcp_timing_enable(self._h, int(on)))

    def timing(self):
        out = []
        name = C.c_char_p()
        ms = C.c_float()
        for i in range(lib().vcp_timing_count(self._h)):
            self._chk(lib().vcp_timing_get(self._h, i, C.byref(name), C.byref(ms)))
            out.append((name.value.decode(), ms.value))
        return out

    # -- DBSCAN --------------------------------------------------------------------------------
    def dbscan(self, coords, eps, min_pts, metric=L1_2D, cf_in=0, in_classed=None, labels=None,
               in_mask=None):
        """Host-buffer entry point.  Returns dict(labels, is_core, is_classed, cf, evals)."""
        coords = _f64(coords)
        if coords.ndim != 2:
            coords = coords.reshape(0, 2)
        n, dim = coords.shape
        if in_classed is not None:
            in_classed = np.ascontiguousarray(in_classed, np.uint8)
            labels = np.array(labels if labels is not None else np.zeros(n), np.int32)
        else:
            labels = np.zeros(n, np.int32)
        if in_mask is not None:
            in_mask = np.ascontiguousarray(in_mask, np.uint8)
        is_core = np.zeros(n, np.uint8)
        is_classed = np.zeros(n, np.uint8)
        cf = C.c_int32(0)
        ev = C.c_int64(0)
        self._chk(lib().vcp_dbscan(self._h, _ptr(coords), C.c_int64(n), int(dim), int(metric),
                                   C.c_double(eps), int(min_pts), C.c_int32(cf_in), _ptr(in_mask),
                                   _ptr(in_classed), _ptr(labels), _ptr(is_core), _ptr(is_classed),
                                   C.byref(cf), C.byref(ev)))
        return dict(labels=labels, is_core=is_core, is_classed=is_classed, cf=cf.value, evals=ev.value)

    def dbscan_dev(self, d_coords, n, dim, eps, min_pts, metric=L1_2D, cf_in=0, d_in_classed=None,
                   d_labels=None, d_is_core=None, d_is_classed=None):
        """Device-pointer entry point (ints from tensor.data_ptr()).  Returns (cf, evals)."""
        cf = C.c_int32(0)
        ev = C.c_int64(0)
        self._chk(lib().vcp_dbscan_dev(self._h, _ptr(d_coords), C.c_int64(n), int(dim), int(metric),
                                       C.c_double(eps), int(min_pts), C.c_int32(cf_in), _ptr(d_in_classed),
                                       _ptr(d_labels), _ptr(d_is_core), _ptr(d_is_classed), C.byref(cf),
                                       C.byref(ev)))
        return cf.value, ev.value

    # -- ICP -----------------------------------------------------------------------------------
    def icp(self, model, data, tol=1e-4, max_iter=100, stop_rule=STOP_SSE_DELTA):
        model = _f64(model, 3)
        data = _f64(data, 3)
        R = np.zeros(9)
        T = np.zeros(3)
        sse, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
        self._chk(lib().vcp_icp(self._h, _ptr(model), C.c_int64(len(model)), _ptr(data), C.c_int64(len(data)),
                                C.c_double(tol), int(max_iter), int(stop_rule), _ptr(R), _ptr(T),
                                C.byref(sse), C.byref(rmse), C.byref(it)))
        return dict(R=R.reshape(3, 3), T=T, sse=sse.value, rmse=rmse.value, iters=it.value)

    def icp_dev(self, d_model, nm, d_data, nd, tol=1e-4, max_iter=100, stop_rule=STOP_SSE_DELTA):
        R = np.zeros(9)
        T = np.zeros(3)
        sse, rmse, it = C.c_double(0), C.c_double(0), C.c_int32(0)
        self._chk(lib().vcp_icp_dev(self._h, _ptr(d_model), C.c_int64(nm), _ptr(d_data), C.c_int64(nd),
                                    C.c_double(tol), int(max_iter), int(stop_rule), _ptr(R), _ptr(T),
                                    C.byref(sse), C.byref(rmse), C.byref(it)))
        return dict(R=R.reshape(3, 3), T=T, sse=sse.value, rmse=rmse.value, iters=it.value)

    def icp_sums(self, model, data, R=None, T=None, want_nn=True):
        model = _f64(model, 3)
        data = _f64(data, 3)
        R = None if R is None else _f64(R).reshape(9)
        T = None if T is None else _f64(T).reshape(3)
        sums = np.zeros(16)
        nn = np.zeros(len(data), np.int32) if want_nn else None
        self._chk(lib().vcp_icp_sums(self._h, _ptr(model), C.c_int64(len(model)), _ptr(data),
                                     C.c_int64(len(data)), _ptr(R), _ptr(T), _ptr(sums), _ptr(nn)))
        return sums, nn

    # -- centroids / merge / match -----------------------------------------------------------------
    def centroids(self, xyz, motor, labels, K):
        """Tools.GetClusList: returns (c3 [K,3], c2 [K,2], counts [K]); empty clusters are NaN rows."""
        xyz = None if xyz is None else _f64(xyz, 3)
        motor = None if motor is None else _f64(motor, 2)
        labels = np.ascontiguousarray(labels, np.int32)
        c3 = np.full((K, 3), np.nan)
        c2 = np.full((K, 2), np.nan)
        counts = np.zeros(K, np.int64)
        self._chk(lib().vcp_centroids(self._h, _ptr(xyz), _ptr(motor), _ptr(labels), C.c_int64(len(labels)),
                                      C.c_int32(K), _ptr(c3), _ptr(c2), _ptr(counts)))
        return c3, c2, counts

    def centroids_weighted(self, xyz, group, cluster_id, pts_count, K, ignore_duplication):
        """Tools.getFixedPtsCentroid: returns (c3 [K,3], inside_num [K])."""
        xyz = _f64(xyz, 3)
        group = np.ascontiguousarray(group, np.int32)
        cluster_id = None if cluster_id is None else np.ascontiguousarray(cluster_id, np.int32)
        pts_count = np.ascontiguousarray(pts_count, np.int32)
        c3 = np.zeros((K, 3))
        inside = np.zeros(K, np.int64)
        self._chk(lib().vcp_centroids_weighted(self._h, _ptr(xyz), _ptr(group), _ptr(cluster_id), _ptr(pts_count),
                                               C.c_int64(len(group)), C.c_int32(K), int(bool(ignore_duplication)),
                                               _ptr(c3), _ptr(inside)))
        return c3, inside

    def centroids_dev(self, d_xyz, d_motor, d_labels, n, K, d_c3, d_c2, d_counts):
        """Device-pointer form of centroids (any of d_xyz / d_motor and its output may be None)."""
        self._chk(lib().vcp_centroids_dev(self._h, _ptr(d_xyz), _ptr(d_motor), _ptr(d_labels), C.c_int64(n),
                                          C.c_int32(K), _ptr(d_c3), _ptr(d_c2), _ptr(d_counts)))

    def merge_centroids(self, cxy, ids, thr):
        cxy = _f64(cxy, 2)
        ids = np.ascontiguousarray(ids, np.int32)
        K = len(ids)
        map_to = np.zeros(K, np.int32)
        mc = C.c_int32(0)
        self._chk(lib().vcp_merge_centroids(self._h, _ptr(cxy), _ptr(ids), C.c_int32(K), C.c_double(thr),
                                            _ptr(map_to), C.byref(mc)))
        return map_to, mc.value

    def refresh_by_dictionary(self, xyz, motor, labels, K, map_by_id):
        xyz = _f64(xyz, 3)
        motor = _f64(motor, 2)
        labels = np.array(labels, np.int32)
        map_by_id = np.ascontiguousarray(map_by_id, np.int32)
        c3 = np.zeros((K, 3))
        c2 = np.zeros((K, 2))
        counts = np.zeros(K, np.int64)
        nk = C.c_int32(0)
        self._chk(lib().vcp_refresh_by_dictionary(self._h, _ptr(xyz), _ptr(motor), _ptr(labels),
                                                  C.c_int64(len(labels)), C.c_int32(K), _ptr(map_by_id),
                                                  C.byref(nk), _ptr(c3), _ptr(c2), _ptr(counts)))
        k = nk.value
        return labels, k, c3[:k], c2[:k], counts[:k]

    def match(self, centers, truths, M, max_dist):
        centers = _f64(centers, 3)
        truths = _f64(truths, 3)
        M = _f64(M).reshape(16)
        K, T = len(centers), len(truths)
        mxyz = np.zeros((K, 3))
        is_m = np.zeros(K, np.uint8)
        nearest = np.zeros(K, np.int32)
        nd = np.zeros(K)
        cnt = C.c_int32(0)
        self._chk(lib().vcp_match(self._h, _ptr(centers), C.c_int32(K), _ptr(truths), C.c_int32(T), _ptr(M),
                                  C.c_double(max_dist), _ptr(mxyz), _ptr(is_m), _ptr(nearest), _ptr(nd),
                                  C.byref(cnt)))
        return dict(matched_xyz=mxyz, is_matched=is_m, nearest=nearest, nearest_dist=nd, count=cnt.value)

    # -- block-partitioned pipeline ------------------------------------------------------------------
    def dbscan_blocks(self, motor, eps, min_pts, pts_in_cell, small_max=3, key_xy=None):
        """MainForm.getClusterFromMotor + StartCode + CompleteWork3 in one call (host buffers).  key_xy: the (X, Y)
        the partition of the twin getClusterFromList reads (FrmMain.cs:1136-1213); None = the motor coordinates."""
        motor = _f64(motor, 2)
        n = len(motor)
        key_xy = None if key_xy is None else _f64(key_xy, 2)
        labels = np.zeros(n, np.int32)
        block_of = np.zeros(n, np.int32)
        order = np.zeros(max(n, 1), np.int64)
        m = C.c_int64(0)
        rows, cols, kept, dels, ca = (C.c_int32(0) for _ in range(5))
        ev = C.c_int64(0)
        self._chk(lib().vcp_dbscan_blocks_keyed(self._h, _ptr(key_xy), _ptr(motor), C.c_int64(n), C.c_double(eps),
                                                int(min_pts), int(pts_in_cell), int(small_max), _ptr(labels),
                                                _ptr(block_of), _ptr(order), C.byref(m), C.byref(rows), C.byref(cols),
                                                C.byref(kept), C.byref(dels), C.byref(ca), C.byref(ev)))
        return dict(labels=labels, block_of=block_of, order=order[: m.value].copy(), rows=rows.value,
                    cols=cols.value, kept=kept.value, del_sum=dels.value, cluster_amount=ca.value,
                    evals=ev.value)

    def blocks_begin(self, motor, eps, min_pts, pts_in_cell, small_max=3, device_ptr=None, n=None, key_xy=None,
                     key_device_ptr=None):
        rows, cols = C.c_int32(0), C.c_int32(0)
        nb, m = C.c_int64(0), C.c_int64(0)
        if key_xy is not None or key_device_ptr is not None:  # getClusterFromList: partition on (X, Y)
            if device_ptr is None:
                motor, key_xy = _f64(motor, 2), _f64(key_xy, 2)
                self._chk(lib().vcp_blocks_begin_keyed(self._h, _ptr(key_xy), _ptr(motor), C.c_int64(len(motor)),
                                                       C.c_double(eps), int(min_pts), int(pts_in_cell), int(small_max),
                                                       C.byref(rows), C.byref(cols), C.byref(nb), C.byref(m)))
            else:
                self._chk(lib().vcp_blocks_begin_keyed_dev(self._h, _ptr(key_device_ptr), _ptr(device_ptr), C.c_int64(n),
                                                           C.c_double(eps), int(min_pts), int(pts_in_cell),
                                                           int(small_max), C.byref(rows), C.byref(cols), C.byref(nb),
                                                           C.byref(m)))
        elif device_ptr is None:
            motor = _f64(motor, 2)
            self._chk(lib().vcp_blocks_begin(self._h, _ptr(motor), C.c_int64(len(motor)), C.c_double(eps),
                                             int(min_pts), int(pts_in_cell), int(small_max), C.byref(rows),
                                             C.byref(cols), C.byref(nb), C.byref(m)))
        else:
            self._chk(lib().vcp_blocks_begin_dev(self._h, _ptr(device_ptr), C.c_int64(n), C.c_double(eps),
                                                 int(min_pts), int(pts_in_cell), int(small_max), C.byref(rows),
                                                 C.byref(cols), C.byref(nb), C.byref(m)))
        return dict(rows=rows.value, cols=cols.value, nblocks=nb.value, m=m.value)

    def blocks_share(self, rank, world):
        lo, hi = C.c_int32(0), C.c_int32(0)
        plo, phi = C.c_int64(0), C.c_int64(0)
        self._chk(lib().vcp_blocks_share(self._h, int(rank), int(world), C.byref(lo), C.byref(hi), C.byref(plo),
                                         C.byref(phi)))
        return lo.value, hi.value, plo.value, phi.value

    def blocks_cluster_dev(self, block_lo, block_hi, d_local):
        ev = C.c_int64(0)
        self._chk(lib().vcp_blocks_cluster_dev(self._h, C.c_int32(block_lo), C.c_int32(block_hi), _ptr(d_local),
                                               C.byref(ev)))
        return ev.value

    def blocks_finish_dev(self, d_local, evals_blocks, d_labels, d_block_of=None, d_merge_order=None):
        m = C.c_int64(0)
        kept, dels, ca = (C.c_int32(0) for _ in range(3))
        ev = C.c_int64(0)
        self._chk(lib().vcp_blocks_finish_dev(self._h, _ptr(d_local), C.c_int64(evals_blocks), _ptr(d_labels),
                                              _ptr(d_block_of), _ptr(d_merge_order), C.byref(m), C.byref(kept),
                                              C.byref(dels), C.byref(ca), C.byref(ev)))
        return dict(m=m.value, kept=kept.value, del_sum=dels.value, cluster_amount=ca.value, evals=ev.value)

    # -- the block pipeline with every stage sharded (include/vcp.h; driver: distributed.sharded_pipeline) -----------
    def blocks_plan(self, d_motor, n, eps, min_pts, pts_in_cell, small_max=3, d_key=None):
        """The streaming passes that decide the partition (identical on every rank).  d_* are device pointers."""
        rows, cols = C.c_int32(0), C.c_int32(0)
        nb, ns = C.c_int64(0), C.c_int64(0)
        self._chk(lib().vcp_blocks_plan_dev(self._h, _ptr(d_key), _ptr(d_motor), C.c_int64(n), C.c_double(eps),
                                            int(min_pts), int(pts_in_cell), int(small_max), C.byref(rows), C.byref(cols),
                                            C.byref(nb), C.byref(ns)))
        return dict(rows=rows.value, cols=cols.value, nblocks=nb.value, nsuper=ns.value)

    def blocks_plan_cuts(self, world):
        cuts = (C.c_int64 * (world + 1))()
        self._chk(lib().vcp_blocks_plan_cuts(self._h, int(world), cuts))
        return [int(c) for c in cuts]

    def blocks_build(self, super_lo, super_hi):
        lo, hi = C.c_int32(0), C.c_int32(0)
        m, nl = C.c_int64(0), C.c_int64(0)
        self._chk(lib().vcp_blocks_build_dev(self._h, C.c_int64(super_lo), C.c_int64(super_hi), C.byref(lo), C.byref(hi),
                                             C.byref(m), C.byref(nl)))
        return dict(block_lo=lo.value, block_hi=hi.value, m=m.value, n_loc=nl.value)

    def blocks_finish_local(self, d_local):
        info = (C.c_int64 * 8)()
        self._chk(lib().vcp_blocks_finish_local_dev(self._h, _ptr(d_local), info))
        keys = ("clusters", "kept", "err", "req", "nonempty", "last_nonzero", "m", "n_loc")
        return dict(zip(keys, (int(v) for v in info)))

    def blocks_finish_zero(self, zero_last):
        """-> (points of the share's zero list, those of them the noise pass can reach: the ones it runs over)"""
        z, a = C.c_int64(0), C.c_int64(0)
        self._chk(lib().vcp_blocks_finish_zero_dev(self._h, int(bool(zero_last)), C.byref(z), C.byref(a)))
        return z.value, a.value

    def blocks_finish_zcoords(self, d_zcoords, swap_xy=True):
        self._chk(lib().vcp_blocks_finish_zcoords_dev(self._h, int(bool(swap_xy)), _ptr(d_zcoords)))

    def blocks_finish_pairs(self, kept_offset, d_zlab, d_pairs):
        self._chk(lib().vcp_blocks_finish_pairs_dev(self._h, C.c_int32(kept_offset), _ptr(d_zlab), _ptr(d_pairs)))

    def scatter_pairs(self, d_pairs, count, n, d_labels):
        self._chk(lib().vcp_scatter_pairs_dev(self._h, _ptr(d_pairs), C.c_int64(count), C.c_int64(n), _ptr(d_labels)))

    # -- exact DBSCAN over several GPUs: staged engine (all d_* are device pointers) --------------------
    def slab_begin(self, d_coords, n, dim, metric, eps, min_pts, d_noexpand, d_ord, d_rep, d_is_core=None):
        """Grid, core flags and local components of own + halo points; returns the number of local components."""
        nc = C.c_int64(0)
        self._chk(lib().vcp_slab_begin(self._h, _ptr(d_coords), C.c_int64(n), C.c_int(dim), C.c_int(int(metric)),
                                       C.c_double(eps), C.c_int(min_pts), _ptr(d_noexpand), _ptr(d_ord), _ptr(d_rep),
                                       _ptr(d_is_core), C.byref(nc)))
        self._slab_ncomp = nc.value
        return nc.value

    def slab_comps(self):
        """Seeds (smallest global list position) of the local components, ascending."""
        out = np.zeros(self._slab_ncomp, np.uint32)
        self._chk(lib().vcp_slab_comps(self._h, _ptr(out)))
        out.sort()
        return out

    def slab_finish(self, map_rep, map_k, tab_gid, tab_seed, own_lo, own_count, d_labels, d_is_classed=None):
        """Border rule and labels from the resolved global clusters; returns the `twice` count of own points."""
        map_rep = np.ascontiguousarray(map_rep, np.uint32)
        map_k = np.ascontiguousarray(map_k, np.uint32)
        tab_gid = np.ascontiguousarray(tab_gid, np.int32)
        tab_seed = np.ascontiguousarray(tab_seed, np.uint32)
        tw = C.c_int64(0)
        self._chk(lib().vcp_slab_finish(self._h, _ptr(map_rep), _ptr(map_k), C.c_int64(len(tab_gid)), _ptr(tab_gid),
                                        _ptr(tab_seed), C.c_uint32(own_lo), C.c_uint32(own_count), _ptr(d_labels),
                                        _ptr(d_is_classed), C.byref(tw)))
        return tw.value

    def mcc(self, xy, labels, K, order=None):
        """Tools.getCircles: minimal bounding circle of every cluster with more than 3 points."""
        xy = _f64(xy, 2)
        labels = np.ascontiguousarray(labels, np.int32)
        order = None if order is None else np.ascontiguousarray(order, np.int64)
        n = len(labels)
        m = n if order is None else len(order)
        centers = np.zeros((K, 2))
        radius = np.zeros(K)
        valid = np.zeros(K, np.uint8)
        hn = np.zeros(K, np.int32)
        self._chk(lib().vcp_mcc(self._h, _ptr(xy), _ptr(labels), _ptr(order), C.c_int64(m), C.c_int64(n), C.c_int32(K),
                                _ptr(centers), _ptr(radius), _ptr(valid), _ptr(hn)))
        return dict(centers=centers, radius=radius, valid=valid, hull_n=hn)

    def assign_truths(self, motor, truths_xy, truth_ids, radius):
        """MainForm.refreshClusList: (ids [n], number of points with no truth within radius)."""
        motor = _f64(motor, 2)
        truths_xy = _f64(truths_xy, 2)
        truth_ids = np.ascontiguousarray(truth_ids, np.int32)
        ids = np.zeros(len(motor), np.int32)
        out = C.c_int64(0)
        self._chk(lib().vcp_assign_truths(self._h, _ptr(motor), C.c_int64(len(motor)), _ptr(truths_xy), _ptr(truth_ids),
                                          C.c_int32(len(truth_ids)), C.c_double(radius), _ptr(ids), C.byref(out)))
        return ids, out.value

    def icp_vtklike(self, source, target, max_iter=100, max_landmarks=200, start_by_centroids=True):
        """MainForm.ICP() (FrmMain.cs:841-907) without VTK: returns dict(M 4x4, mean_dist, iters)."""
        source = _f64(source, 3)
        target = _f64(target, 3)
        M = np.zeros(16)
        md = C.c_double(0)
        it = C.c_int32(0)
        self._chk(lib().vcp_icp_vtklike(self._h, _ptr(source), C.c_int64(len(source)), _ptr(target),
                                        C.c_int64(len(target)), int(max_iter), int(max_landmarks),
                                        int(start_by_centroids), _ptr(M), C.byref(md), C.byref(it)))
        return dict(M=M.reshape(4, 4), mean_dist=md.value, iters=it.value)

    def import_convert(self, rows, x_angle=0.0, y_angle=0.0, xdir=2, ydir=1, dedupe=True):
        """MainForm.AddFolder per-row work: dict(xyz [n,3], state [n] (0 filtered / 1 kept / 2 duplicate), kept, duplicates)."""
        rows = _f64(rows, 3)
        n = len(rows)
        xyz = np.zeros((n, 3))
        state = np.zeros(n, np.uint8)
        kept, dup = C.c_int64(0), C.c_int64(0)
        self._chk(lib().vcp_import_convert(self._h, _ptr(rows), C.c_int64(n), C.c_double(x_angle), C.c_double(y_angle),
                                           int(xdir), int(ydir), int(dedupe), _ptr(xyz), _ptr(state), C.byref(kept),
                                           C.byref(dup)))
        return dict(xyz=xyz, state=state, kept=kept.value, duplicates=dup.value)


def blocks_share_plan(blockstart, world):
    """vcp_blocks_share_plan: first block of every rank, [world + 1] (pure host arithmetic, no device)."""
    bs = np.ascontiguousarray(blockstart, np.uint32)
    cuts = np.zeros(int(world) + 1, np.int64)
    rc = lib().vcp_blocks_share_plan(_ptr(bs), C.c_int64(len(bs) - 1), C.c_int(int(world)), _ptr(cuts))
    if rc != 0:
        raise VcpError(rc, "vcp_blocks_share_plan")
    return cuts


class MultiContext:
    """vcp_multi: several GPUs driven from this one process (one vcp_ctx and one host thread per listed device; an id
    may repeat).  dbscan_blocks = Context.dbscan_blocks with the per-block step sharded over the devices."""

    def __init__(self, device_ids):
        ids = (C.c_int * len(device_ids))(*[int(d) for d in device_ids])
        self._h = C.c_void_p()
        rc = lib().vcp_create_multi(ids, len(device_ids), C.byref(self._h))
        if rc != 0:
            raise VcpError(rc, (lib().vcp_multi_last_error(None) or b"").decode())

    def close(self):
        if self._h:
            lib().vcp_destroy_multi(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def count(self):
        return int(lib().vcp_multi_count(self._h))

    def dbscan_blocks(self, motor, eps, min_pts, pts_in_cell, small_max=3, key_xy=None):
        motor = _f64(motor, 2)
        n = len(motor)
        key_xy = None if key_xy is None else _f64(key_xy, 2)
        labels = np.zeros(n, np.int32)
        block_of = np.zeros(n, np.int32)
        order = np.zeros(max(n, 1), np.int64)
        m = C.c_int64(0)
        rows, cols, kept, dels, ca = (C.c_int32(0) for _ in range(5))
        ev = C.c_int64(0)
        rc = lib().vcp_dbscan_blocks_multi(self._h, _ptr(key_xy), _ptr(motor), C.c_int64(n), C.c_double(eps), int(min_pts),
                                           int(pts_in_cell), int(small_max), _ptr(labels), _ptr(block_of), _ptr(order),
                                           C.byref(m), C.byref(rows), C.byref(cols), C.byref(kept), C.byref(dels),
                                           C.byref(ca), C.byref(ev))
        if rc != 0:
            raise VcpError(rc, (lib().vcp_multi_last_error(self._h) or b"").decode())
        return dict(labels=labels, block_of=block_of, order=order[: m.value].copy(), rows=rows.value, cols=cols.value,
                    kept=kept.value, del_sum=dels.value, cluster_amount=ca.value, evals=ev.value)
