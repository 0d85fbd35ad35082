#!/usr/bin/env python3
"""bench.py -- headline benchmark of the DBSCAN + ICP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--mode single|blocks|exact|replicas]

metric = BASELINE.json's "Mpoints/s DBSCAN+label @10M pts"; one JSON line on rank 0 (driver contract).  With
--gpus N > 1 and no WORLD_SIZE in the environment the script starts `python -m torch.distributed.run` itself (as a
child process, before anything touches the GPU) and relays its output and exit code; a WORLD_SIZE that disagrees with
--gpus is an error, so a 1-GPU number can never be recorded as an N-GPU run.

Workloads (config.workload names the one that ran):
  single    (default at N = 1) one monolithic DBImproved.dbscan (BC/DBImproved.cs:91-114 semantics) over the 10 M-point
            C4 cloud resident in HBM, through vcp_dbscan_dev: grid build + region query + clusters + canonical labels.
  blocks    (default at N > 1; also runs at N = 1) the north_star's multi-GPU form: ONE 10 M-point cloud, the
            reference's block-partitioned pipeline at its UI defaults (eps 0.07, minPts 7, 200 points per block;
            FrmMain.cs:1214-1291 partition, :1358 one DBImproved per block, :1442-1520 CompleteWork3).  Every rank
            computes the identical partition, clusters a contiguous range of blocks, the block-major int32 labels are
            all-gathered over RCCL INSIDE the timed region, every rank finishes with CompleteWork3.  Strong scaling:
            total work is fixed, value = points / time.
  exact     ranks own adjacent x-slabs (10 M points each) of one cloud; exact monolithic result (weak scaling).
  replicas  every rank clusters its own 10 M-point cloud, ids made global (weak scaling; --gather-labels adds the
            label all-gather).  Kept as a named alternative; it is NOT the default because it exchanges 8 bytes.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X spec, /opt/skills/guides/MI355X_MICROARCH.md

# Algorithmic bytes per point of each kernel phase for the 2-D binary64 path: what the
# phase must read/write once if every neighbour access hits cache.
ALGO_BYTES_PER_POINT = {
    "bounds": 16,               # R coords
    "part_hist": 16,            # R coords (the chunk x bucket counts are ~0.4 B per point)
    "part_scatter": 16 + 16,    # R coords, W one 16-byte record (binary32 x, y relative to the grid origin, index, cell)
    # R records (the second read of a bucket is served by L2); W binary32 coordinates 8 + list position 4 +
    # 4 B per cell of the cell table (5 cells per point on this cloud)
    "part_fine": 16 + 8 + 4 + 20,
    # R own point 8, staged binary32 rows ~8; W flags 1, parent 4, minord 4, neighbour lists (3.6 entries
    # per point on this cloud) 14.5 + offset 2; work-list fill R 1 + W 2.4 (+ 0.1 clearing the seed bitmap)
    "core_count": 45,
    # list links R 1 + 9 per expanding point (a quarter of the points) ..., candidate parent words (cache-served, once
    # per point), binary32 coordinates of the expanding points
    "union": 21,
    "flatten_number": 10,       # work-list kernels over the expanding points + seed bitmap
    # rank extension R 5 + W 8 for every point; list walk (flags, ~3 list words, ~3 rank words, sord, labk) for the third
    # of the points that are border candidates
    "border": 24,
    "out_scatter": 8 + 8,       # R (list position, label word), W the 8-byte record
    "out_write": 8 + 6,         # R record, W label 4 + isKeyPoint 1 + isClassed 1
    # sort-based build and gather output (fallback for grids beyond the one-level partition)
    "cell_key": 24, "cell_sort": 48, "cell_scan": 24, "scatter": 48, "output": 14,
}

BLOCK_DEFAULTS = dict(eps=0.07, min_pts=7, pts_in_cell=200, small_max=3)  # Clustering.Designer.cs:86,96,158


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--points", type=int, default=10_000_000)
    ap.add_argument("--metric", default="L1_2D", choices=["L1_2D", "L2_3D"])
    ap.add_argument("--force-sharded", action="store_true",
                    help="mode blocks at one rank: run the sharded pipeline's per-rank program anyway (rehearsal)")
    ap.add_argument("--mode", default=None, choices=["single", "blocks", "exact", "replicas"],
                    help="default: single at --gpus 1, blocks at --gpus > 1 (see the module docstring)")
    ap.add_argument("--gather-labels", action="store_true",
                    help="mode replicas: also all-gather every rank's int32 labels onto every rank")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ICP / block pipeline side measurements")
    return ap.parse_args(argv)


def launch_ranks_if_needed(args):
    """--gpus N > 1 without a torch.distributed environment: run the same command under torch.distributed.run as a
    CHILD process (never an exec, and before torch or HIP is touched) and exit with its code."""
    if "WORLD_SIZE" in os.environ:
        world = int(os.environ["WORLD_SIZE"])
        if world != args.gpus:
            sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d; refusing to print a line for the wrong GPU count\n"
                             % (args.gpus, world))
            sys.exit(2)
        return
    if args.gpus <= 1:
        return
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get(
        "HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    sys.exit(subprocess.call(cmd, env=env))


# ----------------------------------------------------------------------------------------------------------
# the blocks workload: used by main() on GPUs and by tests/_bench_worker.py on CPU tensors (gloo + oracle backend)
# ----------------------------------------------------------------------------------------------------------
def blocks_step_single(backend, ptr, n, local, labels):
    """The block pipeline on ONE device: the staged single-device entry points back to back."""
    bd = BLOCK_DEFAULTS
    info = backend.blocks_begin(None, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"], device_ptr=ptr, n=n)
    ev = backend.blocks_cluster_dev(0, info["nblocks"], local.data_ptr())
    out = backend.blocks_finish_dev(local.data_ptr(), ev, labels.data_ptr())
    out.update(labels=labels[:n], nblocks=info["nblocks"], block_range=(0, info["nblocks"]), collective_bytes=0,
               rows=info["rows"], cols=info["cols"])
    return out


def blocks_workload(backend, motor, device, steps, warmup, motor_dev_ptr=None, sync=None, barrier=None,
                    on_step=None, sharded=None):
    """`steps` timed passes of the block pipeline over ONE cloud (every rank holds it): on several ranks every stage is
    sharded (distributed.sharded_pipeline: each rank builds, clusters and merges its own share of the blocks; the noise
    pass runs as exact_slabs; one all-gather of (index, label) pairs); on one rank the single-device entry points.  The
    job and its result are the same for every N.  Returns (seconds for the timed steps on this rank, result of the last
    step).  sync() drains the device, barrier() is the cross-rank barrier; both bracket the timed region."""
    import torch
    import torch.distributed as dist
    from vtkcloudpoint_amd import distributed as D
    n = len(motor)
    world = dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1
    if sharded is None:
        sharded = world > 1
    ptr = motor_dev_ptr if motor_dev_ptr is not None else motor.ctypes.data
    local = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    labels = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    sync = sync or (lambda: None)
    barrier = barrier or (lambda: None)
    bd = BLOCK_DEFAULTS

    def step():
        if sharded:
            return D.sharded_pipeline(backend, ptr, n, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"],
                                      device=device, labels=labels)
        return blocks_step_single(backend, ptr, n, local, labels)

    r = None
    for _ in range(warmup):
        r = step()
    sync()
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        r = step()
        if on_step:
            on_step(r)
    sync()
    barrier()
    sync()
    return time.perf_counter() - t0, r


# ----------------------------------------------------------------------------------------------------------
# CPU baselines, rank 0 at N = 1 only, bounded samples
# ----------------------------------------------------------------------------------------------------------
def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.lower().startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(cloud, coords, eps, min_pts, metric_id, budget_s=7.0):
    """B1..B4 of BASELINE.md section 3, each on a bounded sample (~5-10 s of CPU work).  The oracle is the
    single-thread C++ restatement of the C# (kind "port"): the reference itself cannot be compiled here."""
    import shutil
    from concurrent.futures import ThreadPoolExecutor
    from oracle import binding as O
    from vtkcloudpoint_amd import synth
    cores = os.cpu_count() or 1
    n = len(coords)
    centre = np.median(coords[cloud["blob"] == 0], axis=0)
    order = np.argsort(np.abs(coords - centre).max(axis=1), kind="stable")

    def window(k):  # spatial window round blob 0, list order kept: same density mix as the workload
        return np.ascontiguousarray(coords[np.sort(order[:k])])

    # B1: literal DBImproved.dbscan, O(n^2), with and without the never-matching dedupe scan of DBImproved.cs:70-83
    b1 = {}
    for tag, dd in (("dedupe_scan_off", False), ("dedupe_scan_on", True)):
        # grow the window until one run costs a fair share of the budget (cost ~ n^2 without the scan, steeper with
        # it: it adds ~(cluster size x neighbours)^2 reference compares per cluster); the last run is the sample
        k, dt, r, crop = 3000, 0.0, None, None
        while True:
            crop = window(min(k, n))
            t0 = time.time()
            r = O.dbscan(crop, eps, min_pts, metric_id, literal=True, dedupe=dd)
            dt = time.time() - t0
            if dt >= budget_s / 4.0 or len(crop) >= n:
                break
            k = int(k * 1.5)
        b1[tag] = {"value": len(crop) / dt / 1e6, "unit": "Mpoints/s", "cores": 1, "points": len(crop),
                   "seconds": round(dt, 2), "dist_evals": int(r["evals"]), "clusters": int(r["cf"])}
    off = b1["dedupe_scan_off"]
    # grid_port: the oracle's order-free O(n k) formulation -- what a CPU rewrite with the GPU path's idea costs
    crop2 = window(min(n, 3_000_000))
    t0 = time.time()
    r2 = O.dbscan(crop2, eps, min_pts, metric_id, literal=False)
    dt2 = time.time() - t0
    out = {
        "value": b1["dedupe_scan_on"]["value"], "unit": "Mpoints/s", "cores": 1, "kind": "port",
        "sample": "B1: literal C++ port of DBImproved.dbscan (O(n^2), WITH the never-matching dedupe scan of "
                  "DBImproved.cs:70-83, as BASELINE.md specifies) on the %d points of the same cloud nearest "
                  "(Chebyshev) to the centre of blob 0: %.1f s; the rate falls at least as 1/n"
                  % (b1["dedupe_scan_on"]["points"], b1["dedupe_scan_on"]["seconds"]),
        "cpu_model": cpu_model(), "host_cores": cores,
        "csharp_toolchain_on_this_host": [t for t in ("dotnet", "mono", "csc", "mcs") if shutil.which(t)],
        "B1_literal_dbscan": dict(b1, extrapolated_hours_at_full_size_dedupe_off=round(
            off["seconds"] * (n / max(off["points"], 1)) ** 2 / 3600.0, 1)),
        "grid_port": {"value": len(crop2) / dt2 / 1e6, "unit": "Mpoints/s", "cores": 1,
                      "sample": "oracle's canonical grid formulation on the %d nearest points: %.1f s, %d clusters"
                                % (len(crop2), dt2, r2["cf"])},
    }
    if metric_id == 0:
        # B2: block pipeline at the UI defaults.  The literal FindAll partition (Tools.cs:510-513) is O(n * blocks):
        # sample = the 600 k points nearest to blob 0; per-block DBImproved in its literal form, noise pass in the
        # grid form (its literal O(Z^2) form is what makes the C# pipeline unusable at this size)
        bd = BLOCK_DEFAULTS
        crop = window(min(n, 600_000))
        t0 = time.time()
        rb = O.block_pipeline(crop, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"], canonical=True, brute=True)
        t_one = time.time() - t0
        sb = O.StagedBlocks()
        t0 = time.time()
        info = sb.blocks_begin(crop, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"])
        local = np.zeros(max(info["m"], 1), np.int32)
        cuts = [sb.blocks_share(r, cores) for r in range(cores)]
        with ThreadPoolExecutor(cores) as ex:  # one block range per thread, like the reference's ThreadPool
            evs = list(ex.map(lambda c: sb.blocks_cluster_dev(c[0], c[1], local.ctypes.data), cuts))
        lab = np.zeros(len(crop), np.int32)
        sb.blocks_finish_dev(local.ctypes.data, int(sum(evs)), lab.ctypes.data)
        t_all = time.time() - t0
        out["B2_block_pipeline"] = {
            "sample": "%d points nearest to blob 0, eps %g, minPts %d, %d per block: %d blocks, %d clusters"
                      % (len(crop), bd["eps"], bd["min_pts"], bd["pts_in_cell"], rb["rows"] * rb["cols"], rb["cluster_amount"]),
            "one_thread_literal_partition": {"value": len(crop) / t_one / 1e6, "unit": "Mpoints/s", "cores": 1,
                                             "seconds": round(t_one, 2)},
            "all_cores_fast_partition": {"value": len(crop) / t_all / 1e6, "unit": "Mpoints/s", "cores": cores,
                                         "seconds": round(t_all, 2),
                                         "note": "sort-based partition (same blocks), per-block step on a thread per "
                                                 "block range, sequential CompleteWork3"},
        }
    # B3: intended ICP (ICP.cs:18-285 structure), 1 M data x 100 model, single thread; 10 rounds timed of C3's 50
    c = synth.config_icp(nd=1_000_000, nm=100, jitter=0.05)
    t0 = time.time()
    ri = O.icp(c["model"], c["data"], 0.0, 10, O.STOP_SSE_DELTA)
    dt = time.time() - t0
    out["B3_icp_1M_x_100"] = {"value": ri["iters"] / dt, "unit": "rounds/s", "cores": 1, "rounds": ri["iters"],
                              "seconds": round(dt, 2), "ms_for_50_rounds": round(dt / ri["iters"] * 50 * 1e3, 1)}
    return out


def cpu_baseline_match(cen, truth, M):
    """B4: calMatchedCoords + RecorrectMatchingPtsByDistance (FrmMain.cs:3572-3618), K centroids x T truths."""
    from oracle import binding as O
    t0 = time.time()
    r = O.match(cen, truth, M, 0.5)
    dt = time.time() - t0
    return {"value": len(cen) / dt / 1e6, "unit": "Mcentroids/s", "cores": 1, "K": len(cen), "T": len(truth),
            "seconds": round(dt, 3), "matched": int(r["count"])}


# ----------------------------------------------------------------------------------------------------------
def roofline_of(avg, n_points, dim, metric_name):
    """The dominant phase by measured time (hipEvents recorded by the library on its launch stream)."""
    if not avg:
        return None
    dom = max(avg, key=avg.get)
    bpp = ALGO_BYTES_PER_POINT.get(dom, 0) + (0 if dim == 2 else 8)
    achieved = bpp * n_points / (avg[dom] * 1e-3) / 1e9
    traffic = None
    pmc = os.path.join(ROOT, "profiles", "pmc_latest.json")
    if os.path.exists(pmc):
        try:
            with open(pmc) as f:
                traffic = json.load(f).get(metric_name, {}).get(dom)
        except Exception:
            traffic = None
    return {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
            "traffic_source": "profiles/pmc_latest.json: 2 x FETCH_SIZE + WRITE_SIZE of this phase's kernels from separate "
                              "rocprofv3 --pmc passes of the same command on an earlier run (tools/profile.sh); NOT "
                              "measured in this run" if traffic is not None else None,
            "algorithmic_bytes_per_point": bpp, "kernel_ms": avg[dom]}


def main():
    args = parse()
    launch_ranks_if_needed(args)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    mode = args.mode or ("single" if world == 1 else "blocks")
    if mode == "single" and world > 1:
        raise SystemExit("--mode single is the 1-GPU workload")
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libvcp has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:  # also at world size 1 under torch.distributed.run: the RCCL path is rehearsed
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    from vtkcloudpoint_amd import _native as N
    from vtkcloudpoint_amd import distributed as D
    from vtkcloudpoint_amd import synth

    n = args.points
    metric_id = N.L1_2D if args.metric == "L1_2D" else N.L2_3D
    one_cloud = mode in ("single", "blocks")
    cloud = synth.config_cloud(n, seed=4 if one_cloud else 4 + 1000 * rank)
    coords = cloud["motor"] if metric_id == N.L1_2D else cloud["xyz"]
    eps = cloud["eps_l1"] if metric_id == N.L1_2D else cloud["eps_l2"]
    min_pts = cloud["min_pts"]
    dim = coords.shape[1]
    if mode == "blocks" and metric_id != N.L1_2D:
        raise SystemExit("the block pipeline clusters on (motor_x, motor_y): --metric L1_2D")
    if mode == "exact":
        # adjacent slabs of one cloud: rank r's points are shifted by r extents along x (exactly representable)
        width = cloud["motor_extent"] if metric_id == N.L1_2D else 100.0 * (n / 1_000_000.0) ** (1.0 / 3.0)
        coords = coords.copy()
        coords[:, 0] += rank * float(np.ceil(width))

    dev = torch.device("cuda", local_rank)
    ctx = N.Context(local_rank)
    ctx.timing_enable(True)
    d_coords = torch.from_numpy(coords).to(dev)
    d_labels = torch.zeros(n, dtype=torch.int32, device=dev)
    d_core = torch.zeros(n, dtype=torch.uint8, device=dev)
    d_cls = torch.zeros(n, dtype=torch.uint8, device=dev)
    torch.cuda.synchronize()

    phase_ms = {}
    collectives = []
    extra_cfg = {}
    same_single = None

    def record(pairs):
        for name, ms in pairs:
            phase_ms.setdefault(name, []).append(ms)

    def sync():
        torch.cuda.synchronize()

    def barrier():
        if dist:
            dist.barrier()

    if mode == "blocks":
        # the timed loop runs without per-phase events; one more step afterwards, untimed, collects them
        ctx.timing_enable(False)
        shd = True if args.force_sharded else None
        dt, last = blocks_workload(ctx, coords, dev, 0, args.warmup, motor_dev_ptr=d_coords.data_ptr(), sync=sync,
                                   barrier=barrier, sharded=shd)
        dt, last = blocks_workload(ctx, coords, dev, args.steps, 0, motor_dev_ptr=d_coords.data_ptr(), sync=sync,
                                   barrier=barrier, sharded=shd)
        cf, total_points = last["cluster_amount"], n
        if rank == 0:
            # the SAME job on one GPU (the single-device entry points), outside the timed region: what the N-GPU line
            # above is a speed-up of
            loc1 = torch.zeros(n, dtype=torch.int32, device=dev)
            lab1 = torch.zeros(n, dtype=torch.int32, device=dev)
            best = None
            for _ in range(4):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                r1 = blocks_step_single(ctx, d_coords.data_ptr(), n, loc1, lab1)
                e = time.perf_counter() - t1
                best = e if best is None else min(best, e)
            same = bool(torch.equal(lab1[:n], last["labels"])) and r1["cluster_amount"] == last["cluster_amount"]
            same_single = {"ms": best * 1e3, "labels_identical_to_the_timed_run": same}
            # phases of the stages (library hipEvents), one untimed pass
            ctx.timing_enable(True)
            bd = BLOCK_DEFAULTS
            info = ctx.blocks_begin(None, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"],
                                    device_ptr=d_coords.data_ptr(), n=n)
            evb = ctx.blocks_cluster_dev(0, info["nblocks"], loc1.data_ptr())
            record([("cluster:" + nm, ms) for nm, ms in ctx.timing()])
            ctx.blocks_finish_dev(loc1.data_ptr(), evb, lab1.data_ptr())
            record([("finish:" + nm, ms) for nm, ms in ctx.timing()])
            del loc1, lab1
        if dist and world > 1:
            collectives.append({"op": "all_gather of 10 int64 per rank (cluster counts, quirk flags, op counters, sizes)",
                                "backend": "nccl (RCCL)", "per_step": 1})
            collectives.append({"op": "all_gather (sizes known from the first exchange) of the active noise points' "
                                      "coordinates: 16 B per point the global noise pass can reach (an eighth of the zero "
                                      "list); the pass itself runs on every rank", "backend": "nccl (RCCL)", "per_step": 1,
                                "active_noise_points": last.get("noise_active")})
            collectives.append({"op": "all_gather_into_tensor((index, label) int64 pairs, padded to the largest share)",
                                "backend": "nccl (RCCL)", "bytes_sent_per_rank_per_step": last["collective_bytes"],
                                "per_step": 1})
        extra_cfg = {"blocks": last["nblocks"], "block_range_rank0": list(last["block_range"]),
                     "kept_clusters": last["kept"], **BLOCK_DEFAULTS}
        scaling = "strong"
    else:
        pipe = None
        if mode == "exact":
            class TimedSlab:
                def __init__(self, c):
                    self.c, self.log = c, []

                def slab_begin(self, *a):
                    r = self.c.slab_begin(*a)
                    self.log += self.c.timing()
                    return r

                def slab_comps(self):
                    return self.c.slab_comps()

                def slab_finish(self, *a):
                    r = self.c.slab_finish(*a)
                    self.log += self.c.timing()
                    return r
            timed = TimedSlab(ctx)
        elif world > 1 and args.gather_labels:
            # a second communicator lets the label gather of step k overlap the clustering of step k+1; new_group is
            # itself collective, so agree on the outcome before anybody uses the group
            ok = 1
            try:
                big = dist.new_group(backend="nccl")
            except Exception as e:
                print("bench: second RCCL communicator unavailable (%s)" % e, file=sys.stderr)
                big, ok = None, 0
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                big = None
            pipe = D.SlabPipeline(ctx, n, dev, depth=2, group=None, big_group=big)

        def step(rec):
            if mode == "exact":
                timed.log = []
                r = D.exact_slabs(timed, d_coords, eps, min_pts, metric_id)
                if rec:
                    record(timed.log)
                    phase_ms.setdefault("halo_points", []).append(r["halo"])
                return r["cf"], r["dist_evals"]
            if world > 1 and pipe is not None:
                allc, ev, _ = pipe.step(d_coords, dim, eps, min_pts, metric_id)
                cf_ = allc
            elif world > 1:
                cf_, ev = D.slab_cluster(ctx, d_coords, n, dim, eps, min_pts, metric_id, d_labels, None, None)
            else:
                cf_, ev = ctx.dbscan_dev(d_coords.data_ptr(), n, dim, eps, min_pts, metric_id, 0, None,
                                         d_labels.data_ptr(), d_core.data_ptr(), d_cls.data_ptr())
            if rec:
                record(ctx.timing())
            return cf_, ev

        for _ in range(args.warmup):
            step(False)
        if pipe:
            pipe.flush()
        sync()
        barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            cf, ev = step(True)
        if pipe:
            pipe.flush()  # every step's labels are fully gathered on every rank before the clock stops
        sync()
        barrier()
        sync()
        dt = time.perf_counter() - t0
        if not (mode == "exact" or world == 1):
            cf = int(cf.sum().item())
        total_points = world * n
        scaling = "weak"
        if world > 1:
            if mode == "exact":
                collectives.append({"op": "all_gather of x-intervals, 2*eps halo strips, boundary (point, seed) pairs, "
                                          "published cluster ids, twice counters", "backend": "nccl (RCCL)",
                                    "per_step": 5})
            else:
                collectives.append({"op": "all_gather of per-rank cluster counts (8 B)", "backend": "nccl (RCCL)",
                                    "per_step": 1})
                if pipe is not None:
                    collectives.append({"op": "all_gather_into_tensor(int32 labels)", "backend": "nccl (RCCL)",
                                        "bytes_sent_per_rank_per_step": 4 * n, "per_step": 1})

    if dist:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    out = None
    if rank == 0:
        # measured device-copy ceiling: 1 GiB device-to-device copy, read + write bytes per second
        a = torch.empty(1 << 27, dtype=torch.float64, device=dev)
        b = torch.empty_like(a)
        b.copy_(a)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            b.copy_(a)
        e1.record()
        torch.cuda.synchronize()
        copy_gbs = 5 * 2 * a.numel() * 8 / (e0.elapsed_time(e1) * 1e-3) / 1e9
        del a, b
        ms_per_step = dt / args.steps * 1e3
        value = total_points / (dt / args.steps) / 1e6
        avg = {k: float(np.mean(v)) for k, v in phase_ms.items()}
        halo = avg.pop("halo_points", None)
        if mode == "blocks":
            # the per-block engine call is this rank's share of the points; begin / finish phases cover all of them
            eng = {k.split(":", 1)[1]: v for k, v in avg.items() if k.startswith("cluster:")}
            roof = roofline_of(eng, n, dim, args.metric)  # (the untimed single-device pass: all n points)
        else:
            roof = roofline_of(avg, n, dim, args.metric)
        if roof:
            roof["copy_ceiling_GBs"] = copy_gbs
            roof["frac_of_copy_ceiling"] = roof["achieved"] / copy_gbs
            if mode == "single":
                total_bpp = sum(ALGO_BYTES_PER_POINT.get(k, 0) for k in avg) + (0 if dim == 2 else 8 * len(avg))
                roof["pipeline_algorithmic_bytes_per_point"] = total_bpp
                roof["pipeline_achieved_GBs"] = total_bpp * n / (ms_per_step * 1e-3) / 1e9
        workload = {
            "single": "C4: %d-pt cloud (half uniform background, %d Gaussian blobs), one monolithic DBImproved.dbscan, "
                      "metric %s, eps %g, minPts %d" % (n, n // 50_000, args.metric, eps, min_pts),
            "blocks": "C4: ONE %d-pt cloud, block-partitioned DBImproved pipeline at the reference defaults (eps %g, "
                      "minPts %d, %d points per block) on %d GPU(s): every rank repeats the streaming passes that decide "
                      "the partition and builds, clusters and merges its own share of the blocks; the noise pass over "
                      "its gathered active points on every rank; RCCL all-gather of (index, label) pairs inside the timed "
                      "region; every rank ends with the full label array" % (n, BLOCK_DEFAULTS["eps"], BLOCK_DEFAULTS["min_pts"],
                                            BLOCK_DEFAULTS["pts_in_cell"], world),
            "exact": "C4 family: %d-pt x-slab per GPU of one cloud, exact monolithic DBImproved.dbscan result (2*eps "
                     "halo + boundary union over RCCL, labels stay on the owning rank), metric %s, eps %g, minPts %d"
                     % (n, args.metric, eps, min_pts),
            "replicas": "C4 family: an independent %d-pt cloud per GPU, cluster ids made global through an RCCL "
                        "all-gather of the per-rank counts%s, metric %s, eps %g, minPts %d"
                        % (n, " + all-gather of the int32 labels" if args.gather_labels else
                           "; labels stay on the owning rank", args.metric, eps, min_pts),
        }[mode]
        out = {
            "metric": "Mpoints/s DBSCAN+label @10M pts", "value": value, "unit": "Mpoints/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": scaling, "vs_baseline": None, "dtype": "f64",
            "data": "synthetic",
            "config": dict({"workload": workload, "mode": mode, "points_total": total_points,
                            "points_per_gpu": n if scaling == "weak" else n // world, "clusters": int(cf),
                            "resident_in_hbm": True, "collectives": collectives}, **extra_cfg),
            "roofline": roof,
            "phase_ms": {k: round(v, 4) for k, v in avg.items()},
        }
        if halo is not None:
            out["config"]["halo_points_rank0"] = halo
        if same_single is not None:
            # the job of the line above through the single-device entry points on rank 0's GPU, outside the timed
            # region: the N = 1 point of the same workload, measured in this very run
            out["single_gpu_same_workload_ms"] = same_single["ms"]
            out["speedup_vs_single_gpu_same_workload"] = same_single["ms"] / ms_per_step
            out["labels_identical_to_single_gpu_run"] = same_single["labels_identical_to_the_timed_run"]

    # ---- side measurements on rank 0 at N=1 -----------------------------------------------------------
    extras = rank == 0 and world == 1 and not args.no_extras
    cen = truth = Mt = None
    if extras and mode == "blocks":
        # the same cloud through the monolithic call, for reference beside the block pipeline
        best = None
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.dbscan_dev(d_coords.data_ptr(), n, dim, eps, min_pts, metric_id, 0, None, d_labels.data_ptr(),
                           d_core.data_ptr(), d_cls.data_ptr())
            e = time.perf_counter() - t1
            best = e if best is None else min(best, e)
        out["monolithic_dbscan"] = {"ms": best * 1e3, "Mpoints_per_s": n / best / 1e6, "eps": eps, "min_pts": min_pts}
    if extras and mode == "single" and metric_id == N.L1_2D:
        # Beside the headline (a sparse, 2^-10-quantised cloud: 0.2 expected neighbours for the background half, and no
        # pair the binary32 screen cannot decide): the same 10 M points at denser settings, an unquantised copy (the
        # exact binary64 re-test of undecided pairs runs), and the Euclidean 3-D form.  Best of three, resident.
        def best_of(ptr, d, e, mp, met, reps=3):
            b, cfx = None, 0
            for _ in range(reps):
                t1 = time.perf_counter()
                cfx, _ev = ctx.dbscan_dev(ptr, n, d, e, mp, met, 0, None, d_labels.data_ptr(), d_core.data_ptr(),
                                          d_cls.data_ptr())
                el = time.perf_counter() - t1
                b = el if b is None else min(b, el)
            return {"ms": b * 1e3, "Mpoints_per_s": n / b / 1e6, "clusters": int(cfx), "eps": e, "min_pts": mp}
        beside = {}
        dens = n / 2 / cloud["motor_extent"] ** 2  # background points per unit area
        for tag, e in (("dense_eps0.3", 0.3), ("dense_eps0.7", 0.7)):
            beside[tag] = dict(best_of(d_coords.data_ptr(), 2, e, min_pts, N.L1_2D),
                               background_neighbours_expected=2.0 * e * e * dens)
        jit = (synth.uniform01(977, 0, 2 * n).reshape(n, 2) - 0.5) * 2.0 ** -10  # real-valued: off the 2^-10 lattice
        d_unq = torch.from_numpy(coords + jit).to(dev)
        beside["unquantised"] = best_of(d_unq.data_ptr(), 2, eps, min_pts, N.L1_2D)
        del d_unq
        d_xyz3 = torch.from_numpy(cloud["xyz"]).to(dev)
        beside["L2_3D"] = best_of(d_xyz3.data_ptr(), 3, cloud["eps_l2"], min_pts, N.L2_3D)
        del d_xyz3
        out["beside_headline"] = beside
        # leave d_labels as the headline call left them (the centroid step below reads them)
        ctx.dbscan_dev(d_coords.data_ptr(), n, dim, eps, min_pts, metric_id, 0, None, d_labels.data_ptr(),
                       d_core.data_ptr(), d_cls.data_ptr())
    if extras and mode == "single":
        icp = {}
        for tag, jit, tol, rule, iters in (("c3_50_rounds_jitter0.05", 0.05, 0.0, N.STOP_SSE_DELTA, 50),
                                           ("noise_free_to_rmse_1e-4", 0.0, 1e-4, N.STOP_RMSE, 100)):
            c = synth.config_icp(nd=1_000_000, nm=100, jitter=jit)
            m = torch.from_numpy(c["model"]).to(dev)
            x = torch.from_numpy(c["data"]).to(dev)
            torch.cuda.synchronize()
            best = None
            for _ in range(5):
                t1 = time.perf_counter()
                r = ctx.icp_dev(m.data_ptr(), 100, x.data_ptr(), 1_000_000, tol, iters, rule)
                e = time.perf_counter() - t1
                best = e if best is None else min(best, e)
            icp[tag] = {"rounds": r["iters"], "ms": best * 1e3, "rounds_per_s": r["iters"] / best,
                        "rmse": r["rmse"]}
        out["icp_1M_vs_100"] = icp
        # the steps after the clustering on the same cloud: centroids of the clusters on the
        # device, ICP of the centroids against a rotated + shifted copy ("truth"), matching
        if metric_id == N.L1_2D:
            K = int(cf)
            d_xyz = torch.from_numpy(cloud["xyz"]).to(dev)
            c3 = torch.zeros(K, 3, dtype=torch.float64, device=dev)
            c2 = torch.zeros(K, 2, dtype=torch.float64, device=dev)
            cnt = torch.zeros(K, dtype=torch.int64, device=dev)
            torch.cuda.synchronize()
            t_c = None
            for _ in range(3):
                t1 = time.perf_counter()
                ctx.centroids_dev(d_xyz.data_ptr(), d_coords.data_ptr(), d_labels.data_ptr(), n, K, c3.data_ptr(),
                                  c2.data_ptr(), cnt.data_ptr())
                e = time.perf_counter() - t1
                t_c = e if t_c is None else min(t_c, e)
            cen = c3.cpu().numpy()
            Rt = synth.rotation_about((1.0, 1.0, 1.0), 0.2)
            truth = cen @ Rt.T + np.array([0.3, -0.2, 0.1])
            t_i = t_m = None
            for _ in range(3):  # host-buffer entry points: best of three (the first call allocates)
                t1 = time.perf_counter()
                ri = ctx.icp(truth, cen, 1e-9, 100, N.STOP_SSE_DELTA)
                e = time.perf_counter() - t1
                t_i = e if t_i is None else min(t_i, e)
                Mt = np.eye(4)
                Mt[:3, :3] = ri["R"]
                Mt[:3, 3] = ri["T"]
                t1 = time.perf_counter()
                mt = ctx.match(cen, truth, Mt, 0.5)
                e = time.perf_counter() - t1
                t_m = e if t_m is None else min(t_m, e)
            out["after_clustering"] = {"clusters": K, "centroids_ms": t_c * 1e3, "icp_centroids_ms": t_i * 1e3,
                                       "icp_rounds": ri["iters"], "icp_rmse": ri["rmse"], "match_ms": t_m * 1e3,
                                       "matched": int(mt["count"])}
            del d_xyz
            # the reference's production form: block-partitioned pipeline at its defaults (FrmMain.cs:1214-1544),
            # device-resident, staged API -- the N = 1 figure of --mode blocks
            local = torch.zeros(n, dtype=torch.int32, device=dev)
            best = None
            for _ in range(8):  # (the first passes size the workspace)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                info = ctx.blocks_begin(None, 0.07, 7, 200, 3, device_ptr=d_coords.data_ptr(), n=n)
                evb = ctx.blocks_cluster_dev(0, info["nblocks"], local.data_ptr())
                fin = ctx.blocks_finish_dev(local.data_ptr(), evb, d_labels.data_ptr())
                e = time.perf_counter() - t1
                best = e if best is None else min(best, e)
            out["block_pipeline"] = {"ms": best * 1e3, "Mpoints_per_s": n / best / 1e6, "blocks": info["nblocks"],
                                     "clusters": fin["cluster_amount"], "eps": 0.07, "min_pts": 7, "pts_in_cell": 200}

    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(cloud, coords, eps, min_pts, metric_id)
        if cen is not None:
            out["cpu_baseline"]["B4_matching"] = cpu_baseline_match(cen, truth, Mt)
    elif rank == 0:
        out["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
