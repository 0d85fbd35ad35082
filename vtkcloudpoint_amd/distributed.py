"""Multi-GPU driver: one process per GPU, torch.distributed for the collectives (backend "nccl" is RCCL on
ROCm, over xGMI; "gloo" on CPU for the tests).  The data path has exactly one exchange step.

Two shardings of the reference's block-partitioned clustering (FrmMain.cs:1262-1285 partition, :1358 one
DBImproved per block, :1442-1520 sequential merge):

* sharded_blocks   every rank holds the whole cloud and computes the same partition; the per-block DBSCAN
                   is split into contiguous block ranges balanced on point count; the block-local labels
                   are all-gathered in block-major order (variable-length: padded to the largest slice);
                   every rank then runs CompleteWork3 on the full label array.  Total work is fixed
                   (strong scaling of one job); result = the single-GPU vcp_dbscan_blocks, bit for bit.
* slab_cluster     every rank owns its own slab of a larger cloud (weak scaling): slabs are clustered
                   independently, cluster ids are made global with an exclusive scan of the per-rank
                   cluster counts, labels are all-gathered.

And the exact form (SURVEY.md 8e mode 2):

* exact_slabs      every rank owns a part of ONE cloud (the global list is the rank-major concatenation);
                   the result is what a single DBImproved.dbscan over the whole list returns -- labels,
                   core flags, cluster count and iritatorNum -- bit for bit.  Exchange: x-intervals, a
                   2*eps halo of coordinates, the (point, local component) pairs of boundary points, and
                   the ids of the clusters that cross a boundary.  All of it is O(boundary), not O(n).
"""
import numpy as np
import torch
import torch.distributed as dist


def _world(group=None):
    if not dist.is_available() or not dist.is_initialized():
        return 0, 1
    return dist.get_rank(group), dist.get_world_size(group)


def allgather_varlen(local_full, pos_lo, pos_hi, m, group=None):
    """local_full: 1-D tensor [m]; this rank has filled [pos_lo, pos_hi).  Returns the tensor with every
    rank's slice filled in.  Slices are padded to the largest one (RCCL all-gather wants equal sizes)."""
    rank, world = _world(group)
    if world == 1:
        return local_full
    dev = local_full.device
    bounds = torch.tensor([pos_lo, pos_hi], dtype=torch.int64, device=dev)
    allb = torch.empty(2 * world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allb, bounds, group=group)
    allb = allb.cpu().tolist()
    width = max(allb[2 * r + 1] - allb[2 * r] for r in range(world))
    if width == 0:
        return local_full
    send = torch.zeros(width, dtype=local_full.dtype, device=dev)
    send[: pos_hi - pos_lo] = local_full[pos_lo:pos_hi]
    recv = torch.empty(world * width, dtype=local_full.dtype, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    for r in range(world):
        lo, hi = allb[2 * r], allb[2 * r + 1]
        if r != rank and hi > lo:
            local_full[lo:hi] = recv[r * width: r * width + (hi - lo)]
    return local_full


def exclusive_offsets(count, device, group=None):
    """Exclusive scan over ranks of a per-rank integer (cluster counts).  Returns (offset, total)."""
    rank, world = _world(group)
    if world == 1:
        return 0, int(count)
    mine = torch.tensor([int(count)], dtype=torch.int64, device=device)
    allc = torch.empty(world, dtype=torch.int64, device=device)
    dist.all_gather_into_tensor(allc, mine, group=group)
    allc = allc.cpu()
    return int(allc[:rank].sum()), int(allc.sum())


def allreduce_sum_int(v, device, group=None):
    rank, world = _world(group)
    if world == 1:
        return int(v)
    t = torch.tensor([int(v)], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return int(t.item())


def allgather_known(local_full, ranges, extra, group=None, force=False):
    """The label exchange of sharded_blocks as ONE collective.  local_full: int32 tensor [m]; ranges[r] = the
    [lo, hi) slice rank r has filled -- every rank derives all of them from the identical partition, so no size
    exchange (and no host round trip) precedes the data.  `extra` (an int64 per rank, the op counter) rides in two
    trailing int32 words of each rank's message.  Slices are padded to the largest one (RCCL all-gathers want equal
    sizes).  Returns (local_full with every slice filled in, sum of all ranks' extra, bytes this rank sent)."""
    rank, world = _world(group)
    if world == 1 and not (force and dist.is_available() and dist.is_initialized()):
        return local_full, int(extra), 0
    dev = local_full.device
    width = max(hi - lo for lo, hi in ranges) + 2
    lo, hi = ranges[rank]
    send = torch.empty(width, dtype=torch.int32, device=dev)
    send[: hi - lo] = local_full[lo:hi]
    send[width - 2:] = torch.tensor([int(extra)], dtype=torch.int64).view(torch.int32).to(dev)
    recv = torch.empty(world * width, dtype=torch.int32, device=dev)
    dist.all_gather_into_tensor(recv, send, group=group)
    for r in range(world):
        rlo, rhi = ranges[r]
        if r != rank and rhi > rlo:
            local_full[rlo:rhi] = recv[r * width: r * width + (rhi - rlo)]
    tails = recv.view(world, width)[:, width - 2:].contiguous().cpu()  # the one host read of the step
    total = int(tails.view(torch.int64).sum().item())
    return local_full, total, width * 4


def sharded_blocks(backend, motor, eps, min_pts, pts_in_cell, small_max=3, group=None, device="cuda",
                   motor_dev_ptr=None, local=None, labels=None, force_collective=False, n=None):
    """Run the block pipeline with the per-block step sharded over the ranks of `group`.

    backend: object with blocks_begin / blocks_share / blocks_cluster_dev / blocks_finish_dev (a
    vtkcloudpoint_amd._native.Context; the CPU tests pass an oracle-backed stand-in with the same methods
    that works on CPU tensors).  local [>= m] / labels [>= n]: optional preallocated int32 work / output tensors
    (a timed loop passes them so that no allocation or fill sits in the step).  motor may be None when motor_dev_ptr
    names a device-resident cloud; n is then required.
    Returns dict(labels [n] int32 tensor, local [m], kept, cluster_amount, ..., collective_bytes).
    """
    rank, world = _world(group)
    if motor is not None:
        if n is not None and int(n) != len(motor):
            raise ValueError("n = %d disagrees with len(motor) = %d" % (int(n), len(motor)))
        n = len(motor)
    elif n is None:
        raise ValueError("a device-only cloud (motor=None, motor_dev_ptr) needs its point count n")
    n = int(n)
    if motor_dev_ptr is not None:
        info = backend.blocks_begin(None, eps, min_pts, pts_in_cell, small_max, device_ptr=motor_dev_ptr, n=n)
    else:
        info = backend.blocks_begin(motor, eps, min_pts, pts_in_cell, small_max)
    m = info["m"]
    shares = [backend.blocks_share(r, world) for r in range(world)]  # same partition, same cuts on every rank
    lo, hi, plo, phi = shares[rank]
    if local is None:
        local = torch.zeros(max(m, 1), dtype=torch.int32, device=device)
    evals = backend.blocks_cluster_dev(lo, hi, local.data_ptr())
    # the library call above has returned = its stream is drained; the collective runs on torch's streams and the
    # host read of the op counters inside allgather_known drains those before CompleteWork3 is launched
    local, evals, sent = allgather_known(local, [(s[2], s[3]) for s in shares], evals, group, force_collective)
    if labels is None:
        labels = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    out = backend.blocks_finish_dev(local.data_ptr(), evals, labels.data_ptr())
    out.update(labels=labels[:n], local=local[:m], rows=info["rows"], cols=info["cols"], nblocks=info["nblocks"],
               block_range=(lo, hi), m=m, collective_bytes=sent)
    return out


def slab_cluster(ctx, d_coords, n, dim, eps, min_pts, metric, d_labels, gathered=None, group=None):
    """Weak-scaling form: cluster this rank's slab (device tensor d_coords [n, dim]), renumber globally,
    all-gather the labels into `gathered` [world*n] if given.  The renumbering stays on the device (no host
    round trip between the clustering and the all-gather).  Returns (per-rank cluster counts tensor, evals)."""
    rank, world = _world(group)
    if d_labels.is_cuda:  # the library writes on its own stream: whatever torch still has queued on these buffers
        torch.cuda.current_stream(d_labels.device).synchronize()  # (the previous step's renumbering) must be done
    cf, ev = ctx.dbscan_dev(d_coords.data_ptr(), n, dim, eps, min_pts, metric, 0, None, d_labels.data_ptr())
    dev = d_labels.device
    mine = torch.tensor([cf], dtype=torch.int64, device=dev)
    if world == 1:
        return mine, ev
    allc = torch.empty(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(allc, mine, group=group)
    if rank > 0:
        off = allc[:rank].sum().to(torch.int32)
        d_labels.add_(torch.where(d_labels > 0, off, torch.zeros((), dtype=torch.int32, device=dev)))
    if gathered is not None:
        dist.all_gather_into_tensor(gathered, d_labels, group=group)
    return allc, ev


class SlabPipeline:
    """slab_cluster with the label all-gather of step k overlapped with the clustering of step k+1.

    The big all-gather runs on its own process group (own RCCL communicator and stream) with async_op=True and
    double-buffered label / gather tensors; the tiny cluster-count gather stays on the default group, so it
    never queues behind a 40 MB-per-rank transfer.  flush() waits for everything still in flight."""

    def __init__(self, ctx, n, device, depth=2, group=None, big_group=None):
        self.ctx, self.n, self.depth = ctx, n, depth
        self.group, self.big_group = group, big_group
        self.rank, self.world = _world(group)
        self.labels = [torch.zeros(n, dtype=torch.int32, device=device) for _ in range(depth)]
        self.gathered = ([torch.zeros(self.world * n, dtype=torch.int32, device=device) for _ in range(depth)]
                         if self.world > 1 else [None] * depth)
        self.pending = [None] * depth
        self.k = 0

    def step(self, d_coords, dim, eps, min_pts, metric):
        b = self.k % self.depth
        self.k += 1
        if self.pending[b] is not None:
            self.pending[b].wait()  # the buffer pair is free again
            self.pending[b] = None
            if self.labels[b].is_cuda:  # wait() only orders torch's stream; the library writes on its own
                torch.cuda.current_stream(self.labels[b].device).synchronize()
        lab = self.labels[b]
        allc, ev = slab_cluster(self.ctx, d_coords, self.n, dim, eps, min_pts, metric, lab, None, self.group)
        if self.world > 1:
            self.pending[b] = dist.all_gather_into_tensor(self.gathered[b], lab, group=self.big_group, async_op=True)
        return allc, ev, b

    def flush(self):
        for b in range(self.depth):
            if self.pending[b] is not None:
                self.pending[b].wait()
                self.pending[b] = None


# ---------------------------------------------------------------------------------------------------
# exact_slabs: one DBImproved.dbscan (BaseClass/DBImproved.cs:91-114) over a cloud spread over the ranks
# ---------------------------------------------------------------------------------------------------
def gather_rows(t, group=None, fixed=False):
    """All-gather of a tensor whose first dimension differs per rank; returns the list of every rank's
    tensor (padded to the largest for the collective: RCCL/gloo all-gathers want equal sizes).  fixed=True:
    every rank sends the same shape, no size exchange; fixed = a list of every rank's row count: the sizes are
    known to everybody already (no size exchange either)."""
    rank, world = _world(group)
    if world == 1:
        return [t]
    dev = t.device
    tail = tuple(t.shape[1:])
    if fixed is True:
        recv = t.new_empty((world * t.shape[0],) + tail)
        dist.all_gather_into_tensor(recv, t.contiguous(), group=group)
        return list(recv.split(t.shape[0]))
    if isinstance(fixed, (list, tuple)):
        allc = [int(c) for c in fixed]
    else:
        cnt = torch.tensor([t.shape[0]], dtype=torch.int64, device=dev)
        allc = torch.empty(world, dtype=torch.int64, device=dev)
        dist.all_gather_into_tensor(allc, cnt, group=group)
        allc = allc.cpu().tolist()
    width = max(allc)
    if width == 0:
        return [t.new_zeros((0,) + tail) for _ in range(world)]
    send = t.new_zeros((width,) + tail)
    send[: t.shape[0]] = t
    recv = t.new_empty((world * width,) + tail)
    dist.all_gather_into_tensor(recv, send.contiguous(), group=group)
    return [recv[r * width: r * width + allc[r]] for r in range(world)]


def _union_min(keys, reps):
    """Classes of `reps` values linked by sharing a key.  Returns (unique reps ascending, smallest rep of the
    class of each).  Host-side union-find over the boundary pairs only."""
    u, inv = np.unique(reps, return_inverse=True)
    parent = list(range(len(u)))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a

    order = np.argsort(keys, kind="stable")
    ks, nodes = keys[order], inv[order]
    same = np.nonzero(ks[1:] == ks[:-1])[0]
    for t in same.tolist():
        a, b = find(int(nodes[t])), find(int(nodes[t + 1]))
        if a != b:  # nodes are ranks in the ascending unique list: the smaller index is the smaller rep
            if a < b:
                parent[b] = a
            else:
                parent[a] = b
    fin = np.array([u[find(a)] for a in range(len(u))], dtype=u.dtype) if len(u) else u
    return u, fin


def _exact_slabs_steps(backend, coords, eps, min_pts, metric, cf_in, rank, world, lean=False, extra=0.0):
    """The per-rank program of exact_slabs as a generator: it yields the tensor it contributes to each
    exchange step and is sent back the list of all ranks' tensors (see exact_slabs / exact_slabs_local)."""
    if not (eps >= 0.0) or eps == float("inf"):
        raise ValueError("exact_slabs needs a finite eps >= 0")
    n, dim = int(coords.shape[0]), int(coords.shape[1])
    dev = coords.device
    f64, i64 = torch.float64, torch.int64
    x = coords[:, 0]
    # lean (the noise pass of sharded_pipeline): the caller vouches for finite coordinates and wants labels only -- no pass
    # over the coordinates, no core / classed flags
    if n and not lean and not bool(torch.isfinite(coords).all()):
        raise ValueError("exact_slabs needs finite coordinates")

    # 1. x-interval and size of every rank's part; global position of my first point
    # (`extra`: one number of the caller's that rides along in this first exchange; all ranks' come back as out["extras"])
    if n:
        info = torch.stack([x.min(), x.max(), torch.tensor(float(n), dtype=f64, device=dev),
                            torch.tensor(float(extra), dtype=f64, device=dev)])
    else:
        info = torch.tensor([float("inf"), float("-inf"), 0.0, float(extra)], dtype=f64, device=dev)
    allinfo = torch.stack((yield info.reshape(1, 4), True)).reshape(world, 4).cpu().numpy()
    sizes = allinfo[:, 2].astype(np.int64)
    gofs = int(sizes[:rank].sum())
    n_total = int(sizes.sum())
    if n_total >= 2 ** 31:
        raise ValueError("exact_slabs: more than 2^31 points in total")
    # margins: |dx| <= eps for every neighbour pair (each |dx| term is rounded once, the sum is monotone), so
    # eps plus a relative 2^-20 and a few ulps of the largest coordinate covers every neighbour
    amax = float(np.abs(allinfo[:, :2][np.isfinite(allinfo[:, :2])]).max()) if n_total else 0.0
    m1 = eps * (1.0 + 2.0 ** -20) + 8.0 * np.spacing(amax) + 1e-300
    m2 = 2.0 * m1
    lo_me, hi_me = float(allinfo[rank, 0]), float(allinfo[rank, 1])

    # 2. my points that lie within 2*eps of another rank's interval (coords + global position, one f64 tensor)
    mask = torch.zeros(n, dtype=torch.bool, device=dev)
    for s in range(world):
        if s != rank and sizes[s] > 0:
            mask |= (x >= float(allinfo[s, 0]) - m2) & (x <= float(allinfo[s, 1]) + m2)
    sidx = torch.nonzero(mask).reshape(-1)
    strip = torch.cat([coords[sidx], (sidx + gofs).to(f64).reshape(-1, 1)], dim=1)
    strips = yield strip, False

    # 3. halo = the other ranks' strip points within 2*eps of my interval; those within eps ("inner") have
    #    their whole neighbourhood here, so their core flag is exact and they may extend clusters
    parts = [strips[s] for s in range(world) if s != rank and strips[s].shape[0]]
    if parts and n:
        h = torch.cat(parts, dim=0)
        hx = h[:, 0]
        h = h[(hx >= lo_me - m2) & (hx <= hi_me + m2)]
    else:
        h = coords.new_zeros((0, dim + 1))
    nh = int(h.shape[0])
    hx = h[:, 0]
    inner = (hx >= lo_me - m1) & (hx <= hi_me + m1)
    out = dict(n=n, halo=nh, gofs=gofs)
    labels = (torch.empty if lean and n else torch.zeros)(max(n + nh, 1), dtype=torch.int32, device=dev)
    is_core = None if lean else torch.zeros(max(n + nh, 1), dtype=torch.uint8, device=dev)
    is_classed = None if lean else torch.zeros(max(n + nh, 1), dtype=torch.uint8, device=dev)
    if n:
        nl = n + nh
        local = torch.cat([coords, h[:, :dim]], dim=0).contiguous() if nh else coords.contiguous()
        ordv = torch.arange(gofs, gofs + n, dtype=torch.int32, device=dev)
        noexp = None
        if nh:
            ordv = torch.cat([ordv, h[:, dim].to(torch.int32)])
            noexp = torch.cat([torch.zeros(n, dtype=torch.uint8, device=dev), (~inner).to(torch.uint8)])
        rep = torch.empty(nl, dtype=torch.int32, device=dev)

        # 4. grid, core flags, local components (HIP: vcp_slab_begin).  The library runs on its own stream:
        #    what torch queued above must have landed first.
        if dev.type == "cuda":
            torch.cuda.current_stream(dev).synchronize()
        backend.slab_begin(local.data_ptr(), nl, dim, metric, float(eps), int(min_pts),
                           None if noexp is None else noexp.data_ptr(), ordv.data_ptr(), rep.data_ptr(),
                           None if is_core is None else is_core.data_ptr())
        comps = np.asarray(backend.slab_comps(), dtype=np.uint32).astype(np.int64)  # ascending seeds
        # 5. (point, local seed) for every expanding point another rank also sees, or that I see of theirs
        if nh or sidx.numel():
            cand = torch.cat([sidx, n + torch.nonzero(inner).reshape(-1)])
            r = rep[cand]
            keep = r != -1
            pairs = torch.stack([ordv[cand][keep].to(i64), r[keep].to(i64)], dim=1)
        else:  # nobody else's interval comes near (a single rank): no boundary
            pairs = torch.zeros((0, 2), dtype=i64, device=dev)
    else:
        comps = np.zeros(0, np.int64)
        pairs = torch.zeros((0, 2), dtype=i64, device=dev)
    allpairs = torch.cat((yield pairs, False), dim=0).cpu().numpy()

    # 6. components that cross a boundary: local seeds that share a point are one cluster; its seed is the
    #    smallest of them (every member is owned by somebody, whose local seed is <= the member)
    u, fin = _union_min(allpairs[:, 0], allpairs[:, 1])
    if len(u):
        at = np.minimum(np.searchsorted(u, comps), len(u) - 1)
        inb = u[at] == comps
        final = np.where(inb, fin[at], comps)
    else:
        inb = np.zeros(len(comps), bool)
        final = comps.copy()

    # 7. canonical numbering: cluster ids follow the global seeds in increasing order.  Each rank counts the
    #    seeds it owns; ids of boundary clusters are published by the owner of the seed.
    mine = (final == comps) & (comps >= gofs) & (comps < gofs + n)
    myseeds = comps[mine]
    pubsel = np.nonzero(inb[mine])[0]
    msg = np.empty((1 + len(pubsel), 2), np.int64)  # row 0: (my seed count, -1); then (seed, its rank among mine)
    msg[0] = (len(myseeds), -1)
    msg[1:, 0] = myseeds[pubsel]
    msg[1:, 1] = pubsel
    allmsg = [m.cpu().numpy() for m in (yield torch.from_numpy(msg).to(dev), False)]
    counts = np.array([int(m[0, 0]) for m in allmsg], np.int64)
    bases = int(cf_in) + np.concatenate([[0], np.cumsum(counts)[:-1]])
    k_total = int(counts.sum())
    myids = int(bases[rank]) + 1 + np.arange(len(myseeds), dtype=np.int64)
    pub_seed = np.concatenate([m[1:, 0] for m in allmsg])
    pub_id = np.concatenate([bases[q] + 1 + m[1:, 1] for q, m in enumerate(allmsg)])

    if n:
        # 8. global cluster of every local component, then the border rule and labels (HIP: vcp_slab_finish)
        seeds_known = np.concatenate([myseeds, pub_seed])
        ids_known = np.concatenate([myids, pub_id])
        o = np.argsort(seeds_known, kind="stable")
        seeds_known, ids_known = seeds_known[o], ids_known[o]
        at = np.minimum(np.searchsorted(seeds_known, final), max(len(seeds_known) - 1, 0))
        if len(final) and (len(seeds_known) == 0 or not np.array_equal(seeds_known[at], final)):
            raise RuntimeError("exact_slabs: a local component has no published global cluster")
        gid = ids_known[at] if len(final) else np.zeros(0, np.int64)
        tab_gid, map_k = np.unique(gid, return_inverse=True)
        tab_seed = np.zeros(len(tab_gid), np.int64)
        tab_seed[map_k] = final
        twice = backend.slab_finish(comps.astype(np.uint32), map_k.astype(np.uint32), tab_gid.astype(np.int32),
                                    tab_seed.astype(np.uint32), gofs, n, labels.data_ptr(),
                                    None if is_classed is None else is_classed.data_ptr())
    else:
        twice = 0
    # 9. iritatorNum of the monolithic call: n_total * (queried points + seeds + border points queried twice)
    alltw = torch.cat((yield torch.tensor([[twice]], dtype=i64, device=dev), True), dim=0).cpu().numpy().reshape(-1)
    out.update(labels=labels[:n], is_core=None if is_core is None else is_core[:n],
               is_classed=None if is_classed is None else is_classed[:n], cf=int(cf_in) + k_total,
               dist_evals=n_total * (n_total + k_total + int(alltw.sum())), n_total=n_total, clusters=k_total,
               twice=int(alltw.sum()), extras=allinfo[:, 3].copy(),
               boundary_pairs=int(allpairs.shape[0]))
    return out


def exact_slabs(backend, coords, eps, min_pts, metric=0, cf_in=0, group=None):
    """DBImproved.dbscan over the rank-major concatenation of every rank's `coords` ([n_r, dim] float64 tensor
    on this rank's device); each rank gets the labels / core flags of its own points.  `backend` provides
    slab_begin / slab_comps / slab_finish (a vtkcloudpoint_amd._native.Context; the CPU tests pass the oracle's
    stand-in).  Points may be distributed arbitrarily; the exchange stays small when ranks own x-slabs."""
    rank, world = _world(group)
    gen = _exact_slabs_steps(backend, coords, eps, min_pts, metric, cf_in, rank, world)
    try:
        msg, fixed = next(gen)
        while True:
            msg, fixed = gen.send(gather_rows(msg, group, fixed))
    except StopIteration as e:
        return e.value


def exact_slabs_local(backends, parts, eps, min_pts, metric=0, cf_in=0):
    """exact_slabs with every rank simulated in this process (one backend / context per part): the same
    per-rank program, the exchange replaced by handing each rank the list of all contributions."""
    world = len(parts)
    gens = [_exact_slabs_steps(backends[r], parts[r], eps, min_pts, metric, cf_in, r, world) for r in range(world)]
    msgs = [next(g)[0] for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(list(msgs))[0])
            except StopIteration as e:
                results[r] = e.value
                nxt.append(None)
        msgs = nxt
    return results


# ---------------------------------------------------------------------------------------------------
# sharded_pipeline: the block pipeline with EVERY stage sharded (include/vcp.h: vcp_blocks_plan_dev ...)
# ---------------------------------------------------------------------------------------------------
def _pipeline_steps(backend, d_motor, n, eps, min_pts, pts_in_cell, small_max, rank, world, device, labels, d_key=None,
                    noise="gather"):
    """The per-rank program as a generator (like _exact_slabs_steps): yields what it contributes to each exchange and
    is sent back every rank's contribution.

    Every rank reads the whole cloud (device pointer d_motor, n points) but repeats only the streaming passes that decide
    the partition (bounds, first block, block of every point: MainForm.getClusterFromMotor, FrmMain.cs:1214-1258); it
    then builds, clusters (StartCode :2782-2794) and merges (CompleteWork3 :1442-1504) its own share of the blocks.
    Exchanges: (1) ten words per rank -- cluster counts for the global renumbering (:1460-1504), who asks whom to zero a
    last entry (the clusLen quirk :1461-1465 / :1485-1488 across a share boundary), op counters, sizes (also of the
    zero list's active part, so that no exchange needs a size exchange in front of it: three collectives per step); (2) the global
    noise pass (:1507-1516) over the ACTIVE points of the ranks' zero lists (the eighth of the noise a cluster of that pass
    can reach, csrc/blocks.hip: k_zero_flag) -- noise="gather": ONE all-gather of their coordinates (16 bytes per active
    point) and the pass itself on every rank (0.3 ms at 10 M points: cheaper than any exchange pattern); noise="slabs":
    exact_slabs over the shares (bands in y), O(boundary), for clouds whose active set is too large to repeat;
    (3) one all-gather of the (index, label) pairs, 8 bytes per point, after which every rank scatters the full label
    array."""
    i64 = torch.int64
    info = backend.blocks_plan(d_motor, n, eps, min_pts, pts_in_cell, small_max, d_key)
    cuts = backend.blocks_plan_cuts(world)
    sh = backend.blocks_build(cuts[rank], cuts[rank + 1])
    m, n_loc = sh["m"], sh["n_loc"]
    local = torch.empty(max(m, 1), dtype=torch.int32, device=device)  # (every position of the share is written)
    evals = backend.blocks_cluster_dev(sh["block_lo"], sh["block_hi"], local.data_ptr()) if m > 0 else 0
    st = backend.blocks_finish_local(local.data_ptr())
    # the zero list as it stands (nobody has asked this share to zero its last entry yet): its sizes travel with the counters,
    # so that the exchange of the active points needs no size exchange of its own
    z_all, z = backend.blocks_finish_zero(False)
    mine = torch.tensor([[st["clusters"], st["kept"], st["err"], st["req"], st["nonempty"], st["last_nonzero"], evals, m,
                          n_loc, z]], dtype=i64, device=device)
    allst = torch.cat((yield mine, True), dim=0).cpu().numpy().astype(np.int64)
    if int(allst[:, 2].sum()) != 0:
        raise IndexError("clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)")
    # a share whose first non-empty block demotes its first cluster zeroes the last entry of the nearest earlier share
    # that has a non-empty block (the C# walks clusForMerge backwards, :1485-1488); none there: the C# throws
    zero_me = False
    for q in range(world):
        if allst[q, 3]:
            prev = [p for p in range(q) if allst[p, 4]]
            if not prev:
                raise IndexError("clusForMerge index -1 while demoting the first cluster (FrmMain.cs:1487)")
            zero_me = zero_me or prev[-1] == rank
    kept_all = allst[:, 1]
    kept_off, kept_total = int(kept_all[:rank].sum()), int(kept_all.sum())
    clusters_total = int(allst[:, 0].sum())
    z_sent = z
    if zero_me:  # (rare) the entry zeroed joins the zero list and its active part: one point more
        z_all, z = backend.blocks_finish_zero(True)
    if noise == "gather":
        # header row (zero-list size, active points) + the active points' coordinates, every rank's to every rank; row
        # counts known from the first exchange (+ the header, + one row in case a last entry was zeroed since)
        zc = torch.empty((z_sent + 2, 2), dtype=torch.float64, device=device)  # (rows behind the active points: never read)
        backend.blocks_finish_zcoords(zc[1:].data_ptr(), False)
        zc[0, 0] = float(z_all)
        zc[0, 1] = float(z)
        allz = yield zc, [int(c) + 2 for c in allst[:, 9]]
        # one read-back: every rank's (zero list, active points)
        heads = torch.stack([t[0] for t in allz]).cpu().numpy() if world > 1 else np.array([[float(z_all), float(z)]])
        z_total = int(round(float(heads[:, 0].sum())))
        acts = [int(round(float(v))) for v in heads[:, 1]]
        a_off = sum(acts[:rank])
        coords = torch.cat([t[1:1 + a] for t, a in zip(allz, acts)], dim=0).contiguous() if world > 1 else zc[1:1 + z]
        a_total = int(coords.shape[0])
        zlab_all = torch.empty(max(a_total, 1), dtype=torch.int32, device=device)
        cf, ev = kept_total, 0
        if a_total:
            if torch.device(device).type == "cuda":
                torch.cuda.current_stream(torch.device(device)).synchronize()  # coords came from torch's stream
            cf, ev = backend.dbscan_dev(coords.data_ptr(), a_total, 2, eps, min_pts, 0, kept_total, None,
                                        zlab_all.data_ptr())
        k = cf - kept_total
        twice = ev // a_total - a_total - k if a_total else 0  # engine: A x (A + K + twice)
        ex = dict(labels=zlab_all[a_off: a_off + z], cf=cf, n_total=a_total, halo=0)
        noise_evals = z_total * (z_total + k + twice)
    else:
        ex = None
    zc = torch.empty((max(z, 1), 2), dtype=torch.float64, device=device) if ex is None else None
    if ex is None:
        backend.blocks_finish_zcoords(zc.data_ptr(), True)  # as (y, x): the shares are bands in y
    # FrmMain.cs:1507-1516: ONE DBImproved over all noise with cf preset -- the zero lists of the shares, in rank order,
    # are the C#'s zero list; the pass runs over its ACTIVE points only (csrc/blocks.hip: k_zero_flag -- everybody else
    # provably keeps 0, and the order among the active points is the zero list's); exact_slabs gives every rank the labels
    # of its own part
    if ex is None:
        ex = yield from _exact_slabs_steps(backend, zc[:z], eps, min_pts, 0, kept_total, rank, world, lean=True,
                                           extra=float(z_all))
        z_total = int(round(float(ex["extras"].sum())))
        noise_evals = z_total * (z_total + ex["clusters"] + ex["twice"])  # iritatorNum of the pass over the whole zero list
    zlab = ex["labels"].to(torch.int32).contiguous()
    pairs = torch.empty(max(n_loc, 1), dtype=i64, device=device)
    if torch.device(device).type == "cuda":
        torch.cuda.current_stream(torch.device(device)).synchronize()  # zlab came from torch's stream
    backend.blocks_finish_pairs(kept_off, zlab.data_ptr() if z > 0 else None, pairs.data_ptr())
    allp = yield pairs[:n_loc], [int(c) for c in allst[:, 8]]
    if labels is None:
        labels = torch.zeros(max(n, 1), dtype=torch.int32, device=device)
    for q in range(world):
        cnt = int(allst[q, 8])
        if cnt:
            pq = allp[q].contiguous()
            backend.scatter_pairs(pq.data_ptr(), cnt, n, labels.data_ptr())
    return dict(labels=labels[:n], rows=info["rows"], cols=info["cols"], nblocks=info["nblocks"],
                block_range=(sh["block_lo"], sh["block_hi"]), m=int(allst[:, 7].sum()), m_local=m, kept=kept_total,
                del_sum=clusters_total - kept_total, cluster_amount=ex["cf"],
                evals=int(allst[:, 6].sum()) + noise_evals, noise_points=z_total, noise_active=ex["n_total"],
                noise_halo=ex["halo"], collective_bytes=8 * max(int(c) for c in allst[:, 8]))


def sharded_pipeline(backend, d_motor, n, eps, min_pts, pts_in_cell, small_max=3, group=None, device="cuda",
                     labels=None, d_key=None, noise="gather"):
    """The block-partitioned pipeline (= vcp_dbscan_blocks, bit for bit) with partition, per-block clustering AND merge
    sharded over the ranks of `group`.  d_motor: device (or, for the CPU stand-in, host) address of the whole cloud
    [n, 2] float64 on every rank.  Returns the dict of _pipeline_steps; every rank gets the full label array."""
    rank, world = _world(group)
    gen = _pipeline_steps(backend, d_motor, n, eps, min_pts, pts_in_cell, small_max, rank, world, device, labels, d_key,
                          noise)
    try:
        msg, fixed = next(gen)
        while True:
            msg, fixed = gen.send(gather_rows(msg, group, fixed))
    except StopIteration as e:
        return e.value


def sharded_pipeline_local(backends, d_motor, n, eps, min_pts, pts_in_cell, small_max=3, device="cuda", d_key=None,
                           noise="gather"):
    """sharded_pipeline with every rank simulated in this process (one backend / context per rank, e.g. several contexts
    on one GPU): the same per-rank program, the exchange replaced by handing each rank the list of all contributions."""
    world = len(backends)
    gens = [_pipeline_steps(backends[r], d_motor, n, eps, min_pts, pts_in_cell, small_max, r, world, device, None, d_key,
                            noise) for r in range(world)]
    msgs = [next(g)[0] for g in gens]
    results = [None] * world
    while any(r is None for r in results):
        nxt = []
        for r, g in enumerate(gens):
            try:
                nxt.append(g.send(list(msgs))[0])
            except StopIteration as e:
                results[r] = e.value
                nxt.append(None)
        msgs = nxt
    return results
