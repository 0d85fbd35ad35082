"""Worker for tests/test_distributed_gloo.py: one rank of a world_size-N gloo group on CPU.  The compute
backend is the oracle's staged block pipeline (CPU); what is under test is the distributed driver
(vtkcloudpoint_amd/distributed.py): share ranges, padded variable-length all-gather, eval all-reduce,
slab renumbering, exact-global slabs."""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import binding as O  # noqa: E402
from vtkcloudpoint_amd import distributed as D  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    d = synth.config_cloud(60_000, seed=21)
    eps, mp, pic = 0.1, 10, 200
    res = {}
    # 1. sharded block pipeline == single-process pipeline
    be = O.StagedBlocks()
    r = D.sharded_blocks(be, d["motor"], eps, mp, pic, 3, device="cpu")
    ref = O.block_pipeline(d["motor"], eps, mp, pic, 3)
    res["blocks_labels_equal"] = bool(np.array_equal(r["labels"].numpy(), ref["labels"]))
    res["blocks_meta_equal"] = (r["kept"] == ref["kept"] and r["cluster_amount"] == ref["cluster_amount"]
                                and r["evals"] == ref["evals"] and r["del_sum"] == ref["del_sum"])
    res["block_range"] = list(r["block_range"])
    # 2. variable-length all-gather with an empty slice on one rank
    m = 1000
    full = torch.zeros(m, dtype=torch.int32)
    lo, hi = (0, 0) if rank == 0 else (0, m) if world == 2 else (0, 0)
    if world > 2:
        lo, hi = (0, 0) if rank == 0 else ((rank - 1) * m // (world - 1), rank * m // (world - 1))
    full[lo:hi] = torch.arange(lo, hi, dtype=torch.int32) + 7
    full = D.allgather_varlen(full, lo, hi, m)
    res["varlen_ok"] = bool(torch.equal(full, torch.arange(m, dtype=torch.int32) + 7))
    # 3. slab renumbering offsets
    off, total = D.exclusive_offsets(10 * (rank + 1), "cpu")
    res["offset"] = off
    res["total"] = total
    res["sum"] = D.allreduce_sum_int(rank + 1, "cpu")
    # 4. slab_cluster (the weak-scaling form bench.py --gpus N runs): oracle-backed stand-in for the context
    import ctypes as C

    class FakeCtx:
        def dbscan_dev(self, cptr, n, dim, eps, mp, metric, cf_in, cls, lptr, *a):
            coords = np.ctypeslib.as_array(C.cast(cptr, C.POINTER(C.c_double)), shape=(n, dim))
            r = O.dbscan(coords, eps, mp, metric, cf_in)
            np.ctypeslib.as_array(C.cast(lptr, C.POINTER(C.c_int32)), shape=(n,))[:] = r["labels"]
            return r["cf"], r["evals"]

    n = 5000
    slabs = [synth.config_cloud(n, seed=50 + q)["motor"] for q in range(world)]
    mine = torch.from_numpy(np.ascontiguousarray(slabs[rank]))
    lab = torch.zeros(n, dtype=torch.int32)
    gathered = torch.zeros(world * n, dtype=torch.int32)
    allc, _ = D.slab_cluster(FakeCtx(), mine, n, 2, 0.3, 5, 0, lab, gathered)
    exp, off = [], 0
    for q in range(world):
        r = O.dbscan(slabs[q], 0.3, 5, 0)
        exp.append(np.where(r["labels"] > 0, r["labels"] + off, 0))
        off += r["cf"]
    res["slab_ok"] = bool(np.array_equal(gathered.numpy(), np.concatenate(exp))) and int(allc.sum()) == off
    # 5. the pipelined form (async all-gather on a second group, double buffered)
    big = dist.new_group(backend="gloo")
    pipe = D.SlabPipeline(FakeCtx(), n, "cpu", depth=2, big_group=big)
    ok = True
    for it in range(5):
        _, _, b = pipe.step(mine, 2, 0.3, 5, 0)
    pipe.flush()
    for b in range(2):
        ok = ok and bool(np.array_equal(pipe.gathered[b].numpy(), np.concatenate(exp)))
    res["pipe_ok"] = ok
    # 6. exact_slabs (SURVEY 8e mode 2): the cloud cut into x-slabs == one monolithic DBImproved.dbscan
    cloud = synth.config_cloud(40_000, seed=33)["motor"]
    cloud = cloud[np.argsort(cloud[:, 0], kind="stable")]
    cuts = [len(cloud) * q // world for q in range(world + 1)]
    part = torch.from_numpy(np.ascontiguousarray(cloud[cuts[rank]:cuts[rank + 1]]))
    ex = D.exact_slabs(O.StagedSlab(), part, 0.3, 5, 0, cf_in=3)
    mono = O.dbscan(cloud, 0.3, 5, 0, cf_in=3)
    sl = slice(cuts[rank], cuts[rank + 1])
    res["exact_ok"] = (bool(np.array_equal(ex["labels"].numpy(), mono["labels"][sl]))
                       and bool(np.array_equal(ex["is_core"].numpy(), mono["is_key"][sl]))
                       and ex["cf"] == mono["cf"] and ex["dist_evals"] == mono["evals"])
    res["exact_halo"] = ex["halo"]
    res["exact_clusters"] = ex["cf"] - 3
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
