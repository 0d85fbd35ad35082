"""Worker for tests/test_bench_path_gloo.py: one rank of a gloo group running bench.py's OWN multi-GPU step
(bench.blocks_workload -> distributed.sharded_pipeline: every stage sharded) on CPU tensors, with the oracle-backed
stand-in for the HIP context.  Checks every rank's final labels against the single-process pipeline."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from oracle import binding as O  # noqa: E402
from vtkcloudpoint_amd import synth  # noqa: E402


def main():
    out_path = sys.argv[1]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    motor = synth.config_cloud(80_000, seed=77)["motor"]
    motor = np.ascontiguousarray(motor)
    dt, r = bench.blocks_workload(O.StagedPipeline(), motor, "cpu", steps=2, warmup=1, barrier=dist.barrier)
    bd = bench.BLOCK_DEFAULTS
    ref = O.block_pipeline(motor, bd["eps"], bd["min_pts"], bd["pts_in_cell"], bd["small_max"])
    res = dict(labels_equal=bool(np.array_equal(r["labels"].numpy(), ref["labels"])),
               meta_equal=(r["kept"] == ref["kept"] and r["cluster_amount"] == ref["cluster_amount"]
                           and r["evals"] == ref["evals"] and r["del_sum"] == ref["del_sum"]),
               block_range=list(r["block_range"]), nblocks=r["nblocks"], bytes=r["collective_bytes"], dt=dt,
               clusters=ref["cluster_amount"])
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
