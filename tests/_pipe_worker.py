"""Worker for tests/test_pipeline_gloo.py: one rank of a gloo group running the product's sharded-pipeline driver
(vtkcloudpoint_amd.distributed.sharded_pipeline: plan / cuts / build / cluster / local merge, the nine-word exchange, the
noise pass as exact_slabs, the all-gather of (index, label) pairs) on CPU tensors, with the oracle-backed stand-in
(oracle.binding.StagedPipeline) in place of the HIP context.  Every rank's result is checked against the oracle's
single-process pipeline."""
import json
import os
import sys

import numpy as np
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
    res = {"cases": []}
    rng = np.random.default_rng(3)
    cases = [(np.ascontiguousarray(synth.config_cloud(60_000, seed=21)["motor"]), 0.1, 10, 200)]
    for trial in range(40):  # small clouds on a lattice: demotions, the clusLen quirk across shares, ties
        n = int(rng.integers(30, 400))
        motor = (rng.integers(0, 40, size=(n, 2)).astype(np.float64) * 0.25 if trial % 2 else rng.random((n, 2)) * 10)
        cases.append((np.ascontiguousarray(motor), float(rng.choice([0.25, 0.5, 0.75])), int(rng.integers(1, 6)),
                      int(rng.integers(3, 60))))
    for case_no, (motor, eps, mp, pic) in enumerate(cases):
        noise = ("gather", "slabs")[case_no % 2]  # the noise pass: active points gathered, or exact slabs over the shares
        try:
            ref = O.block_pipeline(motor, eps, mp, pic, 3)
        except O.OracleError:
            ref = None
        try:
            r = D.sharded_pipeline(O.StagedPipeline(), motor.ctypes.data, len(motor), eps, mp, pic, 3, device="cpu",
                                   noise=noise)
        except IndexError:
            r = None
        if ref is None or r is None:
            res["cases"].append(dict(ok=(ref is None) == (r is None), err=True))
            continue
        ok = (bool(np.array_equal(r["labels"].numpy(), ref["labels"])) and r["kept"] == ref["kept"]
              and r["cluster_amount"] == ref["cluster_amount"] and r["evals"] == ref["evals"]
              and r["del_sum"] == ref["del_sum"] and r["m"] == len(ref["order"]))
        res["cases"].append(dict(ok=ok, err=False, dels=int(ref["del_sum"]), block_range=list(r["block_range"]),
                                 nblocks=int(r["nblocks"]), bytes=int(r["collective_bytes"])))
    with open("%s.%d" % (out_path, rank), "w") as f:
        json.dump(res, f)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
