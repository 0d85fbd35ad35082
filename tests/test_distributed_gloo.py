"""world_size-2 (and 3) gloo runs of the multi-GPU driver on CPU: the collectives and the sharding logic
are the product's (vtkcloudpoint_amd/distributed.py); the per-rank compute is the oracle's staged pipeline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(world, tmp_path, port):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(ROOT, "tests", "_dist_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    return [json.load(open("%s.%d" % (out, r))) for r in range(world)]


@pytest.mark.parametrize("world,port", [(2, 29611), (3, 29612)])
def test_sharded_blocks_and_collectives(world, port, tmp_path, oracle):
    res = _run(world, tmp_path, port)
    for r, x in enumerate(res):
        assert x["blocks_labels_equal"] and x["blocks_meta_equal"], (r, x)
        assert x["varlen_ok"] and x["slab_ok"] and x["pipe_ok"]
        assert x["exact_ok"] and x["exact_clusters"] > 10 and 0 < x["exact_halo"] < 20_000, (r, x)
        assert x["offset"] == sum(10 * (q + 1) for q in range(r))
        assert x["total"] == sum(10 * (q + 1) for q in range(world))
        assert x["sum"] == world * (world + 1) // 2
    # contiguous, non-overlapping block ranges that cover everything
    ranges = [x["block_range"] for x in res]
    assert ranges[0][0] == 0
    for a, b in zip(ranges, ranges[1:]):
        assert a[1] == b[0]
