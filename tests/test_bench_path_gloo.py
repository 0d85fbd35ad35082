"""The code path `bench.py --gpus N` times at N > 1 (ONE cloud; every rank builds, clusters and merges its own share of
the blocks; noise pass as exact slabs; one all-gather of (index, label) pairs), run with world_size 2 and 3 over gloo on
CPU tensors."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29631), (3, 29632)])
def test_bench_blocks_step_over_gloo(world, port, tmp_path, oracle):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_bench_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    for r, x in enumerate(res):
        assert x["labels_equal"] and x["meta_equal"], (r, x)
        assert x["clusters"] > 10 and x["bytes"] > 0
    ranges = [x["block_range"] for x in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == res[0]["nblocks"]
    for a, b in zip(ranges, ranges[1:]):
        assert a[1] == b[0]


def test_bench_refuses_mismatched_world(tmp_path):
    """--gpus must agree with WORLD_SIZE: a 1-rank run can never be recorded as an N-GPU line."""
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "1"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert p.returncode == 2 and "WORLD_SIZE" in p.stderr
