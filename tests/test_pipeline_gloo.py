"""The block pipeline with every stage sharded (what `bench.py --gpus N` times for N > 1), world_size 2 and 3 over gloo on
CPU tensors: the collectives and the per-rank program are the product's (vtkcloudpoint_amd/distributed.py:
sharded_pipeline), the per-rank compute is the oracle-backed stand-in; result = the oracle's single-process pipeline."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world,port", [(2, 29651), (3, 29652)])
def test_sharded_pipeline_over_gloo(world, port, tmp_path, oracle):
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "_pipe_worker.py"), out]
    subprocess.run(cmd, check=True, env=env, timeout=600, cwd=ROOT)
    res = [json.load(open("%s.%d" % (out, r))) for r in range(world)]
    for r, x in enumerate(res):
        assert len(x["cases"]) == 41
        for k, c in enumerate(x["cases"]):
            assert c["ok"], (r, k, c)
    good = [c for c in res[0]["cases"] if not c["err"]]
    assert any(c["dels"] > 0 for c in good) and any(c["err"] for c in res[0]["cases"])  # the quirk paths were exercised
    # contiguous shares that cover every block
    for k in range(len(res[0]["cases"])):
        if res[0]["cases"][k]["err"]:
            continue
        ranges = [x["cases"][k]["block_range"] for x in res]
        assert ranges[0][0] == 0 and ranges[-1][1] == res[0]["cases"][k]["nblocks"]
        for a, b in zip(ranges, ranges[1:]):
            assert a[1] == b[0]
