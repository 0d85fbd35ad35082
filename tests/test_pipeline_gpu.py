"""GPU parity of the block pipeline with EVERY stage sharded (vcp_blocks_plan_dev / build / cluster / finish_local / zero /
zcoords / pairs + the exact multi-GPU noise pass; driver: vtkcloudpoint_amd.distributed.sharded_pipeline) against the CPU
oracle's single-process pipeline (MainForm.getClusterFromMotor + StartCode + CompleteWork3, FrmMain.cs:1214-1291,
:2782-2794, :1442-1520).  The ranks are simulated in this process: one context per rank on the one GPU, the exchanges
replaced by handing each rank every rank's message (sharded_pipeline_local) -- the per-rank program is the one the
multi-process driver runs."""
import numpy as np
import pytest
import torch

from vtkcloudpoint_amd import _native as N
from vtkcloudpoint_amd import distributed as D
from vtkcloudpoint_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ranks():
    ctxs = [N.Context(0) for _ in range(4)]
    yield ctxs
    for c in ctxs:
        c.close()


def _run(ctxs, motor, eps, mp, pic, noise="gather"):
    d = torch.from_numpy(np.ascontiguousarray(motor)).cuda()
    torch.cuda.synchronize()
    res = D.sharded_pipeline_local(ctxs, d.data_ptr(), len(motor), eps, mp, pic, 3, device="cuda", noise=noise)
    torch.cuda.synchronize()
    return res


def _same(r, o, what):
    assert np.array_equal(r["labels"].cpu().numpy(), o["labels"]), what + ": labels"
    for k in ("rows", "cols", "kept", "del_sum", "cluster_amount", "evals"):
        assert r[k] == o[k], "%s: %s %r != %r" % (what, k, r[k], o[k])
    assert r["m"] == len(o["order"]), what + ": m"


def test_random_small_incl_demotion_quirks_across_shares(ranks, oracle):
    rng = np.random.default_rng(1)
    n_err = n_del = 0
    for trial in range(200):
        n = int(rng.integers(5, 400))
        motor = (rng.integers(0, 40, size=(n, 2)).astype(np.float64) * 0.25 if trial % 2
                 else rng.random((n, 2)) * 10)
        eps = float(rng.choice([0.25, 0.5, 0.75]))
        mp = int(rng.integers(1, 6))
        pic = int(rng.integers(3, 60))
        world = int(rng.integers(2, 5))
        try:
            o = oracle.block_pipeline(motor, eps, mp, pic, 3)
        except oracle.OracleError as e:
            n_err += 1
            with pytest.raises((N.VcpError, IndexError)) as ge:
                _run(ranks[:world], motor, eps, mp, pic)
            if isinstance(ge.value, N.VcpError):
                assert ge.value.code == e.code, "trial %d error code" % trial
            else:  # found by the driver from the ranks' flags: the C#'s clusForMerge[-1] (VCP_ERR_INDEX)
                assert e.code == -4, "trial %d" % trial
            continue
        res = _run(ranks[:world], motor, eps, mp, pic, noise=("gather", "slabs")[trial // 2 % 2])
        for q, r in enumerate(res):
            _same(r, o, "trial %d rank %d of %d" % (trial, q, world))
        n_del += o["del_sum"] > 0
    assert n_err > 0 and n_del > 0  # the quirk paths were really exercised


@pytest.mark.parametrize("world,noise", [(1, "gather"), (3, "gather"), (3, "slabs"), (4, "slabs")])
def test_reference_defaults_200k(ranks, oracle, world, noise):
    d = synth.config_cloud(200_000, seed=9)
    o = oracle.block_pipeline(d["motor"], 0.07, 7, 200, 3)
    res = _run(ranks[:world], d["motor"], 0.07, 7, 200, noise)
    for q, r in enumerate(res):
        _same(r, o, "rank %d of %d" % (q, world))
    ranges = [r["block_range"] for r in res]
    assert ranges[0][0] == 0 and ranges[-1][1] == o["rows"] * o["cols"]
    for a, b in zip(ranges, ranges[1:]):
        assert a[1] == b[0]
    # the noise pass ran over the part of the zero list it can reach ...
    assert 0 < res[0]["noise_active"] < res[0]["noise_points"] // 3
    if noise == "slabs":  # ... and exchanged a halo of that, not the points
        assert 0 < res[0]["noise_halo"] < res[0]["noise_active"] // 2


def test_4m_shares_equal_the_single_device_call(ranks, vcp_ctx):
    d = synth.config_cloud(4_000_000, seed=4)
    ref = vcp_ctx.dbscan_blocks(d["motor"], 0.07, 7, 200, 3)
    res = _run(ranks[:4], d["motor"], 0.07, 7, 200)
    for q, r in enumerate(res):
        assert np.array_equal(r["labels"].cpu().numpy(), ref["labels"]), "rank %d" % q
        for k in ("rows", "cols", "kept", "del_sum", "cluster_amount", "evals"):
            assert r[k] == ref[k], (q, k, r[k], ref[k])


@pytest.mark.parametrize("noise", ["gather", "slabs"])
def test_keyed_partition_shares(ranks, vcp_ctx, noise):
    """getClusterFromList (partition on X, Y; clustering on motor): the block geometry says nothing about motor distances,
    so every zero-list point is active in the noise pass."""
    d = synth.config_cloud(150_000, seed=21)
    motor = np.ascontiguousarray(d["motor"])
    key = np.ascontiguousarray(d["xyz"][:, :2])
    ref = vcp_ctx.dbscan_blocks(motor, 0.07, 7, 200, 3, key_xy=key)
    dm, dk = torch.from_numpy(motor).cuda(), torch.from_numpy(key).cuda()
    torch.cuda.synchronize()
    res = D.sharded_pipeline_local(ranks[:3], dm.data_ptr(), len(motor), 0.07, 7, 200, 3, device="cuda", d_key=dk.data_ptr(),
                                   noise=noise)
    for q, r in enumerate(res):
        assert np.array_equal(r["labels"].cpu().numpy(), ref["labels"]), "rank %d" % q
        for k in ("rows", "cols", "kept", "del_sum", "cluster_amount", "evals"):
            assert r[k] == ref[k], (q, k, r[k], ref[k])
        assert r["noise_active"] == r["noise_points"]
