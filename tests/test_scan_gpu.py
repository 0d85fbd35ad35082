"""The library's device prefix scan (single-pass, decoupled look-back) against numpy, through the C-ABI self-test
entry: sizes round the tile edges, one tile .. thousands of tiles, unaligned pointers, in place, sum and running max,
and many calls on one context (the descriptors are told apart by a per-call generation number, never cleared)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _check(ctx, a, op, offset=0, inplace=False):
    n = len(a)
    dev = torch.zeros(n + 8, dtype=torch.int32, device="cuda")
    dev[offset:offset + n] = torch.from_numpy(a.view(np.int32)).cuda()
    out = dev if inplace else torch.full((n + 8,), -1, dtype=torch.int32, device="cuda")
    torch.cuda.synchronize()
    tot = ctx.selftest_scan_dev(dev.data_ptr() + 4 * offset, out.data_ptr() + 4 * offset, n, op)
    got = out[offset:offset + n].cpu().numpy().view(np.uint32)
    if op == 0:
        inc = np.cumsum(a, dtype=np.uint64).astype(np.uint32)  # modulo 2^32 like the device
    else:
        inc = np.maximum.accumulate(a)
    want = np.concatenate([[0], inc[:-1]]).astype(np.uint32) if n else inc
    assert np.array_equal(got, want)
    assert tot == (int(inc[-1]) if n else 0)
    if not inplace and n:
        assert int(out[offset + n].item()) == -1  # nothing written past the end


def test_scan_sizes_and_ops(vcp_ctx):
    rng = np.random.default_rng(11)
    for n in (1, 2, 3, 5, 63, 64, 255, 1023, 1024, 1025, 8191, 8192, 8193, 16384, 100_003, 8192 * 70 + 1, 3_000_001):
        a = rng.integers(0, 1000, n, dtype=np.uint32)
        for op in (0, 1):
            _check(vcp_ctx, a, op)
    a = rng.integers(0, 2**32, 1_000_000, dtype=np.uint64).astype(np.uint32)  # sums wrap, maxima do not
    _check(vcp_ctx, a, 0)
    _check(vcp_ctx, a, 1)
    a = rng.integers(0, 50, 777_777, dtype=np.uint32)
    for off in (1, 2, 3):  # pointers that are not 16-byte aligned take the scalar loads
        _check(vcp_ctx, a, 0, offset=off)
    _check(vcp_ctx, a, 0, inplace=True)
    _check(vcp_ctx, a, 1, offset=1, inplace=True)


def test_scan_many_calls_and_large(vcp_ctx):
    rng = np.random.default_rng(12)
    a = rng.integers(0, 9, 500_000, dtype=np.uint32)
    for _ in range(200):  # generations 1..200 of the same descriptors
        _check(vcp_ctx, a, 0)
    big = rng.integers(0, 3, 60_000_000, dtype=np.uint32)  # 7325 tiles: far more than are resident at once
    _check(vcp_ctx, big, 0)
    _check(vcp_ctx, big, 1)
    _check(vcp_ctx, a, 0)  # and a small one again after the descriptor array has grown
