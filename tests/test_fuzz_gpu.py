"""The randomised GPU-vs-oracle parity sweep (tests/fuzz_parity.py) as ONE bounded, fixed-seed pass: every entry point
(DBSCAN with all metrics / isClassed inputs / cf presets, the block pipeline and its keyed twin, nearest neighbour of
ICP and of the matching, centroids, weighted centroids, centroid merge), bit-exact against the CPU oracle, with a bound
on the GPU time of every DBSCAN call (the open-ended sweep found two performance cliffs in round 2: eps = 0 with far
outliers, and a cloud that sits inside one eps-ball)."""
import pytest

pytestmark = pytest.mark.gpu


def test_bounded_fixed_seed_sweep(vcp_ctx, oracle):
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(os.path.dirname(__file__), "fuzz_parity.py"))
    F = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(F)
    F.ctx = vcp_ctx  # the session's context (run() leaves a context it did not create open)
    try:
        # clouds of up to 10^5.6 = 400 k points keep the CPU oracle in the tens of milliseconds per case; a DBSCAN call of
        # that size takes ~3 ms on the GPU through the host-buffer ABI -- 0.25 s + 1 us per point is a cliff, not noise
        done = F.run(budget=45.0, seed=20261004, max_log_n=5.6, gpu_bound=lambda n: 0.25 + 1e-6 * n, quiet=True, nn_log=4.0)
    finally:
        F.ctx = None
    assert done["dbscan"] >= 20 and done["blocks"] >= 3 and done.get("nn", 0) >= 3 and done.get("tools", 0) >= 1, done
    assert done.get("db", 0) >= 1, done
