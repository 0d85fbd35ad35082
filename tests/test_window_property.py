"""The candidate window of the all-pairs block kernel (csrc/blocks.hip: brute_window) rests on one inequality: the
partition's sort key d = max(x - x_Min, y - y_Min) (FrmMain.cs:1231-1232) is 1-Lipschitz in the maximum norm, so two points
that DBImproved.getDisP (BC/DBImproved.cs:14-25) puts within eps of each other have keys within eps -- up to the rounding
of the binary64 subtractions, which the kernel covers with a relative slack of 2^-40.  Checked here on the CPU with the
kernel's own expressions (numpy binary64 = the device's arithmetic for +, -, max), on the kinds of clouds the GPU tests
use: random, lattice (distances exactly eps), large offsets (coarse spacing of the doubles), duplicates."""
import numpy as np


def _keys(m, x_min, y_min):
    return np.maximum(m[:, 0] - x_min, m[:, 1] - y_min)


def _check(m, eps):
    x_min, y_min = m[:, 0].min(), m[:, 1].min()
    d = _keys(m, x_min, y_min)
    # every pair the reference's predicate accepts ...
    dx = np.abs(m[:, None, 0] - m[None, :, 0])
    dy = np.abs(m[:, None, 1] - m[None, :, 1])
    near = (dx + dy) <= eps
    i, j = np.nonzero(near)
    # ... lies inside the window the kernel derives from the smaller / larger key of a group (here: of the point itself,
    # the tightest case): lo_v = d_i - eps - (|d_i| + eps) 2^-40 <= d_j <= d_i + eps + (|d_i| + eps) 2^-40
    lo_v = d[i] - eps - (np.abs(d[i]) + eps) * 2.0 ** -40
    hi_v = d[i] + eps + (np.abs(d[i]) + eps) * 2.0 ** -40
    assert np.all(d[j] >= lo_v) and np.all(d[j] <= hi_v)
    return len(i)


def test_neighbours_have_close_keys():
    rng = np.random.default_rng(5)
    pairs = 0
    for trial in range(60):
        n = int(rng.integers(50, 700))
        m = rng.random((n, 2)) * float(rng.choice([1.0, 10.0, 300.0]))
        if trial % 3 == 0:
            m = np.round(m * 4.0) / 4.0                      # lattice: many pairs at exactly eps
        if trial % 4 == 1:
            m += np.array([3.0e8, -7.0e8])                   # spacing of the doubles ~ 6e-8 / 1e-7
        if trial % 5 == 2:
            m = np.concatenate([m, m[: n // 3]])            # duplicates: equal keys
        if trial % 7 == 3:
            m *= 1.0e-9                                     # tiny coordinates
        scale = np.abs(m).max() if trial % 7 == 3 else 1.0
        eps = float(rng.choice([0.25, 0.07, 0.5, 1.0])) * (scale if trial % 7 == 3 else 1.0)
        pairs += _check(np.ascontiguousarray(m), eps)
    assert pairs > 100_000


def test_extreme_eps_disables_the_window():
    """eps = inf, NaN or negative: the kernel's guard `lo_v <= d_first and hi_v >= d_last` decides -- inf keeps a window that
    spans everything, NaN / negative switch the shortcut off (every position is a candidate; the predicate then accepts
    nothing for NaN / negative eps, as the reference's `<=` does)."""
    d = np.array([0.0, 1.0, 5.0])
    for eps in (np.inf, np.nan, -1.0):
        with np.errstate(invalid="ignore"):
            lo_v = d[0] - eps - (abs(d[0]) + eps) * 2.0 ** -40
            hi_v = d[-1] + eps + (abs(d[-1]) + eps) * 2.0 ** -40
            ok = (lo_v <= d[0]) and (hi_v >= d[-1])
        if eps == np.inf:
            assert ok and lo_v == -np.inf and hi_v == np.inf
        else:
            assert not ok
