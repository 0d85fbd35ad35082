"""The C-ABI library on a CPU-only box: it loads, exports every symbol include/vcp.h declares, and fails
loudly (no CPU fallback) when there is no GPU.  No compute calls here."""
import ctypes as C
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    with open(os.path.join(ROOT, "include", "vcp.h")) as f:
        src = f.read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(vcp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from vtkcloudpoint_amd import _native
    lib = _native.lib()
    decl = _declared()
    assert len(decl) >= 25
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, missing
    assert sorted(_native.SYMBOLS) == decl  # the Python binding's list tracks the header
    assert lib.vcp_version() == 1


def test_no_silent_cpu_fallback():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from vtkcloudpoint_amd import _native
    with pytest.raises(_native.VcpError) as e:
        _native.Context(0)
    assert e.value.code == -6  # VCP_ERR_NO_DEVICE
    assert "no CPU fallback" in str(e.value)


def test_product_does_not_import_the_oracle():
    """oracle/ is test infrastructure: nothing under vtkcloudpoint_amd/ may reference it."""
    pkg = os.path.join(ROOT, "vtkcloudpoint_amd")
    for dp, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                with open(os.path.join(dp, fn), errors="ignore") as f:
                    text = f.read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", text, flags=re.M), fn
                assert "vcp_oracle" not in text and "libvcp_oracle" not in text, fn
    with open(os.path.join(ROOT, "vtkcloudpoint_amd", "csrc", "Makefile")) as f:
        assert "oracle" not in f.read()


def test_horn_step_cold_start_on_a_nan_or_garbage_basis():
    """The Horn step of vcp_icp (host run of the same __host__ __device__ source, no device needed): a warm-start basis
    full of NaN, or one that is not orthonormal, must take the cold-start path -- fmax() would have dropped the NaN."""
    import numpy as np
    from vtkcloudpoint_amd import _native
    lib = _native.lib()
    rng = np.random.default_rng(5)
    P = rng.uniform(-3, 3, (500, 3))
    a = 0.4
    Rz = np.array([[np.cos(a), -np.sin(a), 0], [np.sin(a), np.cos(a), 0], [0, 0, 1.0]])
    Y = P @ Rz.T + np.array([0.5, -1.0, 2.0])
    sums = np.concatenate([P.sum(0), Y.sum(0), (P[:, :, None] * Y[:, None, :]).sum(0).reshape(9), [0.0]])

    def solve(V, use):
        R, T = np.zeros(9), np.zeros(3)
        Vb = None if V is None else np.ascontiguousarray(V, np.float64).reshape(16).copy()
        rc = lib.vcp_selftest_horn(sums.ctypes.data_as(C.c_void_p), C.c_int64(len(P)),
                                   None if Vb is None else Vb.ctypes.data_as(C.c_void_p), C.c_int(use),
                                   R.ctypes.data_as(C.c_void_p), T.ctypes.data_as(C.c_void_p))
        assert rc == 1
        return R.reshape(3, 3), T, Vb

    R0, T0, _ = solve(None, 0)
    assert np.allclose(R0, Rz, atol=1e-12) and np.allclose(T0, [0.5, -1.0, 2.0], atol=1e-12)
    Rc, Tc, Vc = solve(np.eye(4), 1)             # identity basis = the cold start, bit for bit
    assert np.array_equal(Rc, R0) and np.array_equal(Tc, T0)
    for bad in (np.full((4, 4), np.nan), rng.uniform(-1, 1, (4, 4)), np.eye(4) * 2.0):
        Rb, Tb, Vb = solve(bad, 1)
        assert np.array_equal(Rb, R0) and np.array_equal(Tb, T0)
        assert np.array_equal(Vb, Vc)             # the basis stored back is the cold start's
    Rw, Tw, _ = solve(Vc.reshape(4, 4), 1)        # a good basis: warm start, same answer to rounding
    assert np.allclose(Rw, R0, atol=1e-12) and np.allclose(Tw, T0, atol=1e-12)


def test_block_range_plan_for_2_3_8_ranks():
    """vcp_blocks_share_plan (host arithmetic behind vcp_blocks_share and vcp_dbscan_blocks_multi): contiguous ranges that
    cover every block once, cut at the first block whose first position reaches m * r / world, balanced on points."""
    import numpy as np
    from vtkcloudpoint_amd import _native
    rng = np.random.default_rng(11)
    for trial in range(200):
        nb = int(rng.integers(1, 400))
        sizes = rng.integers(0, 50, nb)
        if trial % 5 == 0:
            sizes[rng.integers(0, nb)] = 5000  # one block that dwarfs the rest
        if trial % 7 == 0:
            sizes[:] = 0
        bs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.uint32)
        m = int(bs[-1])
        for world in (1, 2, 3, 8):
            cuts = _native.blocks_share_plan(bs, world)
            assert cuts[0] == 0 and cuts[-1] == nb and np.all(np.diff(cuts) >= 0)
            for r in range(1, world):
                target = (m * r) // world
                want = int(np.searchsorted(bs[:nb], target, side="left"))
                assert cuts[r] == max(want, cuts[r - 1])
            if m > 0 and sizes.max() > 0:
                share = np.diff(bs[cuts].astype(np.int64))
                assert share.sum() == m
                assert share.max() <= m / world + sizes.max()  # never worse than one block over the fair share
