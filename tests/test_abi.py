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
