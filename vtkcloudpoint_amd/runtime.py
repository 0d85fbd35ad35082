"""Process-wide default context (one GPU, LOCAL_RANK aware).  There is no CPU fallback: creating the
context raises when libvcp.so or the GPU is missing."""
import os

from . import _native

_default = None


def default_context():
    global _default
    if _default is None:
        _default = _native.Context(int(os.environ.get("LOCAL_RANK", "0")))
    return _default


def set_default_context(ctx):
    global _default
    _default = ctx
