"""graph_framework_amd — MI355X (gfx950) backend for graph_framework work items.

Host-side mirror of the reference's backend-context / workflow interface over
the C ABI of include/gf_hip.h.  There is no CPU path in this package.
"""
from . import _lib  # noqa: F401
from .backend import Context, Kernel, GfHipError, key_of, generate_source  # noqa: F401
