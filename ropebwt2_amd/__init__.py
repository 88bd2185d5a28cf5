"""ropebwt2_amd -- MI355X-native multi-string BWT construction behind ropebwt2's mrope API.

This package is a thin ctypes binding over the C-ABI shared libraries built from
``ropebwt2_amd/csrc`` (see ``include/rb2_hip.h`` and ``include/mrope.h``).  All computation of
the hot path (``mr_insert_multi``, /root/reference/mrope.c:258-345) happens in hand-written HIP
kernels for gfx950; there is no Python or CPU fallback -- a missing library or GPU raises.
"""
from .build import build_all, lib_path  # noqa: F401
from .hipbwt import HipBwt, MultiBwt, load_hip_lib, K_NAMES  # noqa: F401

__all__ = ["HipBwt", "MultiBwt", "load_hip_lib", "build_all", "lib_path", "K_NAMES"]
