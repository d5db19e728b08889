"""ogc_amd — MI355X-native (gfx950) implementation of OGC's point-cloud hot path.

Layout:
  csrc/            hand-written HIP kernels + the C ABI (include/ogc_ops.h) -> libogc_ops.so
  _lib.py          ctypes binding of that ABI
  pointnet2_cuda   drop-in for the reference's native extension module of the same name
  pointnet2/       operator API (autograd Functions, QueryAndGroup) — reference pointnet2/pointnet2.py
  utils/ models/ losses/ oa_icp   host-side layers of the path restated on these operators

There is no CPU implementation in this package: operators raise on non-HIP tensors, and on a
machine with a GPU the import itself fails if libogc_ops.so has not been built.
"""
import sys

import torch

from . import _lib

__version__ = "0.1.0"

if torch.cuda.is_available():
    # fail loudly on a GPU box if the native library is missing (no silent eager fallback)
    _lib.load()


def install_drop_in():
    """Register ``pointnet2_cuda`` (and ``pointnet2.pointnet2``) in ``sys.modules`` so that code written
    against the reference's module names imports this implementation."""
    from . import pointnet2_cuda
    from . import pointnet2 as _pkg
    from .pointnet2 import pointnet2 as _api
    sys.modules.setdefault("pointnet2_cuda", pointnet2_cuda)
    sys.modules.setdefault("pointnet2", _pkg)
    sys.modules.setdefault("pointnet2.pointnet2", _api)
    return pointnet2_cuda
