"""fnft_amd -- MI355X-native fast nonlinear Fourier transform (fnft_nsev continuous spectrum).

The product is the C-ABI shared library fnft_amd/lib/libfnft_amd.so (hand-written HIP for gfx950,
see include/fnft_amd.h); this package only builds it (fnft_amd.build), binds it (fnft_amd.capi)
and holds the multi-GPU sharding helpers (fnft_amd.sharding).  No CPU fallback exists.
"""
from . import capi  # noqa: F401

__all__ = ["capi"]
