"""leon_amd -- MI355X-native implementation of Leon's DNA-sequence encode path.

The product is the C-ABI shared library `leon_amd/lib/libleon_dna.so` (include/leon_dna.h) built from
hand-written HIP kernels for gfx950 (leon_amd/csrc).  This package is the thin Python binding used by the
tests and bench.py; it has no CPU path and raises if the library is missing.
"""
from .capi import LeonDnaError, DnaEncodeContext, lib_path, load_library  # noqa: F401
from .build import build_library  # noqa: F401
from .shard import block_range, merge_block_tables  # noqa: F401

__all__ = ["LeonDnaError", "DnaEncodeContext", "lib_path", "load_library", "build_library"]
