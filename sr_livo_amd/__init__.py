"""sr_livo_amd -- MI355X (gfx950) implementation of SR-LIVO's LIO scan-matching hot path.

The product is the C-ABI shared library ``libsrlivo_hip.so`` (include/srlivo_hip.h) plus the C++ host
mirror of the reference classes (sr_livo_amd/csrc/host/).  This package is the thin ctypes layer the
parity tests and bench.py use to drive that library; it contains no arithmetic of the path and no
CPU fallback: every compute entry point fails loudly when the HIP extension or a GPU is missing.
"""
from .capi import (  # noqa: F401
    LIB_PATH, SrlError, IcpOpts, Frame, NormalEq, Timing, Context, Lio,
    load_library, library_symbols, declared_symbols, default_opts, shard_range, shard_budget,
    grid_sampling, heap_topk, PinnedArray, comm_backend_info, comm_set_library, tr1_order,
)
