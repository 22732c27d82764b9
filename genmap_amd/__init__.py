"""genmap_amd -- MI355X-native (k,e)-mappability engine.

The product is libgenmap_amd.so (hand-written HIP for gfx950 behind the C ABI of include/genmap_amd.h)
plus the `genmap` host program.  This package is only the thin ctypes binding used by the tests, by
bench.py and by the one-process-per-GPU launcher; it contains no compute path of its own and raises if
the HIP library cannot be loaded or no GPU is present.
"""
from .capi import (GenmapError, Index, MapParams, default_infix_length, device_count, tuned_infix_length, lib_path, load_library,  # noqa: F401
                   device_alloc, device_free, ipc_export, ipc_open, ipc_close, push_pieces, host_pin, host_unpin)
