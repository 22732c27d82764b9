"""ctypes binding of include/genmap_amd.h (one class per opaque handle, same names and argument meaning).

No fallback: if libgenmap_amd.so is missing this module raises, and every compute call needs a HIP device
(the library returns GM_ERR_NO_DEVICE otherwise, surfaced as GenmapError)."""
import ctypes as C
import os
from pathlib import Path

import numpy as np

_ROOT = Path(__file__).resolve().parent
_LIB = None
_LIB_NAME = None


class GenmapError(RuntimeError):
    def __init__(self, status, text):
        super().__init__(f"genmap_amd error {status}: {text}")
        self.status = status


class MapParams(C.Structure):
    """struct gm_map_params"""
    _fields_ = [("K", C.c_uint32), ("E", C.c_uint32), ("overlap", C.c_int32), ("infix", C.c_int32),
                ("revcompl", C.c_int32), ("value_bits", C.c_int32), ("exclude_pseudo", C.c_int32),
                ("flags", C.c_int32), ("kmer_begin", C.c_uint64), ("kmer_end", C.c_uint64),
                ("chunk_blocks", C.c_uint32), ("chunk_index", C.c_uint32), ("chunk_stride", C.c_uint32), ("reserved1", C.c_uint32),
                ("whole_begin", C.c_uint64), ("whole_end", C.c_uint64)]


class IndexInfo(C.Structure):
    _fields_ = [("n_rows", C.c_uint64), ("text_len", C.c_uint64), ("n_seq", C.c_uint32), ("sampling", C.c_uint32),
                ("alphabet_size", C.c_uint32), ("block_bytes", C.c_uint32), ("device_bytes", C.c_uint64),
                ("device", C.c_int32), ("row_bits", C.c_uint32), ("verify_records", C.c_uint32)]


class MapStats(C.Structure):
    _fields_ = [("kmers", C.c_uint64), ("roots", C.c_uint64), ("node_steps", C.c_uint64), ("rank_lines", C.c_uint64),
                ("detail", C.c_uint64 * 48), ("search_ms", C.c_double), ("total_ms", C.c_double)]


class Runs(C.Structure):
    """struct gm_runs"""
    _fields_ = [("n_runs", C.c_uint64), ("start", C.POINTER(C.c_uint64)), ("length", C.POINTER(C.c_uint64)), ("value", C.POINTER(C.c_uint16))]


class Locations(C.Structure):
    """struct gm_locations"""
    _fields_ = [("pos_begin", C.c_uint64), ("n_positions", C.c_uint64), ("plus_off", C.POINTER(C.c_uint64)),
                ("minus_off", C.POINTER(C.c_uint64)), ("plus", C.POINTER(C.c_uint64)), ("minus", C.POINTER(C.c_uint64))]


MAP_FLAG_RANGE = 1
MAP_FLAG_PIECE = 2    # GM_MAP_FLAG_PIECE: the call is one launch of the share (whole_begin, whole_end)
WIDE_ROWS = 0x10000   # GM_BLOCK_WIDE_ROWS: OR into block_bytes to force 64-bit rows

EXPORTS = ["gm_map_files", "gm_device_alloc", "gm_device_free", "gm_ipc_export", "gm_ipc_open", "gm_ipc_close", "gm_push_pieces", "gm_map_shard", "gm_host_pin", "gm_host_unpin", "gm_index_set_tuning", "gm_index_sync", "gm_map_kernel_times", "gm_map_runs", "gm_runs_free", "gm_tuned_infix_length", "gm_tuned_infix_length_locating", "gm_index_export_sa", "gm_locate", "gm_locations_free", "gm_status_string", "gm_last_error", "gm_device_count", "gm_index_build", "gm_index_import", "gm_index_import_sampled", "gm_index_export_sa_sampled",
           "gm_index_export_bwt", "gm_index_get_info", "gm_index_free", "gm_map", "gm_map_device",
           "gm_last_map_stats", "gm_default_infix_length"]


def lib_path(profiling=False):
    name = "libgenmap_amd_prof.so" if profiling else "libgenmap_amd.so"
    return _ROOT / "lib" / name


def load_library(profiling=False):
    """Load the in-tree HIP library.  Raises if it has not been built (python __graft_entry__.py build)."""
    global _LIB, _LIB_NAME
    path = lib_path(profiling)
    if _LIB is not None and _LIB_NAME == str(path):
        return _LIB
    if not path.exists():
        raise GenmapError(-1, f"{path} not built; run `python -c 'import __graft_entry__ as g; g.build()'`")
    # One HIP runtime per process: PyTorch ships its own libamdhip64 (same SONAME as /opt/rocm's).  If this library
    # pulled in /opt/rocm's copy first, a later `import torch` would be bound to it and fail to see the GPU; loading
    # torch first makes both share torch's runtime, so torch tensors' device pointers are valid here.
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the binding itself
        pass
    lib = C.CDLL(str(path))
    vp = C.c_void_p
    lib.gm_status_string.restype = C.c_char_p
    lib.gm_status_string.argtypes = [C.c_int]
    lib.gm_last_error.restype = C.c_char_p
    lib.gm_device_count.restype = C.c_int
    lib.gm_index_build.restype = C.c_int
    lib.gm_index_build.argtypes = [vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.gm_index_import.restype = C.c_int
    lib.gm_index_import.argtypes = [vp, vp, vp, C.c_uint32, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.gm_index_export_bwt.restype = C.c_int
    lib.gm_index_export_bwt.argtypes = [vp, vp, vp]
    lib.gm_index_import_sampled.restype = C.c_int
    lib.gm_index_import_sampled.argtypes = [vp, vp, vp, vp, C.c_uint32, C.c_uint64, vp, vp, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.POINTER(vp)]
    lib.gm_index_export_sa_sampled.restype = C.c_int
    lib.gm_index_export_sa_sampled.argtypes = [vp, vp, vp, C.c_uint32, C.POINTER(C.c_uint64)]
    lib.gm_index_export_sa.restype = C.c_int
    lib.gm_index_export_sa.argtypes = [vp, vp, C.c_uint32]
    lib.gm_index_get_info.restype = C.c_int
    lib.gm_index_get_info.argtypes = [vp, C.POINTER(IndexInfo)]
    lib.gm_index_free.argtypes = [vp]
    lib.gm_map.restype = C.c_int
    lib.gm_map.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(MapParams), vp, C.c_uint64, vp, vp]
    lib.gm_map_files.restype = C.c_int
    lib.gm_map_files.argtypes = [vp, C.c_uint32, vp, vp, C.POINTER(MapParams), vp, vp]
    lib.gm_map_shard.restype = C.c_int
    lib.gm_map_shard.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(MapParams), vp, C.c_uint64, vp, vp]
    lib.gm_host_pin.restype = C.c_int
    lib.gm_host_pin.argtypes = [vp, C.c_uint64]
    lib.gm_host_unpin.restype = C.c_int
    lib.gm_host_unpin.argtypes = [vp]
    lib.gm_device_alloc.restype = C.c_int
    lib.gm_device_alloc.argtypes = [C.c_int, C.c_uint64, C.POINTER(vp)]
    lib.gm_device_free.restype = C.c_int
    lib.gm_device_free.argtypes = [C.c_int, vp]
    lib.gm_ipc_export.restype = C.c_int
    lib.gm_ipc_export.argtypes = [C.c_int, vp, C.c_char_p]
    lib.gm_ipc_open.restype = C.c_int
    lib.gm_ipc_open.argtypes = [C.c_int, C.c_char_p, C.POINTER(vp)]
    lib.gm_ipc_close.restype = C.c_int
    lib.gm_ipc_close.argtypes = [C.c_int, vp]
    lib.gm_push_pieces.restype = C.c_int
    lib.gm_push_pieces.argtypes = [C.c_int, vp, vp, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, C.c_uint64, vp]
    lib.gm_map_device.restype = C.c_int
    lib.gm_map_device.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(MapParams), vp, C.c_uint64, vp, vp, vp]
    lib.gm_map_runs.restype = C.c_int
    lib.gm_map_runs.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(MapParams), vp, C.c_uint64, vp, C.POINTER(C.POINTER(Runs))]
    lib.gm_runs_free.argtypes = [C.POINTER(Runs)]
    lib.gm_locate.restype = C.c_int
    lib.gm_locate.argtypes = [vp, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.POINTER(MapParams), vp, C.c_uint64, C.POINTER(C.POINTER(Locations))]
    lib.gm_locations_free.argtypes = [C.POINTER(Locations)]
    lib.gm_last_map_stats.restype = C.c_int
    lib.gm_last_map_stats.argtypes = [vp, C.POINTER(MapStats)]
    lib.gm_index_set_tuning.restype = C.c_int
    lib.gm_index_set_tuning.argtypes = [vp, C.c_char_p, C.c_int64]
    lib.gm_index_sync.restype = C.c_int
    lib.gm_index_sync.argtypes = [vp]
    lib.gm_map_kernel_times.restype = C.c_int
    lib.gm_map_kernel_times.argtypes = [vp, C.POINTER(C.c_double), C.c_uint32, C.POINTER(C.c_uint32)]
    lib.gm_tuned_infix_length.restype = C.c_uint32
    lib.gm_tuned_infix_length.argtypes = [C.c_uint32, C.c_uint32]
    lib.gm_tuned_infix_length_locating.restype = C.c_uint32
    lib.gm_tuned_infix_length_locating.argtypes = [C.c_uint32, C.c_uint32]
    lib.gm_default_infix_length.restype = C.c_uint32
    lib.gm_default_infix_length.argtypes = [C.c_uint32, C.c_uint32, C.c_int32]
    _LIB, _LIB_NAME = lib, str(path)
    return lib


def _check(lib, rc):
    if rc != 0:
        detail = lib.gm_last_error().decode() or lib.gm_status_string(rc).decode()
        raise GenmapError(rc, f"{lib.gm_status_string(rc).decode()} [{detail}]")


def device_count():
    return int(load_library().gm_device_count())


def default_infix_length(K, E, xo=None):
    return int(load_library().gm_default_infix_length(K, E, -1 if xo is None else xo))


def tuned_infix_length(K, E, locating=False):
    """common-infix length the library schedules with by default; locating: --exclude-pseudo / gm_locate calls"""
    lib = load_library()
    return int(lib.gm_tuned_infix_length_locating(K, E) if locating else lib.gm_tuned_infix_length(K, E))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def device_alloc(device, nbytes):
    """gm_device_alloc: a zeroed base allocation on `device` (exportable through HIP IPC); returns the device pointer"""
    lib = load_library()
    p = C.c_void_p()
    _check(lib, lib.gm_device_alloc(device, nbytes, C.byref(p)))
    return p.value


def device_free(device, ptr):
    lib = load_library()
    _check(lib, lib.gm_device_free(device, C.c_void_p(ptr)))


def host_pin(arr):
    """gm_host_pin: page-lock a numpy array so that device-to-host copies into it are asynchronous DMA transfers"""
    lib = load_library()
    _check(lib, lib.gm_host_pin(_ptr(arr), arr.nbytes))


def host_unpin(arr):
    lib = load_library()
    _check(lib, lib.gm_host_unpin(_ptr(arr)))


def ipc_export(device, ptr):
    lib = load_library()
    buf = C.create_string_buffer(64)
    _check(lib, lib.gm_ipc_export(device, C.c_void_p(ptr), buf))
    return bytes(buf.raw)


def ipc_open(device, handle):
    lib = load_library()
    p = C.c_void_p()
    _check(lib, lib.gm_ipc_open(device, C.c_char_p(handle) if False else C.create_string_buffer(handle, 64), C.byref(p)))
    return p.value


def ipc_close(device, ptr):
    lib = load_library()
    _check(lib, lib.gm_ipc_close(device, C.c_void_p(ptr)))


def push_pieces(device, dst, src, first_byte, pitch_bytes, piece_bytes, n_rows, last_piece_bytes=0, stream=None):
    """gm_push_pieces: this shard's chunks of src -> the same offsets of dst (device-to-device, one async copy per chunk)"""
    lib = load_library()
    _check(lib, lib.gm_push_pieces(device, C.c_void_p(dst), C.c_void_p(src), first_byte, pitch_bytes, piece_bytes, n_rows, last_piece_bytes, C.c_void_p(stream or 0)))


class _OwnedArray(np.ndarray):
    """Read-only ndarray view of memory owned by the library.  `_owner` frees that memory when the last view is collected; it is
    handed on explicitly to every view / slice / reshape (__array_finalize__), not left to the .base chain.  Whoever keeps a small part
    of a large result for long should .copy() it: a view keeps the whole result alive; pickling or sending to another process copies."""
    _owner = None

    def __array_finalize__(self, obj):
        if obj is not None:
            self._owner = getattr(obj, "_owner", None)


class _LocationsOwner:
    def __init__(self, lib, L):
        self._lib, self._L = lib, L

    def __del__(self):
        if self._L is not None:
            self._lib.gm_locations_free(self._L)
            self._L = None


class Index:
    """gm_index handle: the bidirectional FM index resident in one GPU's HBM."""

    def __init__(self, handle, lib, codes, seq_len):
        self._h, self._lib = handle, lib
        self.seq_len = np.ascontiguousarray(seq_len, dtype=np.uint64)
        self.cum = np.concatenate([[0], np.cumsum(self.seq_len)]).astype(np.uint64)
        self.text_len = int(self.cum[-1])

    @classmethod
    def build(cls, codes, seq_len, sampling=0, block_bytes=0, device=0, profiling=False):
        """gm_index_build: suffix-sort and pack both indexes on the GPU."""
        lib = load_library(profiling)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        sl = np.ascontiguousarray(seq_len, dtype=np.uint64)
        h = C.c_void_p()
        _check(lib, lib.gm_index_build(_ptr(codes), _ptr(sl), len(sl), sampling, block_bytes, device, C.byref(h)))
        return cls(h, lib, codes, sl)

    @classmethod
    def from_bwt(cls, bwt_fwd, bwt_rev, codes, seq_len, sa_fwd=None, sampling=0, block_bytes=0, device=0, profiling=False):
        """gm_index_import"""
        lib = load_library(profiling)
        bf = np.ascontiguousarray(bwt_fwd, dtype=np.uint8)
        br = np.ascontiguousarray(bwt_rev, dtype=np.uint8)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        sl = np.ascontiguousarray(seq_len, dtype=np.uint64)
        wide = bool(block_bytes & WIDE_ROWS) or len(bf) >= 0xFFFFFFFF
        sa = None if sa_fwd is None else np.ascontiguousarray(sa_fwd, dtype=np.uint64 if wide else np.uint32)
        h = C.c_void_p()
        _check(lib, lib.gm_index_import(_ptr(bf), _ptr(br), _ptr(sa), 0 if sa is None else sa.itemsize, _ptr(codes), _ptr(sl), len(sl), sampling, block_bytes, device, C.byref(h)))
        return cls(h, lib, codes, sl)

    @classmethod
    def from_sampled(cls, bwt_fwd, bwt_rev, mark_words, samples, codes, seq_len, sampling, block_bytes=0, device=0, profiling=False):
        """gm_index_import_sampled: the sampled suffix array as an index directory holds it"""
        lib = load_library(profiling)
        bf = np.ascontiguousarray(bwt_fwd, dtype=np.uint8)
        br = np.ascontiguousarray(bwt_rev, dtype=np.uint8)
        mk = np.ascontiguousarray(mark_words, dtype=np.uint32)
        wide = bool(block_bytes & WIDE_ROWS) or len(bf) >= 0xFFFFFFFF
        sm = np.ascontiguousarray(samples, dtype=np.uint64 if wide else np.uint32)
        codes = np.ascontiguousarray(codes, dtype=np.uint8)
        sl = np.ascontiguousarray(seq_len, dtype=np.uint64)
        if len(mk) != (len(bf) + 31) // 32:
            raise ValueError("mark_words: one bit per row")
        h = C.c_void_p()
        _check(lib, lib.gm_index_import_sampled(_ptr(bf), _ptr(br), _ptr(mk), _ptr(sm), sm.itemsize, len(sm), _ptr(codes), _ptr(sl), len(sl), sampling, block_bytes, device, C.byref(h)))
        return cls(h, lib, codes, sl)

    def close(self):
        if self._h:
            self._lib.gm_index_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def info(self):
        i = IndexInfo()
        _check(self._lib, self._lib.gm_index_get_info(self._h, C.byref(i)))
        return {k: getattr(i, k) for k, _ in IndexInfo._fields_}

    def export_bwt(self):
        n = self.info()["n_rows"]
        bf, br = np.empty(n, np.uint8), np.empty(n, np.uint8)
        _check(self._lib, self._lib.gm_index_export_bwt(self._h, _ptr(bf), _ptr(br)))
        return bf, br

    def export_sa(self):
        i = self.info()
        sa = np.empty(i["n_rows"], np.uint64 if i["row_bits"] == 64 else np.uint32)
        _check(self._lib, self._lib.gm_index_export_sa(self._h, _ptr(sa), sa.itemsize))
        return sa

    def export_sa_sampled(self):
        """(mark_words, samples): one bit per row, and SA[row] of the marked rows in row order"""
        n = C.c_uint64(0)
        i = self.info()
        _check(self._lib, self._lib.gm_index_export_sa_sampled(self._h, None, None, 0, C.byref(n)))
        mk = np.empty((i["n_rows"] + 31) // 32, np.uint32)
        sm = np.empty(n.value, np.uint64 if i["row_bits"] == 64 else np.uint32)
        _check(self._lib, self._lib.gm_index_export_sa_sampled(self._h, _ptr(mk), _ptr(sm), sm.itemsize, C.byref(n)))
        return mk, sm

    def _params(self, K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, kmer_range, chunks=None, piece_of=None):
        kb, ke = kmer_range if kmer_range is not None else (0, 0)
        flags = MAP_FLAG_RANGE if kmer_range is not None else 0   # an explicit range is literal: (b, b) computes nothing
        cb, ci, cs = chunks if chunks is not None else (0, 0, 0)   # (chunk_blocks, chunk_index, chunk_stride): interleaved chunks
        wb, we = (0, 0)
        if piece_of is not None:                                    # one launch of the share piece_of = (whole_begin, whole_end)
            assert kmer_range is not None
            flags |= MAP_FLAG_PIECE
            wb, we = piece_of
        return MapParams(K, E, -1 if overlap is None else overlap, infix, int(revcompl), value_bits, int(exclude_pseudo), flags, kb, ke, cb, ci, cs, 0, wb, we)

    def _slice(self, first_seq, n_seq):
        if n_seq is None:
            n_seq = len(self.seq_len) - first_seq
        tb = int(self.cum[first_seq])
        return n_seq, tb, int(self.cum[first_seq + n_seq]) - tb

    def map(self, K, E, first_seq=0, n_seq=None, overlap=None, infix=0, revcompl=True, value_bits=16,
            exclude_pseudo=False, intervals=None, seq_file_id=None, kmer_range=None, chunks=None, out=None):
        """gm_map: computeMappability for the fasta slice made of sequences [first_seq, first_seq+n_seq); host result
        (a new array, or `out` -- e.g. a page-locked one, host_pin)."""
        n_seq, tb, tl = self._slice(first_seq, n_seq)
        p = self._params(K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, kmer_range, chunks)
        if out is None:
            out = np.zeros(tl, dtype=np.uint8 if value_bits == 8 else np.uint16)
        assert out.size == tl and out.dtype == (np.uint8 if value_bits == 8 else np.uint16) and out.flags.c_contiguous
        iv = None if not intervals else np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        _check(self._lib, self._lib.gm_map(self._h, tb, tl, first_seq, n_seq, C.byref(p), _ptr(iv), 0 if iv is None else len(iv) // 2, _ptr(sf), _ptr(out)))
        return out

    def map_files(self, files, K, E, overlap=None, infix=0, revcompl=True, value_bits=16, exclude_pseudo=False, seq_file_id=None):
        """gm_map_files: the reference's loop over the FASTA files of an index (src/mappability.hpp:289-365) as one call; files = [(first_seq, n_seq), ..]
        consecutive and ascending; returns one host array per file."""
        p = self._params(K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, None, None)
        fs = np.ascontiguousarray([f for f, _ in files], dtype=np.uint32)
        ns = np.ascontiguousarray([n for _, n in files], dtype=np.uint32)
        outs = [np.zeros(self._slice(f, n)[2], dtype=np.uint8 if value_bits == 8 else np.uint16) for f, n in files]
        ptrs = (C.c_void_p * len(outs))(*[o.ctypes.data for o in outs])
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        _check(self._lib, self._lib.gm_map_files(self._h, len(files), _ptr(fs), _ptr(ns), C.byref(p), _ptr(sf), ptrs))
        return outs

    def map_shard(self, out, K, E, first_seq=0, n_seq=None, overlap=None, infix=0, revcompl=True, value_bits=16,
                  exclude_pseudo=False, intervals=None, seq_file_id=None, kmer_range=None, chunks=None):
        """gm_map_shard: this device's share (interleaved chunks or a k-mer range) written into the shared host vector `out`
        (numpy array of the slice's length); other positions are not touched."""
        n_seq, tb, tl = self._slice(first_seq, n_seq)
        assert out.size == tl and out.dtype == (np.uint8 if value_bits == 8 else np.uint16) and out.flags.c_contiguous
        p = self._params(K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, kmer_range, chunks)
        iv = None if not intervals else np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        _check(self._lib, self._lib.gm_map_shard(self._h, tb, tl, first_seq, n_seq, C.byref(p), _ptr(iv), 0 if iv is None else len(iv) // 2, _ptr(sf), _ptr(out)))

    def map_device(self, out_ptr, K, E, first_seq=0, n_seq=None, overlap=None, infix=0, revcompl=True, value_bits=16,
                   exclude_pseudo=False, intervals=None, seq_file_id=None, kmer_range=None, chunks=None, stream=None, piece_of=None):
        """gm_map_device: result written to device memory at out_ptr (e.g. a torch tensor's data_ptr()).
        chunks = (chunk_blocks, chunk_index, chunk_stride): only the interleaved chunks of this shard (ShardPlan.chunk_arg).
        piece_of = (whole_begin, whole_end): kmer_range is one launch of that share (GM_MAP_FLAG_PIECE): one clear, one correction pass per share."""
        n_seq, tb, tl = self._slice(first_seq, n_seq)
        p = self._params(K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, kmer_range, chunks, piece_of)
        iv = None if not intervals else np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        _check(self._lib, self._lib.gm_map_device(self._h, tb, tl, first_seq, n_seq, C.byref(p), _ptr(iv), 0 if iv is None else len(iv) // 2,
                                                  _ptr(sf), C.c_void_p(out_ptr), C.c_void_p(stream or 0)))

    def map_runs(self, K, E, first_seq=0, n_seq=None, overlap=None, infix=0, revcompl=True, value_bits=16, exclude_pseudo=False,
                 intervals=None, seq_file_id=None):
        """gm_map_runs: (start, length, value) arrays of the non-zero runs of the result."""
        n_seq, tb, tl = self._slice(first_seq, n_seq)
        p = self._params(K, E, overlap, infix, revcompl, value_bits, exclude_pseudo, None)
        iv = None if not intervals else np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        sf = None if seq_file_id is None else np.ascontiguousarray(seq_file_id, dtype=np.uint32)
        R = C.POINTER(Runs)()
        _check(self._lib, self._lib.gm_map_runs(self._h, tb, tl, first_seq, n_seq, C.byref(p), _ptr(iv), 0 if iv is None else len(iv) // 2, _ptr(sf), C.byref(R)))
        try:
            n = int(R.contents.n_runs)
            if n == 0:
                return np.zeros(0, np.uint64), np.zeros(0, np.uint64), np.zeros(0, np.uint16)
            return (np.ctypeslib.as_array(R.contents.start, shape=(n,)).copy(), np.ctypeslib.as_array(R.contents.length, shape=(n,)).copy(),
                    np.ctypeslib.as_array(R.contents.value, shape=(n,)).copy())
        finally:
            self._lib.gm_runs_free(R)

    def locate(self, K, E, first_seq=0, n_seq=None, overlap=None, infix=0, revcompl=True, intervals=None, kmer_range=None):
        """gm_locate: (pos_begin, plus_off, plus, minus_off, minus) -- occurrence lists per slice position,
        occurrences packed as seqNo << 32 | seqPos (global sequence numbers), each list sorted."""
        n_seq, tb, tl = self._slice(first_seq, n_seq)
        p = self._params(K, E, overlap, infix, revcompl, 16, False, kmer_range)
        iv = None if not intervals else np.ascontiguousarray(np.asarray(intervals, dtype=np.uint64).reshape(-1))
        L = C.POINTER(Locations)()
        _check(self._lib, self._lib.gm_locate(self._h, tb, tl, first_seq, n_seq, C.byref(p), _ptr(iv), 0 if iv is None else len(iv) // 2, C.byref(L)))
        # The arrays are VIEWS of the library's host arrays (no copy: a csv window of config C5 holds 2 GB of locations and copying them
        # here cost more than computing them); they keep the result alive, gm_locations_free runs when the last of them is collected.
        owner = _LocationsOwner(self._lib, L)
        c = L.contents
        n = int(c.n_positions)

        def view(ptr, count):
            if count <= 0 or not ptr:
                return np.zeros(0, np.uint64)      # empty windows / shards carry NULL payload pointers
            a = np.ctypeslib.as_array(ptr, shape=(count,)).view(_OwnedArray)
            a._owner = owner
            a.flags.writeable = False      # the library's memory: results are not scratch space
            return a
        po, mo = view(c.plus_off, n + 1), view(c.minus_off, n + 1)
        return int(c.pos_begin), po, view(c.plus, int(po[-1])), mo, view(c.minus, int(mo[-1]))

    def set_tuning(self, **knobs):
        """gm_index_set_tuning: scheduling knobs (verify_t, fetch_batch, ...); results never depend on them."""
        for k, v in knobs.items():
            _check(self._lib, self._lib.gm_index_set_tuning(self._h, k.encode(), int(v)))

    def sync(self):
        """gm_index_sync: wait for every call issued on this index; raises if a device-side check tripped."""
        _check(self._lib, self._lib.gm_index_sync(self._h))

    def kernel_times(self, n=64):
        """gm_map_kernel_times: search-kernel ms of the last n calls (oldest first), HIP events on the calls' streams."""
        buf = (C.c_double * n)()
        got = C.c_uint32(0)
        _check(self._lib, self._lib.gm_map_kernel_times(self._h, buf, n, C.byref(got)))
        return [float(buf[i]) for i in range(got.value)]

    def last_stats(self):
        s = MapStats()
        _check(self._lib, self._lib.gm_last_map_stats(self._h, C.byref(s)))
        d = {k: getattr(s, k) for k, _ in MapStats._fields_}
        d["detail"] = dict(zip(("steps_oss", "steps_ext", "ext_w1", "ext_w2_4", "oss_w1", "pushes", "verify_items", "verify_items_oss", "verify_chunks", "wave_iterations", "active_lane_sum", "verify_rounds", "cyc_fetch", "cyc_verify", "cyc_step", "cyc_pop", "cyc_share", "cyc_stage32", "cyc_stage1", "stolen",
                                "w_pop", "w_saturated", "w_share", "w_stage3", "w_stage2", "w_stage1", "w_defer", "w_split", "w_miss_round", "w_leaf", "w_leaf_flush",
                                "w_v_block", "w_v_chunk", "w_v_event", "w_v_kmer", "w_push_hbm", "jump_lookups", "correction_us", "table_q", "jump_filtered", "located_rows", "lf_steps", "max_stack", "self_hits", "verified_runs", "jump_words", "jump_filtered_rows2", "packets_slices"), list(s.detail)))
        d["detail"]["packets"] = d["detail"]["packets_slices"] & ((1 << 48) - 1)      # node packets written by phase A of the split search (profiling build)
        d["detail"]["slices"] = d["detail"].pop("packets_slices") >> 48                # slices the call's split search took (0: the one-loop kernel ran)
        return d
