"""ctypes binding of libpantas_gpu.so (include/pantas_gpu.h).

The library holds the HIP kernels; there is no Python or CPU substitute for it.  Importing
this module without the built library raises ImportError (run ``python -c 'import
__graft_entry__ as g; g.build()'`` or ``make -C pantas_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

ABI_VERSION = 8      # PANTAS_ABI_VERSION of include/pantas_gpu.h this file mirrors
_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpantas_gpu.so")


class PantasError(RuntimeError):
    """A failed libpantas_gpu call.  ``kind`` is the library's error class; for inputs the
    reference script dies on it is the name of the Python exception the reference raises."""

    def __init__(self, status: int, kind: str, message: str):
        super().__init__(message)
        self.status = status
        self.kind = kind


class _Sized(C.Structure):
    """A struct of the C ABI: starts with struct_size, which the library reads to know how many bytes of this
    declaration it may touch (include/pantas_gpu.h).  Set on construction."""

    def __init__(self, *args, **kw):
        super().__init__(*args, **kw)
        self.struct_size = C.sizeof(self)


_HEAD = [("struct_size", C.c_uint32), ("reserved0_", C.c_uint32)]


class GraphStats(_Sized):
    _fields_ = _HEAD + [("segments", C.c_uint64), ("link_slots", C.c_uint64), ("lines", C.c_uint64),
                ("s_lines", C.c_uint64), ("l_lines", C.c_uint64), ("p_lines", C.c_uint64),
                ("text_bytes", C.c_uint64), ("max_segment_len", C.c_uint32), ("dense_names", C.c_uint32),
                ("load_seconds", C.c_double)]


class BatchStats(_Sized):
    _fields_ = _HEAD + [("lines", C.c_uint64), ("rejected", C.c_uint64), ("skipped", C.c_uint64),
                ("records", C.c_uint64), ("path_nodes", C.c_uint64), ("cs_ops", C.c_uint64),
                ("del_records", C.c_uint64), ("del_path_nodes", C.c_uint64), ("text_bytes", C.c_uint64), ("alg_bytes", C.c_uint64),
                ("tokenize_seconds", C.c_double), ("upload_seconds", C.c_double),
                ("tokenizer", C.c_int32), ("reserved_", C.c_int32), ("tok_index_seconds", C.c_double),
                ("tok_measure_seconds", C.c_double), ("tok_write_seconds", C.c_double), ("tok_bucket_seconds", C.c_double),
                ("short_records", C.c_uint64), ("short_alg_bytes", C.c_uint64)]


class LaunchStats(_Sized):
    _fields_ = _HEAD + [("rare_ms", C.c_float), ("main_ms", C.c_float), ("main_short_ms", C.c_float), ("reserved_", C.c_float)]


_ALL_REDUCE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
_ALL_GATHER = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)
_COMM_COUNT = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_int))


class RcclApi(_Sized):
    """the RCCL entry points pantas_counters_allreduce calls: those of the library that made the communicator"""
    _fields_ = _HEAD + [("all_reduce", _ALL_REDUCE), ("all_gather", _ALL_GATHER), ("comm_count", _COMM_COUNT)]


class AllreduceStats(_Sized):
    _fields_ = _HEAD + [("mode", C.c_uint32), ("world", C.c_uint32), ("block_words", C.c_uint32), ("reserved1_", C.c_uint32),
                ("counter_bytes", C.c_uint64), ("exchanged_bytes", C.c_uint64),
                ("bitmap_ms", C.c_float), ("pack_ms", C.c_float), ("allreduce_ms", C.c_float), ("unpack_ms", C.c_float),
                ("total_ms", C.c_float), ("reserved2_", C.c_float)]


class CallOpts(_Sized):
    _fields_ = _HEAD + [("rca", C.c_int32), ("isw", C.c_int32), ("minintronsize", C.c_int32), ("events", C.c_uint32),
                ("novel", C.c_int32), ("no_annotated", C.c_int32), ("junction", C.c_char_p),
                ("shard_rank", C.c_uint32), ("shard_world", C.c_uint32), ("debug", C.c_int32), ("release_device", C.c_int32)]


class CallStats(_Sized):
    _fields_ = _HEAD + [("junctions", C.c_uint64), ("novel_links", C.c_uint64), ("evaluated", C.c_uint64),
                ("events", C.c_uint64), ("ex_entries", C.c_uint64), ("jn_entries", C.c_uint64),
                ("parse_seconds", C.c_double), ("upload_seconds", C.c_double), ("render_seconds", C.c_double),
                ("annotated_ms", C.c_float), ("novel_ms", C.c_float), ("around_ms", C.c_float), ("passes_ms", C.c_float)]


class SynthParams(_Sized):
    _fields_ = _HEAD + [("seed", C.c_uint64), ("n_genes", C.c_uint32), ("n_chrom", C.c_uint32), ("n_lines", C.c_uint64),
                ("base_id", C.c_uint64), ("bubble_rate", C.c_double), ("max_exons", C.c_uint32),
                ("intron_long_max", C.c_uint32), ("intergenic_min", C.c_uint32), ("intergenic_max", C.c_uint32),
                ("read_len", C.c_uint32), ("with_path", C.c_int32), ("zipf_a", C.c_double),
                ("p_lowmapq", C.c_double), ("p_unmapped", C.c_double), ("p_sub", C.c_double), ("p_indel", C.c_double),
                ("p_highdv", C.c_double), ("p_novel_skip", C.c_double), ("p_novel_ce", C.c_double),
                ("p_intron", C.c_double), ("p_novel_ss", C.c_double), ("p_clip", C.c_double),
                ("line_offset", C.c_uint64), ("gaf_only", C.c_int32), ("hot_sites", C.c_uint32), ("p_del_hot", C.c_double), ("p_two_indels", C.c_double)]


class SynthStats(_Sized):
    _fields_ = _HEAD + [("nodes", C.c_uint64), ("links", C.c_uint64), ("junction_links", C.c_uint64),
                ("transcripts", C.c_uint64), ("gaf_lines", C.c_uint64), ("gfa_bytes", C.c_uint64),
                ("gtf_bytes", C.c_uint64), ("gaf_bytes", C.c_uint64), ("seconds", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is not built: the pantas_amd engine is a HIP library and has no "
                          "fallback.  Build it with `make -C pantas_amd/csrc` (needs /opt/rocm/bin/hipcc).")
    lib = C.CDLL(LIB_PATH)
    vp, u64, i32 = C.c_void_p, C.c_uint64, C.c_int
    P = C.POINTER
    sig = {
        "pantas_last_error": (C.c_char_p, []),
        "pantas_last_error_kind": (C.c_char_p, []),
        "pantas_version": (C.c_char_p, []),
        "pantas_device_count": (i32, []),
        "pantas_abi_version": (i32, []),
        "pantas_device_warmup": (None, [i32]),
        "pantas_device_init": (None, [i32]),
        "pantas_graph_load": (i32, [C.c_char_p, P(vp)]),
        "pantas_graph_load_for_call": (i32, [C.c_char_p, P(vp)]),
        "pantas_graph_release_device": (i32, [vp]),
        "pantas_graph_free": (None, [vp]),
        "pantas_graph_get_stats": (i32, [vp, P(GraphStats)]),
        "pantas_graph_to_device": (i32, [vp, i32]),
        "pantas_augment_reset": (i32, [vp]),
        "pantas_augment_load_gaf": (i32, [vp, C.c_char_p, i32, u64, u64, u64, P(vp), P(BatchStats)]),
        "pantas_gaf_split": (i32, [C.c_char_p, i32, i32, P(u64), P(u64), P(u64)]),
        "pantas_batch_free": (None, [vp]),
        "pantas_augment_launch": (i32, [vp, vp, P(LaunchStats)]),
        "pantas_augment_launch_async": (i32, [vp, vp]),
        "pantas_augment_launch_collect": (i32, [vp, P(C.c_float), u64, P(u64)]),
        "pantas_augment_reserve": (i32, [vp, u64]),
        "pantas_augment_launch_rare": (i32, [vp, vp, P(C.c_float)]),
        "pantas_augment_launch_main": (i32, [vp, vp, P(C.c_float)]),
        "pantas_augment_flags_export": (i32, [vp, vp, u64, P(u64)]),
        "pantas_augment_flags_import": (i32, [vp, vp, u64]),
        "pantas_augment_counters": (i32, [vp, P(vp), P(u64)]),
        "pantas_counters_bitmap": (i32, [vp, C.c_uint32, P(vp), P(u64)]),
        "pantas_augment_track_touched": (i32, [vp, i32]),
        "pantas_counters_pack": (i32, [vp, vp, C.c_uint32, P(vp), P(u64)]),
        "pantas_counters_unpack": (i32, [vp]),
        "pantas_quant_run": (i32, [P(C.c_char_p), P(C.c_char_p), C.c_uint32, i32, i32, i32]),
        "pantas_remap_run": (i32, [C.c_char_p, C.c_char_p, C.c_int64, i32]),
        "pantas_counters_allreduce": (i32, [vp, P(RcclApi), vp, vp, C.c_uint32, P(AllreduceStats)]),
        "pantas_augment_sparse_count": (i32, [vp, P(u64)]),
        "pantas_augment_sparse_export": (i32, [vp, P(u64), u64]),
        "pantas_augment_sparse_import": (i32, [vp, P(u64), u64]),
        "pantas_augment_finish": (i32, [vp]),
        "pantas_augment_read": (i32, [vp, vp, vp, vp, vp]),
        "pantas_augment_write_gfa": (i32, [vp, i32]),
        "pantas_augment_run": (i32, [vp, C.c_char_p, i32, i32, P(BatchStats), P(LaunchStats)]),
        "pantas_augment_stream_rare": (i32, [vp, C.c_char_p, i32, u64, u64, u64, P(BatchStats), P(C.c_float)]),
        "pantas_augment_stream_main": (i32, [vp, P(C.c_float)]),
        "pantas_augment_stream_info": (i32, [vp, P(u64), P(u64), P(u64)]),
        "pantas_call_prepare": (i32, [vp, C.c_char_p, i32]),
        "pantas_call_run": (i32, [vp, P(CallOpts), i32, i32, P(CallStats)]),
        "pantas_synth_defaults": (None, [P(SynthParams)]),
        "pantas_synth_generate": (i32, [P(SynthParams), P(vp), P(SynthStats)]),
        "pantas_synth_text": (i32, [vp, i32, P(C.c_char_p), P(u64)]),
        "pantas_synth_write": (i32, [vp, C.c_char_p, u64]),
        "pantas_synth_free": (None, [vp]),
        "pantas_graph_load_mem": (i32, [C.c_char_p, u64, P(vp)]),
        "pantas_augment_load_gaf_mem": (i32, [vp, C.c_char_p, u64, i32, u64, P(vp), P(BatchStats)]),
        "pantas_call_prepare_mem": (i32, [vp, C.c_char_p, u64, i32]),
        "pantas_call_prepare_fused": (i32, [vp, C.c_char_p, C.c_char_p, u64, i32]),
        "pantas_augment_tokenizer": (i32, [vp, i32]),
        "pantas_graph_load_cached": (i32, [C.c_char_p, C.c_char_p, i32, P(vp), P(i32)]),
        "pantas_batch_download": (i32, [vp, vp, vp, vp, vp, vp, vp]),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    if lib.pantas_abi_version() != ABI_VERSION:
        raise ImportError(f"{LIB_PATH} was built for ABI version {lib.pantas_abi_version()}, this binding declares "
                          f"{ABI_VERSION} (include/pantas_gpu.h PANTAS_ABI_VERSION): rebuild with `make -C pantas_amd/csrc`")
    return lib, sig


lib, SIGNATURES = _load()


def check(status: int) -> None:
    if status != 0:
        raise PantasError(status, (lib.pantas_last_error_kind() or b"").decode(),
                          (lib.pantas_last_error() or b"").decode())


def device_count() -> int:
    return lib.pantas_device_count()
