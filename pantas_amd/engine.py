"""Object wrappers over the C ABI: a loaded graph and resident GAF batches."""
from __future__ import annotations

import ctypes as C
import os

from . import _lib
from ._lib import (BatchStats, CallOpts, CallStats, GraphStats, LaunchStats, PantasError, SynthParams,  # noqa: F401
                   SynthStats, check, lib)


class Batch:
    def __init__(self, handle, stats: BatchStats):
        self._h = handle
        self.stats = stats

    def free(self):
        if self._h:
            lib.pantas_batch_free(self._h)
            self._h = None

    def __del__(self):
        self.free()


def graph_cache_path(gfa_path: str):
    """Where the binary graph cache of `gfa_path` lives according to PANTAS_GRAPH_CACHE: unset/0 = no cache,
    1 = next to the GFA (<gfa>.pgc), anything else = a directory for the cache files."""
    where = os.environ.get("PANTAS_GRAPH_CACHE", "")
    if where in ("", "0"):
        return None
    if where == "1":
        return gfa_path + ".pgc"
    os.makedirs(where, exist_ok=True)
    return os.path.join(where, os.path.basename(gfa_path) + ".pgc")


class Graph:
    """A GFA loaded into the CSR node/link layout (host) and, after ``to_device``, HBM."""

    def __init__(self, gfa_path: str = None, *, text=None, cache: str = None, for_call: bool = False):
        """From a GFA file, or from (address, size) of GFA text kept alive by the caller.  ``cache``: path of the
        binary cache of the parsed arrays (used when it matches the file, written otherwise).  ``for_call``: the caller goes
        on to ``call_prepare`` (the tags of the lines are taken while the text is parsed, pantas_graph_load_for_call)."""
        h = C.c_void_p()
        self.used_cache = False
        if cache is not None and text is None:
            used = C.c_int32()
            check(lib.pantas_graph_load_cached(os.fsencode(gfa_path), os.fsencode(cache), 1, C.byref(h), C.byref(used)))
            self.used_cache = bool(used.value)
        elif text is not None:
            self._keep = text
            check(lib.pantas_graph_load_mem(C.c_char_p(text[0]), text[1], C.byref(h)))
        elif for_call:
            check(lib.pantas_graph_load_for_call(os.fsencode(gfa_path), C.byref(h)))
        else:
            check(lib.pantas_graph_load(os.fsencode(gfa_path), C.byref(h)))
        self._h = h
        self.path = gfa_path

    def release_device(self):
        """Give the device memory of this handle back; host results (a finished augment run) stay."""
        check(lib.pantas_graph_release_device(self._h))

    def close(self):
        if self._h:
            lib.pantas_graph_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    @property
    def stats(self) -> GraphStats:
        s = GraphStats()
        check(lib.pantas_graph_get_stats(self._h, C.byref(s)))
        return s

    def to_device(self, device: int = 0) -> "Graph":
        check(lib.pantas_graph_to_device(self._h, device))
        return self

    # ---- augment -------------------------------------------------------------------------
    def augment_reset(self):
        check(lib.pantas_augment_reset(self._h))

    def load_gaf(self, gaf_path: str, thr: int = 20, begin: int = 0, end: int = 0, first_line: int = 0) -> Batch:
        b = C.c_void_p()
        st = BatchStats()
        check(lib.pantas_augment_load_gaf(self._h, os.fsencode(gaf_path), thr, begin, end, first_line,
                                          C.byref(b), C.byref(st)))
        return Batch(b, st)

    def load_gaf_mem(self, text, thr: int = 20, first_line: int = 0) -> Batch:
        b = C.c_void_p()
        st = BatchStats()
        check(lib.pantas_augment_load_gaf_mem(self._h, C.c_char_p(text[0]), text[1], thr, first_line,
                                              C.byref(b), C.byref(st)))
        return Batch(b, st)

    TOKENIZER = {"auto": 0, "host": 1, "device": 2}

    def set_tokenizer(self, where: str):
        """Where GAF text becomes records: "device" (GPU kernels), "host" (threads) or "auto"."""
        check(lib.pantas_augment_tokenizer(self._h, self.TOKENIZER[where]))

    def download_batch(self, batch: Batch):
        """(hdr[n,4] int32 {start, tail, line_id, flags}, node_off, op_off, nodes, ops) of a resident batch."""
        import numpy as np
        st = batch.stats
        hdr = np.empty((st.records, 4), dtype=np.int32)
        node_off = np.empty(st.records + 1, dtype=np.uint32)
        op_off = np.empty(st.records + 1, dtype=np.uint32)
        nodes = np.empty(max(1, st.path_nodes), dtype=np.uint32)
        ops = np.empty(max(1, st.cs_ops), dtype=np.uint32)
        check(lib.pantas_batch_download(self._h, batch._h, hdr.ctypes.data, node_off.ctypes.data, op_off.ctypes.data,
                                        nodes.ctypes.data, ops.ctypes.data))
        return hdr, node_off, op_off, nodes[:st.path_nodes], ops[:st.cs_ops]

    def launch(self, batch: Batch) -> LaunchStats:
        st = LaunchStats()
        check(lib.pantas_augment_launch(self._h, batch._h, C.byref(st)))
        return st

    def launch_async(self, batch: Batch):
        """the two kernels enqueued, not waited for (at most 64 in flight); launch_collect() waits and returns their times"""
        check(lib.pantas_augment_launch_async(self._h, batch._h))

    def launch_collect(self):
        """[(rare_ms, main_ms, main_short_ms)] of the launches enqueued by launch_async since the last collect, oldest first"""
        ms = (C.c_float * (3 * 64))()
        n = C.c_uint64()
        check(lib.pantas_augment_launch_collect(self._h, ms, 64, C.byref(n)))
        return [(ms[3 * i], ms[3 * i + 1], ms[3 * i + 2]) for i in range(min(n.value, 64))]

    def reserve(self, keys: int):
        check(lib.pantas_augment_reserve(self._h, keys))

    def launch_rare(self, batch: Batch) -> float:
        ms = C.c_float()
        check(lib.pantas_augment_launch_rare(self._h, batch._h, C.byref(ms)))
        return ms.value

    def launch_main(self, batch: Batch) -> float:
        ms = C.c_float()
        check(lib.pantas_augment_launch_main(self._h, batch._h, C.byref(ms)))
        return ms.value

    def flags_export(self):
        import numpy as np
        n = C.c_uint64()
        check(lib.pantas_augment_flags_export(self._h, None, 0, C.byref(n)))
        codes = np.zeros(n.value, dtype=np.uint32)
        if n.value:
            check(lib.pantas_augment_flags_export(self._h, codes.ctypes.data, n.value, C.byref(n)))
        return codes

    def flags_import(self, codes):
        import numpy as np
        codes = np.ascontiguousarray(codes, dtype=np.uint32)
        check(lib.pantas_augment_flags_import(self._h, codes.ctypes.data, codes.shape[0]))

    def counters(self):
        """(device pointer, words) of the dense counters [NC | IL0 | OLL | RC]."""
        p = C.c_void_p()
        n = C.c_uint64()
        check(lib.pantas_augment_counters(self._h, C.byref(p), C.byref(n)))
        return p.value, n.value

    def _device_i32(self, ptr, words, device):
        """zero-copy int32 torch view of `words` words of library-owned device memory"""
        import torch

        class _Dev:
            __cuda_array_interface__ = {"shape": (max(int(words), 1),), "typestr": "<i4", "data": (ptr, False), "version": 2}
        try:
            return torch.as_tensor(_Dev(), device=torch.device("cuda", device))[:int(words)]
        except RuntimeError as e:
            if "No HIP GPUs" in str(e):
                raise RuntimeError("torch.cuda cannot start: torch bundles its own HIP runtime, which must be the first one a "
                                   "process loads -- `import torch` before `pantas_amd` wherever device tensors are shared") from e
            raise

    def counters_tensor(self, device: int = 0):
        ptr, words = self.counters()
        return self._device_i32(ptr, words, device)

    def counters_numpy(self):
        """a host copy of the dense device counters ({NC, IL0, OLL, RC first slot} per node, then RC per slot) without torch:
        hipMemcpy through ctypes (tests of the C-ABI collective, whose process has no torch)"""
        import numpy as np
        ptr, words = self.counters()
        out = np.empty(words, dtype=np.uint32)
        hip = C.CDLL("libamdhip64.so")
        hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
        rc = hip.hipMemcpy(out.ctypes.data, ptr, words * 4, 2)      # hipMemcpyDeviceToHost
        if rc != 0:
            raise RuntimeError(f"hipMemcpy: {rc}")
        return out

    def track_touched(self, on: bool = True):
        """from the next augment_reset on the kernels keep the bitmap of touched counter blocks themselves (read-sharded runs:
        touched_bitmap then folds 20 MB instead of reading 5 GB of counters); pantas_augment_track_touched"""
        check(lib.pantas_augment_track_touched(self._h, 1 if on else 0))

    # block-sparse counter exchange (exchange.hip; the collectives between the three steps are the caller's)
    def touched_bitmap(self, block_words: int = 16, device: int = 0):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib.pantas_counters_bitmap(self._h, block_words, C.byref(p), C.byref(n)))
        return self._device_i32(p.value, n.value, device)

    def counters_allreduce(self, comm, stream=None, block_words: int = 16, api=None):
        """the read-sharded run's one collective through the C ABI (pantas_counters_allreduce): `comm` an ncclComm_t (integer
        address), `api` an _lib.RcclApi with the entry points of the RCCL that made it (None: found in the process)"""
        from ._lib import AllreduceStats
        st = AllreduceStats()
        check(lib.pantas_counters_allreduce(self._h, C.byref(api) if api is not None else None, C.c_void_p(comm),
                                            C.c_void_p(stream) if stream else None, block_words, C.byref(st)))
        return st

    def counters_pack(self, bitmaps, n_bitmaps: int, device: int = 0):
        p, n = C.c_void_p(), C.c_uint64()
        check(lib.pantas_counters_pack(self._h, bitmaps.data_ptr(), n_bitmaps, C.byref(p), C.byref(n)))
        return self._device_i32(p.value, n.value, device)

    def counters_unpack(self):
        check(lib.pantas_counters_unpack(self._h))

    def finish(self):
        check(lib.pantas_augment_finish(self._h))

    def read_counters(self):
        import numpy as np
        s = self.stats
        nc = np.zeros(s.segments, dtype=np.uint32)
        il0 = np.zeros(s.segments, dtype=np.uint32)
        oll = np.zeros(s.segments, dtype=np.uint32)
        rc = np.zeros(s.link_slots, dtype=np.uint32)
        check(lib.pantas_augment_read(self._h, nc.ctypes.data, il0.ctypes.data, oll.ctypes.data, rc.ctypes.data))
        return nc, il0, oll, rc

    def sparse_export(self):
        import numpy as np
        n = C.c_uint64()
        check(lib.pantas_augment_sparse_count(self._h, C.byref(n)))
        rows = np.zeros((n.value, 4), dtype=np.uint64)
        check(lib.pantas_augment_sparse_export(self._h, rows.ctypes.data_as(C.POINTER(C.c_uint64)), n.value))
        return rows

    def sparse_import(self, rows):
        import numpy as np
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        check(lib.pantas_augment_sparse_import(self._h, rows.ctypes.data_as(C.POINTER(C.c_uint64)), rows.shape[0]))

    def write_gfa(self, fd: int):
        check(lib.pantas_augment_write_gfa(self._h, fd))

    def augment(self, gaf_path: str, thr: int = 20, fd_out: int = -1):
        """The whole script on a GAF of any size (cut into batches of PANTAS_BATCH_BYTES of text)."""
        bs, ls = BatchStats(), LaunchStats()
        check(lib.pantas_augment_run(self._h, os.fsencode(gaf_path), thr, fd_out, C.byref(bs), C.byref(ls)))
        return bs, ls

    def stream_rare(self, gaf_path: str, thr: int = 20, begin: int = 0, end: int = 0, first_line: int = 0):
        """Rare-key pass of a streamed run over bytes [begin, end) of the GAF: (totals over its batches, kernel ms)."""
        bs, ms = BatchStats(), C.c_float()
        check(lib.pantas_augment_stream_rare(self._h, os.fsencode(gaf_path), thr, begin, end, first_line, C.byref(bs), C.byref(ms)))
        return bs, ms.value

    def stream_main(self) -> float:
        ms = C.c_float()
        check(lib.pantas_augment_stream_main(self._h, C.byref(ms)))
        return ms.value

    def stream_info(self):
        """(batches, resident after the rare-key pass, tokenised twice) of the streamed run."""
        a, b, c = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.pantas_augment_stream_info(self._h, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    # ---- call ----------------------------------------------------------------------------
    def call_prepare(self, gtf_path: str, device: int = 0):
        check(lib.pantas_call_prepare(self._h, os.fsencode(gtf_path), device))

    def call_prepare_mem(self, text, device: int = 0):
        check(lib.pantas_call_prepare_mem(self._h, C.c_char_p(text[0]), text[1], device))

    def call_prepare_fused(self, gtf_path: str = None, *, text=None, device: int = 0):
        """Prepare call on the result of this handle's last augment run (no GFA text round trip)."""
        if text is not None:
            check(lib.pantas_call_prepare_fused(self._h, None, C.c_char_p(text[0]), text[1], device))
        else:
            check(lib.pantas_call_prepare_fused(self._h, os.fsencode(gtf_path), None, 0, device))

    def call(self, fd_csv: int = -1, fd_err: int = -1, rca: int = -1, novel: bool = False,
             no_annotated: bool = False, events=("ES", "SS", "IR"), isw: int = 5, minintronsize: int = 100,
             junction=None, shard=(0, 1), debug: bool = False, release_device: bool = False) -> CallStats:
        mask = 0
        for e in events:
            mask |= {"ES": 1, "SS": 2, "IR": 4}.get(e, 0)
        o = CallOpts(rca=rca, isw=isw, minintronsize=minintronsize, events=mask, novel=int(bool(novel)),
                     no_annotated=int(bool(no_annotated)), junction=None if junction is None else os.fsencode(junction),
                     shard_rank=shard[0], shard_world=shard[1], debug=int(bool(debug)), release_device=int(bool(release_device)))
        st = CallStats()
        check(lib.pantas_call_run(self._h, C.byref(o), fd_csv, fd_err, C.byref(st)))
        return st


class Synth:
    """In-memory synthetic GFA / GTF / GAF text (SURVEY.md 8-d recipe at BASELINE.json scales)."""

    def __init__(self, **kw):
        p = SynthParams()
        lib.pantas_synth_defaults(C.byref(p))
        for k, v in kw.items():
            if not hasattr(p, k):
                raise TypeError(f"unknown synth parameter {k}")
            setattr(p, k, v)
        self.params = p
        self.stats = SynthStats()
        h = C.c_void_p()
        check(lib.pantas_synth_generate(C.byref(p), C.byref(h), C.byref(self.stats)))
        self._h = h

    def text(self, which: int):
        """(address, size) of the GFA (0), GTF (1) or GAF (2) text; valid while this object lives."""
        d = C.c_char_p()
        n = C.c_uint64()
        pp = C.cast(C.byref(d), C.POINTER(C.c_char_p))
        check(lib.pantas_synth_text(self._h, which, pp, C.byref(n)))
        return (C.cast(d, C.c_void_p).value, n.value)

    @property
    def gfa(self): return self.text(0)
    @property
    def gtf(self): return self.text(1)
    @property
    def gaf(self): return self.text(2)

    def write(self, prefix: str, max_gaf_lines: int = 0):
        check(lib.pantas_synth_write(self._h, os.fsencode(prefix), max_gaf_lines))

    def close(self):
        if self._h:
            lib.pantas_synth_free(self._h)
            self._h = None

    def __del__(self):
        self.close()
