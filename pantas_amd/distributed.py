"""Sharded `pantas augment` over the GPUs of one node.

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on ROCm, "gloo" in the
CPU tests).  The graph is replicated; the GAF is cut into `world` contiguous byte ranges at
line starts; global line index = lines before the shard + local index (first-seen ordering).

    rank r:  tokenise shard r -> rare-key kernel
    all:     all-gather the node flags the rare kernels set      (tiny, variable length)
    rank r:  main kernel over shard r
    all:     ONE sum over the dense counters [NC | IL0 | OLL | RC]: bitmap of touched blocks (all-gather) ->
             all-reduce of the union's blocks only (allreduce_counters; dense all-reduce when most are touched)
    rank 0:  gathers the sparse side results (novel links, rare IL/OL keys; counts add,
             first-seen takes the minimum) and writes the GFA

The driver is written against a small backend interface so that the CPU test-suite can run the
identical exchange logic over gloo with the CPU harness standing in for the kernels.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np


class LibBackend:
    """libpantas_gpu on one GPU."""

    def __init__(self, gfa_path: str, device: int, dist=None):
        """`dist` (an initialised torch.distributed with more than one rank): ONE rank parses the GFA text and leaves the binary
        cache of the parsed arrays (graph_cache.cpp) where the others load it -- eight ranks on a 16-core quota parsing 19 GB each
        would make the parse, not xGMI, the limit of the run (whole-genome graph: 11 s of parse, 2.4 s from the cache)."""
        from .engine import Graph, graph_cache_path
        self.device = device
        cache = graph_cache_path(gfa_path)
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1 and not os.environ.get("PANTAS_EVERY_RANK_PARSES"):
            self.g = self._load_shared(Graph, gfa_path, cache, dist)
            self.g.to_device(device)
        else:
            self.g = Graph(gfa_path, cache=cache).to_device(device)
        # a read-sharded run ends with the exchange of the touched counter blocks: the kernels keep that bitmap themselves
        # (PANTAS_TRACK_TOUCHED=0: the exchange reads the counters to find the blocks, as until round 4)
        if dist is not None and dist.is_initialized() and dist.get_world_size() > 1 and os.environ.get("PANTAS_TRACK_TOUCHED", "1") != "0":
            self.g.track_touched(True)
        self.g.augment_reset()
        self.batch = None

    @staticmethod
    def _load_shared(Graph, gfa_path, cache, dist):
        """Rank 0 parses and writes the cache, the others load it.  With no cache asked for, the cache of this run lives in a
        directory only this user can enter (mkdtemp, mode 0700: nobody else can plant a file there), its path goes to the others by
        broadcast, and a rank that does not find the file (another node) parses the GFA itself WITHOUT leaving a cache of its own.
        A failure of rank 0 is broadcast in the place of the path, so that the others raise instead of waiting in a barrier; a
        failure of ANY rank reaches every rank the same way (the verdicts are all-gathered where a bare barrier stood): a rank
        that raised alone used to leave the others waiting in their next collective until the process group timed out."""
        import shutil
        import tempfile
        rank0 = dist.get_rank() == 0
        tmpdir, g, msg = None, None, [None, None]        # [cache path of the run, error of rank 0]
        if rank0:
            try:
                path = cache
                if path is None:
                    where = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
                    tmpdir = tempfile.mkdtemp(prefix="pantas_pgc_", dir=where)
                    path = os.path.join(tmpdir, "graph.pgc")
                g = Graph(gfa_path, cache=path)
                msg[0] = path
            except BaseException as e:       # the others must hear of it before it is raised here
                msg[1] = f"{type(e).__name__}: {e}"
                err = e
        dist.broadcast_object_list(msg, src=0)
        try:
            if msg[1] is not None:
                if rank0:
                    raise err
                raise RuntimeError(f"rank 0 could not load {gfa_path}: {msg[1]}")
            if not rank0:
                path = msg[0]
                mine = cache is not None and path == cache      # a cache the user asked for: every rank may use and write it
                if not mine:
                    try:
                        st = os.stat(path)
                        ok = st.st_uid == os.getuid()           # ours, in a directory only we can enter
                    except OSError:
                        ok = False                              # another node: parse, leave nothing behind
                    path = path if ok else None
                g = Graph(gfa_path, cache=path) if path is not None else Graph(gfa_path)
        except BaseException as e:
            mine_err = e
        else:
            mine_err = None
        # every rank has loaded (or failed): the cache of the run can go, and every rank hears of every failure
        verdicts = [None] * dist.get_world_size()
        dist.all_gather_object(verdicts, None if mine_err is None else f"{type(mine_err).__name__}: {mine_err}")
        if tmpdir is not None:
            shutil.rmtree(tmpdir, ignore_errors=True)
        if mine_err is not None:
            raise mine_err
        failed = [(r, v) for r, v in enumerate(verdicts) if v is not None]
        if failed:
            raise RuntimeError(f"rank {failed[0][0]} could not load {gfa_path}: {failed[0][1]}")
        return g

    def split(self, gaf_path, rank, world):
        from ._lib import check, lib
        b, e, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.pantas_gaf_split(os.fsencode(gaf_path), rank, world, C.byref(b), C.byref(e), C.byref(n)))
        return b.value, e.value, n.value

    def load_shard(self, gaf_path, thr, begin, end, first_line):
        """Tokenises the shard batch by batch (a shard of any size: PANTAS_BATCH_BYTES of text per batch) and runs every
        batch's rare-key kernel -- the first pass of a streamed run (pantas_augment_stream_rare)."""
        # begin == end is an empty shard -- of a file with fewer lines (or bytes) than ranks, every cut of the low ranks is 0 --
        # except for an EMPTY FILE, whose one "shard" [0, 0) is passed on (end 0 means "to EOF" to the stream API: harmless there)
        self.empty = begin == end and os.path.getsize(gaf_path) > 0
        if self.empty:                               # more ranks than lines: nothing to read, no batch, no kernel
            self.rare_ms = 0.0
            return dict(lines=0, rejected=0, del_path_nodes=0)
        s, self.rare_ms = self.g.stream_rare(gaf_path, thr, begin, end, first_line)
        return dict(lines=s.lines, rejected=s.rejected, del_path_nodes=s.del_path_nodes)

    def reserve(self, keys): self.g.reserve(keys)      # the key table grows in place when a later launch needs the room
    def launch_rare(self): return self.rare_ms          # ran batch by batch inside load_shard
    def launch_main(self): return 0.0 if self.empty else self.g.stream_main()
    def flags_export(self): return self.g.flags_export()
    def flags_import(self, codes): self.g.flags_import(codes)

    def counters_tensor(self): return self.g.counters_tensor(self.device)
    def touched_bitmap(self, block_words): return self.g.touched_bitmap(block_words, self.device)
    def counters_pack(self, bitmaps, n): return self.g.counters_pack(bitmaps, n, self.device)
    def counters_unpack(self): self.g.counters_unpack()

    def finish(self): self.g.finish()
    def sparse_export(self): return self.g.sparse_export()
    def sparse_import(self, rows): self.g.sparse_import(rows)
    def write_gfa(self, fd): self.g.write_gfa(fd)


def _gather_var(dist, arr: np.ndarray, device):
    """all-gather of 1-D/2-D integer arrays of different lengths; returns the list per rank."""
    import torch
    world = dist.get_world_size()
    flat = np.ascontiguousarray(arr).reshape(-1)
    as_i64 = flat.view(np.int64) if flat.dtype == np.uint64 else flat.astype(np.int64)
    n = torch.tensor([as_i64.shape[0]], dtype=torch.int64, device=device)
    sizes = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    buf = torch.zeros(m, dtype=torch.int64, device=device)
    if as_i64.shape[0]:
        buf[:as_i64.shape[0]] = torch.from_numpy(as_i64.copy()).to(device)
    outs = [torch.zeros(m, dtype=torch.int64, device=device) for _ in range(world)]
    dist.all_gather(outs, buf)
    return [o[:s].cpu().numpy() for o, s in zip(outs, sizes)]


EXCHANGE_BLOCK_WORDS = 16      # counters per exchanged block (64 B); PANTAS_EXCHANGE_BLOCK = 8 | 16 | 32 | 64 | 0 (dense)


def allreduce_counters(backend, dist, block_words=None) -> dict:
    """THE data-path collective of a read-sharded replicate: sums the dense counters [NC | IL0 | OLL | RC] over the
    ranks (uint32 sums as int32 bit patterns), in place.

    Reads cover exons only, so almost every counter is zero on every rank, and xGMI is point-to-point: a ring
    all-reduce over the whole array (4 GB at human whole-genome scale) is bound by one link.  The ranks therefore
    exchange a bitmap of the blocks they touched (all-gather, W / block_words bits each), derive the same union,
    and all-reduce only those blocks (exchange.hip).  When more than half of the blocks are touched the dense
    all-reduce is used instead."""
    import time
    import torch
    if block_words is None:
        block_words = int(os.environ.get("PANTAS_EXCHANGE_BLOCK", EXCHANGE_BLOCK_WORDS))
    world = dist.get_world_size()
    ct = backend.counters_tensor()
    info = {"mode": "dense", "counter_bytes": ct.numel() * 4, "exchanged_bytes": ct.numel() * 4, "block_words": block_words}

    def wait():
        if ct.is_cuda:
            torch.cuda.synchronize()
    t0 = time.perf_counter()
    if block_words and ct.numel():
        bm = backend.touched_bitmap(block_words)          # (returns when the kernel is done)
        info["bitmap_kernel_ms"] = (time.perf_counter() - t0) * 1e3
        t1 = time.perf_counter()
        allbm = torch.empty(world * bm.numel(), dtype=bm.dtype, device=bm.device)
        dist.all_gather_into_tensor(allbm, bm)
        wait()
        info["bitmap_allgather_ms"] = (time.perf_counter() - t1) * 1e3
        info["bitmap_ms"] = (time.perf_counter() - t0) * 1e3
        # every collective of the exchange with its message size, so that a run on N GPUs explains itself
        info["collectives"] = [{"op": "all_gather", "what": "bitmaps of touched blocks", "bytes_per_rank": bm.numel() * 4,
                                "bytes_gathered": allbm.numel() * 4, "ms": info["bitmap_allgather_ms"]}]
        t1 = time.perf_counter()
        staging = backend.counters_pack(allbm, world)
        info["pack_ms"] = (time.perf_counter() - t1) * 1e3
        if 2 * staging.numel() <= ct.numel():
            t1 = time.perf_counter()
            if staging.numel():
                dist.all_reduce(staging)
                wait()
            info["allreduce_ms"] = (time.perf_counter() - t1) * 1e3
            info["collectives"].append({"op": "all_reduce", "what": "blocks of the union (staging buffer)", "bytes": staging.numel() * 4,
                                        "ms": info["allreduce_ms"]})
            t1 = time.perf_counter()
            backend.counters_unpack()
            info["unpack_ms"] = (time.perf_counter() - t1) * 1e3
            info.update(mode="block-sparse", exchanged_bytes=staging.numel() * 4 + allbm.numel() * 4,
                        touched_fraction=staging.numel() / max(1, ct.numel()), total_ms=(time.perf_counter() - t0) * 1e3)
            return info
    t1 = time.perf_counter()
    dist.all_reduce(ct)
    wait()
    info["allreduce_ms"] = (time.perf_counter() - t1) * 1e3
    info.setdefault("collectives", []).append({"op": "all_reduce", "what": "the dense counter array", "bytes": ct.numel() * 4, "ms": info["allreduce_ms"]})
    info["total_ms"] = (time.perf_counter() - t0) * 1e3
    return info


def augment_sharded(backend, gaf_path: str, thr: int, out_fd: int, dist, device=None) -> dict:
    """Runs one sharded augment; rank 0 writes the augmented GFA to out_fd.  Returns totals."""
    import torch
    rank, world = dist.get_rank(), dist.get_world_size()
    device = device if device is not None else torch.device("cpu")
    begin, end, lines_before = backend.split(gaf_path, rank, world)
    st = backend.load_shard(gaf_path, thr, begin, end, lines_before)
    tot = torch.tensor([st["lines"], st["rejected"], 4 * st["del_path_nodes"]], dtype=torch.int64, device=device)
    dist.all_reduce(tot)
    lines, rejected, hist_keys = (int(x) for x in tot.tolist())
    backend.reserve(hist_keys + 1024)
    backend.launch_rare()
    codes = _gather_var(dist, backend.flags_export().astype(np.int64), device)
    union = np.unique(np.concatenate(codes)).astype(np.uint32) if codes else np.zeros(0, np.uint32)
    backend.flags_import(union)
    backend.launch_main()
    xinfo = allreduce_counters(backend, dist)    # THE data-path collective
    backend.finish()
    rows = _gather_var(dist, backend.sparse_export(), device)
    if rank == 0:
        others = [r.view(np.uint64).reshape(-1, 4) for i, r in enumerate(rows) if i != 0 and r.shape[0]]
        if others:
            backend.sparse_import(np.concatenate(others))
        if out_fd is not None and out_fd >= 0:
            backend.write_gfa(out_fd)
    dist.barrier()
    return dict(lines=lines, rejected=rejected, exchange=xinfo)


def call_sharded(run_shard, out, dist) -> int:
    """Sharded `pantas call`: junctions and novel links are independent given the graph and
    the coverage, so rank r evaluates every world-th one -- ``run_shard(rank, world)`` returns that shard's
    CSV text WITHOUT the header -- and rank 0 concatenates the shards behind one header.  No data-path
    collective; the row multiset equals a single-process run."""
    rank, world = dist.get_rank(), dist.get_world_size()
    body = run_shard(rank, world)
    parts = [None] * world
    dist.all_gather_object(parts, body)
    n = 0
    if rank == 0:
        from .call import CSV_HEADER
        out.write(CSV_HEADER + "\n")
        for p in parts:
            out.write(p)
            n += p.count("\n")
        out.flush()
    return n


def call_main(argv=None) -> int:
    """torchrun entry:  python -m torch.distributed.run --nproc-per-node N -m pantas_amd.distributed call [call options] GFA GTF"""
    import sys
    import tempfile
    import torch
    import torch.distributed as dist
    from .call import build_parser
    from .engine import Graph
    args = build_parser().parse_args(argv)
    local = int(os.environ.get("LOCAL_RANK", "0"))
    sys.stdout.flush()
    data_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    g = Graph(args.GFA)
    g.call_prepare(args.GTF, local)

    def run_shard(rank, world):
        with tempfile.TemporaryFile() as f:
            g.call(f.fileno(), 2, rca=args.rca, novel=args.novel, no_annotated=args.annotated, events=args.events,
                   isw=args.isw, minintronsize=args.minintronsize, junction=args.junction, shard=(rank, world))
            f.seek(0)
            text = f.read().decode()
        return text.split("\n", 1)[1] if "\n" in text else ""

    with os.fdopen(data_fd, "w") as out:
        call_sharded(run_shard, out, dist)
    dist.destroy_process_group()
    return 0


def replicates_main(argv=None) -> int:
    """One replicate per GPU, no collective at all (the layout of exps/3-homo-real: six samples on one graph):

        python -m torch.distributed.run --nproc-per-node N -m pantas_amd.distributed replicates \\
            --out-dir DIR [--thr 20] [--gtf anno.gtf --call "--rca 3 --novel" [--no-aug-gfa]] GFA GAF1 GAF2 ...

    Rank r (LOCAL_RANK / RANK, WORLD_SIZE from the environment: under torchrun or any other launcher) parses the graph
    once, keeps it in the HBM of GPU r, and runs `augment` for GAFs r, r + N, ...: DIR/<gaf name>.aug.gfa; with --gtf also
    `call` on each result without the GFA text round trip: DIR/<gaf name>.events.csv (--no-aug-gfa: the events only, the
    augmented GFA between the two stages is not written).  No process group is created."""
    import argparse
    import sys
    import time
    from .call import CSV_HEADER  # noqa: F401  (the renderer writes it)
    from .engine import Graph, graph_cache_path
    ap = argparse.ArgumentParser(prog="pantas_amd.distributed replicates")
    ap.add_argument("--out-dir", required=True)
    ap.add_argument("--thr", type=int, default=20)
    ap.add_argument("--gtf", default=None)
    ap.add_argument("--call", default="", help="options of `pantas call` as one string, e.g. \"--rca 3 --novel\"")
    ap.add_argument("--no-aug-gfa", action="store_true", help="with --gtf: do not write the augmented GFA, only the events")
    ap.add_argument("GFA")
    ap.add_argument("GAF", nargs="+")
    a = ap.parse_args(argv)
    if a.no_aug_gfa and not a.gtf:
        ap.error("--no-aug-gfa without --gtf would write nothing")
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    names = [os.path.basename(p) for p in a.GAF]
    twice = sorted({n for n in names if names.count(n) > 1})
    if twice:      # sample1/reads.gaf + sample2/reads.gaf would write the same two files, maybe from two ranks at once
        ap.error(f"replicates with the same file name would overwrite each other's outputs in {a.out_dir}: {', '.join(twice)}")
    mine = a.GAF[rank::world]
    os.makedirs(a.out_dir, exist_ok=True)
    if not mine:
        return 0
    t0 = time.perf_counter()
    g = Graph(a.GFA, cache=graph_cache_path(a.GFA)).to_device(local)
    call_opts = None
    if a.gtf:
        from .call import build_parser
        call_opts = build_parser().parse_args(a.call.split() + ["-", "-"])
    for gaf in mine:
        base = os.path.join(a.out_dir, os.path.basename(gaf))
        t1 = time.perf_counter()
        if a.no_aug_gfa:
            bs, _ = g.augment(gaf, a.thr, -1)
        else:
            with open(base + ".aug.gfa", "wb") as f:
                bs, _ = g.augment(gaf, a.thr, f.fileno())
        t2 = time.perf_counter()
        msg = f"[rank {rank} gpu {local}] {gaf}: {bs.lines} lines, {bs.rejected} rejected, augment {t2 - t1:.2f} s"
        if call_opts is not None:
            g.call_prepare_fused(a.gtf, device=local)
            with open(base + ".events.csv", "wb") as f:
                cs = g.call(f.fileno(), 2, rca=call_opts.rca, novel=call_opts.novel, no_annotated=call_opts.annotated,
                            events=call_opts.events, isw=call_opts.isw, minintronsize=call_opts.minintronsize, junction=call_opts.junction)
            msg += f", call {time.perf_counter() - t2:.2f} s ({cs.events} events)"
        print(msg, file=sys.stderr)
    print(f"[rank {rank} gpu {local}] {len(mine)} replicate(s) in {time.perf_counter() - t0:.2f} s (graph parsed once)", file=sys.stderr)
    return 0


def main(argv=None) -> int:
    """torchrun entry:  python -m torch.distributed.run --nproc-per-node N -m pantas_amd.distributed GAF GFA [thr]"""
    import sys
    argv = sys.argv[1:] if argv is None else argv
    if argv and argv[0] == "replicates":
        return replicates_main(argv[1:])
    import torch
    import torch.distributed as dist
    if argv and argv[0] == "call":
        return call_main(argv[1:])
    gaf, gfa = argv[0], argv[1]
    thr = int(argv[2]) if len(argv) > 2 else 20
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # a rehearsal of N ranks on a box with ONE GPU (tests): every rank on cuda:0, gloo carrying the device tensors
    rehearsal = os.environ.get("PANTAS_DIST_REHEARSE_ON_ONE_GPU")
    if rehearsal:
        local = 0
    # stdout is the data channel (the augmented GFA): keep a private handle on it and point fd 1 at
    # stderr, because RCCL prints a version banner to stdout when the communicator comes up
    sys.stdout.flush()
    data_fd = os.dup(1)
    os.dup2(2, 1)
    torch.cuda.set_device(local)
    if rehearsal:
        dist.init_process_group(backend="gloo")
    else:
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local))
    if dist.get_rank() == 0:
        print("Read GFA", file=sys.stderr)
    be = LibBackend(gfa, local, dist)
    if dist.get_rank() == 0:
        print("Augmentation by GAF alignments", file=sys.stderr)
    sys.stdout.flush()
    tot = augment_sharded(be, gaf, thr, data_fd if dist.get_rank() == 0 else -1, dist,
                          torch.device("cuda", local))
    os.close(data_fd)
    if dist.get_rank() == 0:
        print(f"Rejected alignments: {tot['rejected']}", file=sys.stderr)
        print("Annotating GFA", file=sys.stderr)
    dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
