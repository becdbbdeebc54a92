"""The N>1 path (pantas_amd/distributed.py) over gloo, world_size 2 and 3, on CPU.

The exchange logic under test is the product's (shard split, node-flag all-gather, the dense
all-reduce, sparse merge, rank-0 writer); the kernels are replaced by the CPU harness
(tests/hostsim/hostsim_lib.cpp: the device walk compiled for the host), because there is no GPU
here.  The result must be byte-identical to the reference golden, like a single-GPU run."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import GOLDEN, ROOT

def free_port():
    import socket
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


BUILD = os.path.join(ROOT, "build", "hostsim")
CSRC = os.path.join(ROOT, "pantas_amd", "csrc")


@pytest.fixture(scope="session")
def hostsim_lib():
    os.makedirs(BUILD, exist_ok=True)
    out = os.path.join(BUILD, "libhostsim.so")
    srcs = [os.path.join(ROOT, "tests/hostsim/hostsim_lib.cpp")] + [os.path.join(CSRC, f) for f in
            ("graph.cpp", "common.cpp", "gaf.cpp", "augment_host.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-shared", "-fPIC", "-I", CSRC, "-o", out, *srcs, "-lpthread"])
    return out


WORKER = r'''
import ctypes as C, os, sys
import numpy as np
import torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pantas_amd.distributed import augment_sharded
from pantas_amd._lib import lib, check

class HostsimBackend:
    """CPU stand-in with the interface of pantas_amd.distributed.LibBackend"""
    def __init__(self, so, gfa):
        self.L = C.CDLL(so)
        self.L.hs_open.restype = C.c_void_p
        self.L.hs_counters.restype = C.POINTER(C.c_uint32)
        self.L.hs_flags_export.restype = C.c_uint64
        self.L.hs_sparse_count.restype = C.c_uint64
        self.h = C.c_void_p(self.L.hs_open(gfa.encode()))
        assert self.h
    def split(self, gaf, rank, world):
        b, e, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
        check(lib.pantas_gaf_split(gaf.encode(), rank, world, C.byref(b), C.byref(e), C.byref(n)))
        return b.value, e.value, n.value
    def load_shard(self, gaf, thr, begin, end, first_line):
        st = (C.c_uint64 * 3)()
        if begin == end:
            return dict(lines=0, rejected=0, del_path_nodes=0)
        assert self.L.hs_load_shard(self.h, gaf.encode(), thr, C.c_uint64(begin), C.c_uint64(end), C.c_uint64(first_line), st) == 0
        return dict(lines=st[0], rejected=st[1], del_path_nodes=st[2])
    def reserve(self, keys): pass
    def launch_rare(self): assert self.L.hs_rare(self.h) == 0
    def launch_main(self): assert self.L.hs_main(self.h) == 0
    def flags_export(self):
        n = self.L.hs_flags_export(self.h, None, C.c_uint64(0))
        a = np.zeros(n, dtype=np.uint32)
        self.L.hs_flags_export(self.h, a.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(n))
        return a
    def flags_import(self, codes):
        codes = np.ascontiguousarray(codes, dtype=np.uint32)
        self.L.hs_flags_import(self.h, codes.ctypes.data_as(C.POINTER(C.c_uint32)), C.c_uint64(codes.shape[0]))
    def counters_tensor(self):
        w = C.c_uint64()
        p = self.L.hs_counters(self.h, C.byref(w))
        a = np.ctypeslib.as_array(p, shape=(w.value,)).view(np.int32)
        return torch.from_numpy(a)          # shares memory: the all-reduce lands in the harness' array
    # numpy stand-ins for the exchange kernels (pantas_amd/csrc/exchange.hip), same contract
    def touched_bitmap(self, block_words):
        c = self.counters_tensor().numpy()
        self._bw = block_words
        per_word = 32 * block_words
        padded = np.zeros((c.shape[0] + 2047) // 2048 * 2048, dtype=np.int32)
        padded[:c.shape[0]] = c
        bits = (padded.reshape(-1, block_words) != 0).any(axis=1)
        return torch.from_numpy(np.packbits(bits.reshape(-1, 32), axis=1, bitorder="little").view(np.uint32).reshape(-1).view(np.int32).copy())
    def counters_pack(self, bitmaps, n):
        c = self.counters_tensor().numpy()
        u = np.bitwise_or.reduce(bitmaps.numpy().view(np.uint32).reshape(n, -1), axis=0)
        self._blocks = np.flatnonzero(np.unpackbits(u.view(np.uint8), bitorder="little"))
        padded = np.zeros(u.shape[0] * 32 * self._bw, dtype=np.int32)
        padded[:c.shape[0]] = c
        self._staging = torch.from_numpy(padded.reshape(-1, self._bw)[self._blocks].reshape(-1).copy())
        return self._staging
    def counters_unpack(self):
        c = self.counters_tensor().numpy()
        padded = np.zeros((c.shape[0] + 2047) // 2048 * 2048, dtype=np.int32)
        padded[:c.shape[0]] = c
        padded.reshape(-1, self._bw)[self._blocks] = self._staging.numpy().reshape(-1, self._bw)
        c[:] = padded[:c.shape[0]]
    def finish(self): assert self.L.hs_finish(self.h) == 0
    def sparse_export(self):
        n = self.L.hs_sparse_count(self.h)
        a = np.zeros((n, 4), dtype=np.uint64)
        self.L.hs_sparse_export(self.h, a.ctypes.data_as(C.POINTER(C.c_uint64)))
        return a
    def sparse_import(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.uint64)
        assert self.L.hs_sparse_import(self.h, rows.ctypes.data_as(C.POINTER(C.c_uint64)), C.c_uint64(rows.shape[0])) == 0
    def write_gfa(self, fd): assert self.L.hs_write(self.h, fd) == 0

so, gfa, gaf, out = sys.argv[2:6]
dist.init_process_group(backend="gloo")
be = HostsimBackend(so, gfa)
fd = os.open(out, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644) if dist.get_rank() == 0 else -1
tot = augment_sharded(be, gaf, 20, fd, dist)
if dist.get_rank() == 0:
    os.close(fd)
    open(out + ".rej", "w").write(str(tot["rejected"]))
    open(out + ".mode", "w").write(tot["exchange"]["mode"])
dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,world,block", [("synth_a", 2, 16), ("quirks", 2, 8), ("synth_b", 3, 64), ("micro", 2, 32), ("synth_a", 2, 0)])
def test_sharded_augment_gloo(hostsim_lib, tmp_path, case, world, block):
    import json
    d = os.path.join(GOLDEN, case)
    worker = tmp_path / "worker.py"
    worker.write_text(WORKER)
    out = str(tmp_path / "out.gfa")
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(worker), ROOT, hostsim_lib,
                        f"{d}/in.gfa", f"{d}/in.gaf", out], capture_output=True, text=True, timeout=300,
                       env={**os.environ, "OMP_NUM_THREADS": "1", "PANTAS_EXCHANGE_BLOCK": str(block)})
    assert r.returncode == 0, r.stderr[-3000:]
    # block 0 = the dense all-reduce; otherwise bitmap -> pack -> all-reduce -> unpack unless most blocks are touched
    assert open(out + ".mode").read() in (("dense",) if block == 0 else ("block-sparse", "dense"))
    assert open(out).read() == open(f"{d}/expected.aug.gfa").read()
    assert int(open(out + ".rej").read()) == json.load(open(f"{d}/expected.json"))["rejected"]


def test_gaf_split_covers_file_once():
    from pantas_amd._lib import check, lib
    gaf = os.path.join(GOLDEN, "synth_a", "in.gaf")
    size = os.path.getsize(gaf)
    text = open(gaf, "rb").read()
    for world in (1, 2, 3, 7, 64):
        pos, lines = 0, 0
        for r in range(world):
            b, e, n = C.c_uint64(), C.c_uint64(), C.c_uint64()
            check(lib.pantas_gaf_split(gaf.encode(), r, world, C.byref(b), C.byref(e), C.byref(n)))
            assert b.value == pos and e.value >= b.value
            assert b.value == 0 or text[b.value - 1:b.value] == b"\n"
            assert n.value == lines
            lines += text[b.value:e.value].count(b"\n")
            pos = e.value
        assert pos == size


CALL_WORKER = r'''
import io, os, subprocess, sys
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pantas_amd.distributed import call_sharded
exe, gfa, gtf, out = sys.argv[2:6]
flags = sys.argv[6:]
dist.init_process_group(backend="gloo")

def run_shard(rank, world):      # the CPU harness stands in for the call kernels
    r = subprocess.run([exe, *flags, "--shard", str(rank), str(world), "--no-header", gfa, gtf], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout

buf = io.StringIO()
call_sharded(run_shard, buf, dist)
if dist.get_rank() == 0:
    open(out, "w").write(buf.getvalue())
dist.destroy_process_group()
'''


@pytest.mark.parametrize("case,world", [("synth_a", 2), ("synth_b", 3)])
def test_sharded_call_gloo(tmp_path, case, world):
    import json
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from test_oracle_golden import check_rows
    import call_oracle
    exe = os.path.join(BUILD, "hostsim_call")
    if True:      # always rebuild: the binary may predate the --shard option
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-I", CSRC, "-o", exe, os.path.join(ROOT, "tests/hostsim/hostsim_call.cpp"),
                               os.path.join(CSRC, "graph.cpp"), os.path.join(CSRC, "common.cpp"), os.path.join(CSRC, "call_host.cpp"), "-lpthread"])
    d = os.path.join(GOLDEN, case)
    worker = tmp_path / "worker.py"
    worker.write_text(CALL_WORKER)
    gc = [c for c in json.load(open(f"{d}/expected.json"))["calls"] if "--novel" in c["flags"]][0]
    out = str(tmp_path / "events.csv")
    port = free_port()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), str(worker), ROOT, exe,
                        f"{d}/expected.aug.gfa", f"{d}/in.gtf", out, *gc["flags"]], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    rows = open(out).read().splitlines()
    assert rows[0] == call_oracle.HEADER
    check_rows(rows[1:], gc)


SHARED_WORKER = r'''
import os, sys, stat
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from pantas_amd.distributed import LibBackend
mode, log = sys.argv[2], sys.argv[3]
dist.init_process_group(backend="gloo")
rank = dist.get_rank()

class FakeGraph:                     # stands in for engine.Graph: records how each rank was asked to load
    def __init__(self, gfa, cache=None):
        if mode == "fail" and rank == 0:
            raise ValueError("no such graph")
        if cache is not None and rank == 0:
            st = os.stat(os.path.dirname(cache))
            assert stat.S_IMODE(st.st_mode) == 0o700 and st.st_uid == os.getuid()
            open(cache, "w").write("x")
        with open(f"{log}.{rank}", "w") as f:
            f.write(f"{cache} {cache is not None and os.path.exists(cache)}")

try:
    LibBackend._load_shared(FakeGraph, "some.gfa", None, dist)
    out = "ok"
except Exception as e:
    out = f"{type(e).__name__}: {e}"
with open(f"{log}.{rank}.result", "w") as f:
    f.write(out)
dist.destroy_process_group()
'''


@pytest.mark.parametrize("mode", ["ok", "fail"])
def test_one_rank_parses_private_cache_and_failure_reaches_every_rank(tmp_path, mode):
    """the cache of a multi-rank run lives in a 0700 directory made by rank 0 (path broadcast, removed after the load), and a
    rank 0 that cannot load the graph makes every rank raise instead of leaving them in a barrier"""
    w = tmp_path / "w.py"
    w.write_text(SHARED_WORKER)
    log = str(tmp_path / "log")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(free_port()), str(w), ROOT, mode, log], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    res = [open(f"{log}.{k}.result").read() for k in range(2)]
    if mode == "ok":
        assert res == ["ok", "ok"]
        p0, e0 = open(f"{log}.0").read().rsplit(" ", 1)
        p1, e1 = open(f"{log}.1").read().rsplit(" ", 1)
        assert p0 == p1 and e1 == "True" and "pantas_pgc_" in p0        # rank 1 was pointed at rank 0's file, and it was there
        assert not os.path.exists(os.path.dirname(p0))                  # ... and nothing is left behind
    else:
        assert res[0] == "ValueError: no such graph"
        assert res[1].startswith("RuntimeError: rank 0 could not load") and "no such graph" in res[1]
