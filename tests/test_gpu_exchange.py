"""Block-sparse counter exchange (pantas_amd/csrc/exchange.hip, pantas_amd/distributed.py allreduce_counters).

Two read shards of one GAF are walked into two device graphs on the one GPU of the box; the exchange kernels run
through the C ABI on both, and torch ops stand in for the two collectives (concatenation for the all-gather, an
addition for the all-reduce).  Result: both graphs hold counts(A) + counts(B) = counts(whole), bit for bit, at
every block size.  A one-rank RCCL group then runs the product's own allreduce_counters.

The checks run in child processes that import torch BEFORE the library: torch bundles its own HIP runtime, and a
process that loads /opt/rocm's first (as the other GPU tests do) cannot initialise torch.cuda afterwards."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

PARAMS = dict(seed=41, n_genes=1200, n_lines=300_000, base_id=10**7, intron_long_max=4000, n_chrom=3, zipf_a=1.0,
              p_indel=0.05, p_novel_skip=0.03)



def run_child(code, timeout=600, env=None):
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=timeout, env={**os.environ, **(env or {})})
    if r.returncode != 0:
        print(r.stdout[-3000:])
        print(r.stderr[-6000:], file=sys.stderr)
    assert r.returncode == 0
    return r.stdout


CHECK = r"""
import ctypes as C, sys
import numpy as np
import torch
sys.path.insert(0, ROOT)
from pantas_amd.engine import Graph, PantasError, Synth
torch.cuda.set_device(0)
syn = Synth(**PARAMS)
addr, size = syn.gaf
raw = np.ctypeslib.as_array((C.c_char * size).from_address(addr)).view(np.uint8)
half = int(np.flatnonzero(raw[size // 2:] == 10)[0]) + size // 2 + 1
lines_a = int((raw[:half] == 10).sum())
ga, gb, gw = (Graph(text=syn.gfa).to_device(0) for _ in range(3))
parts = [(addr, half, 0), (addr + half, size - half, lines_a), (addr, size, 0)]

def run_shard(g, part):
    a, n, first = part
    g.augment_reset()
    b = g.load_gaf_mem((a, n), first_line=first)
    g.launch(b)
    return b

for block in (8, 16, 32, 64):
    keep = [run_shard(g, p) for g, p in zip((ga, gb, gw), parts)]
    ca, cb, cw = (g.counters_tensor(0) for g in (ga, gb, gw))
    want = (ca + cb).cpu()
    assert torch.equal(want, cw.cpu())                      # shard additivity (the kernels themselves)
    bms = [g.touched_bitmap(block, 0) for g in (ga, gb)]
    for g_c, bm in zip((ca, cb), bms):                      # the bitmap is exactly "block holds a non-zero word"
        pad = torch.zeros(bm.numel() * 32 * block, dtype=torch.int32, device="cuda")
        pad[:g_c.numel()] = g_c
        bits = (pad.view(-1, block) != 0).any(dim=1).cpu().numpy()
        got = np.unpackbits(bm.cpu().numpy().view(np.uint8), bitorder="little").astype(bool)
        assert (bits == got).all()
    gathered = torch.cat(bms).contiguous()                  # stands in for all_gather_into_tensor
    torch.cuda.synchronize()
    sa = ga.counters_pack(gathered, 2, 0)
    sb = gb.counters_pack(gathered, 2, 0)
    assert sa.numel() == sb.numel() and sa.numel() % block == 0 and 0 < sa.numel() < ca.numel()
    total = sa + sb                                         # stands in for all_reduce(sum)
    sa.copy_(total)
    sb.copy_(total)
    torch.cuda.synchronize()
    ga.counters_unpack()
    gb.counters_unpack()
    assert torch.equal(ga.counters_tensor(0).cpu(), want)
    assert torch.equal(gb.counters_tensor(0).cpu(), want)
    print("block", block, "touched", sa.numel() / ca.numel())
    del keep

def raises(f):
    try:
        f()
    except PantasError:
        return True
    return False

assert raises(lambda: ga.touched_bitmap(12, 0))
ga.touched_bitmap(16, 0)
ga.counters_pack(ga.touched_bitmap(16, 0), 1, 0)
ga.counters_unpack()
assert raises(lambda: ga.counters_unpack())                  # nothing packed any more
print("ok")
"""


def test_block_sparse_sum_equals_dense():
    out = run_child(f"ROOT = {ROOT!r}\nPARAMS = {PARAMS!r}\n" + CHECK)
    assert out.strip().endswith("ok"), out


TRACKED = r"""
import ctypes as C, os, sys
import numpy as np
import torch
sys.path.insert(0, ROOT)
from pantas_amd.engine import Graph, Synth
torch.cuda.set_device(0)
os.environ["PANTAS_EXCHANGE_VERIFY"] = "1"        # touched_bitmap also reads the counters and refuses a kept bitmap that misses a block
syn = Synth(p_two_indels=0.1, **PARAMS)            # (records of the general-walk class among them)
plain, kept = Graph(text=syn.gfa).to_device(0), Graph(text=syn.gfa).to_device(0)
kept.track_touched(True)
for variant in ("120", "110", "101", "0"):
    os.environ["PANTAS_MAIN_KERNEL"] = variant
    out = []
    for g in (plain, kept):
        g.augment_reset()
        b = g.load_gaf_mem(syn.gaf)
        g.launch(b)
        g.launch(b)
        out.append(b)
    cp, ck = plain.counters_tensor(0), kept.counters_tensor(0)
    assert torch.equal(cp.cpu(), ck.cpu())                  # marking changes no counter
    for block in (8, 16, 32, 64):
        a = np.unpackbits(plain.touched_bitmap(block, 0).cpu().numpy().view(np.uint8), bitorder="little").astype(bool)    # by reading the counters
        k = np.unpackbits(kept.touched_bitmap(block, 0).cpu().numpy().view(np.uint8), bitorder="little").astype(bool)     # the kernels' own, folded
        assert not (a & ~k).any(), (variant, block)          # every block that holds a count is in the kept bitmap ...
        assert k.sum() <= a.sum() * 1.02 + 8, (variant, block, int(a.sum()), int(k.sum()))      # ... and hardly anything else
        # and the exchange built on it leaves the same counters: one "rank", pack + unpack
        st = kept.counters_pack(kept.touched_bitmap(block, 0), 1, 0)
        assert st.numel() > 0
        kept.counters_unpack()
        assert torch.equal(kept.counters_tensor(0).cpu(), cp.cpu())
    print("variant", variant, "ok")
    del out
# tracking asked for in the middle of a run only starts at the next reset: until then the counters are read
late = Graph(text=syn.gfa).to_device(0)
late.augment_reset()
b = late.load_gaf_mem(syn.gaf)
late.launch(b)
late.track_touched(True)
late.launch(b)
a = late.touched_bitmap(16, 0).cpu()
late.augment_reset(); b = late.load_gaf_mem(syn.gaf); late.launch(b); late.launch(b)
assert torch.equal(a, late.touched_bitmap(16, 0).cpu())
print("ok")
"""


def test_bitmap_kept_by_the_kernels_covers_every_touched_block():
    """round 5 (pantas_augment_track_touched): every site of every main-kernel variant that adds to a counter sets its block's bit --
    checked against the bitmap made by reading the counters, at every block size, with the library's own cross-check switched on"""
    out = run_child(f"ROOT = {ROOT!r}\nPARAMS = {PARAMS!r}\n" + TRACKED)
    assert out.strip().endswith("ok"), out


def test_allreduce_counters_one_rank_rccl(tmp_path):
    """the product's exchange routine over a real RCCL communicator (one rank: the sum is the identity)"""
    code = f"""
import os, sys
sys.path.insert(0, {ROOT!r})
import numpy as np, torch, torch.distributed as dist
from pantas_amd.engine import Graph, Synth
from pantas_amd.distributed import allreduce_counters
torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=torch.device("cuda", 0))
syn = Synth(**{PARAMS!r})
g = Graph(text=syn.gfa).to_device(0)
g.augment_reset()
b = g.load_gaf_mem(syn.gaf)
g.launch(b)
class BE:
    def counters_tensor(self): return g.counters_tensor(0)
    def touched_bitmap(self, bw): return g.touched_bitmap(bw, 0)
    def counters_pack(self, bm, n): return g.counters_pack(bm, n, 0)
    def counters_unpack(self): g.counters_unpack()
before = g.counters_tensor(0).cpu()
info = allreduce_counters(BE(), dist, 16)
assert info["mode"] == "block-sparse", info
assert info["exchanged_bytes"] < info["counter_bytes"] // 2, info
assert torch.equal(g.counters_tensor(0).cpu(), before)
info = allreduce_counters(BE(), dist, 0)
assert info["mode"] == "dense"
assert torch.equal(g.counters_tensor(0).cpu(), before)
g.finish()
dist.destroy_process_group()
print("ok", info)
"""
    run_child(code, env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29577", "RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})


def test_two_ranks_share_the_gpu_over_gloo():
    """two real ranks (both on cuda:0, gloo carrying the device tensors): the exchange at every block size
    against the dense all-reduce of the same counters (tools/exchange_probe.py asserts equality)"""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29613", os.path.join(ROOT, "tools", "exchange_probe.py"), "--scale", "s2", "--lines", "400000",
                        "--backend", "gloo"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    import json
    rows = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(rows) == 6 and all(x["equal_to_dense"] and x["world"] == 2 for x in rows)
    assert any(x["mode"] == "block-sparse" for x in rows)


def test_bench_sharded_path_rehearsal():
    """bench.py's N > 1 code path (per-rank batch, the block-sparse sum inside the timed region, MAX over ranks, one JSON
    line from rank 0) with two ranks on the one GPU over gloo -- what the 8-GPU run does, minus RCCL."""
    import json
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", "29617", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--scale", "s2", "--lines", "300000",
                        "--steps", "3", "--warmup", "1", "--no-cpu", "--rehearse-on-one-gpu"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [x for x in r.stdout.splitlines() if x.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["steps"] == 3 and d["unit"] == "reads/s"
    assert d["allreduce"]["mode"] in ("block-sparse", "dense") and d["allreduce_ms"] > 0
    assert d["value"] > 0 and d["roofline"]["frac"] > 0 and d["call"]["events"] > 0


ABI_COLLECTIVE = r"""
import ctypes as C, io, os, sys, tempfile
import numpy as np
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
from pantas_amd import _lib
from pantas_amd.engine import Graph, Synth
import augment_oracle

# a communicator of ONE rank made with the system's RCCL, no torch in the process
rccl = C.CDLL("librccl.so.1", mode=C.RTLD_GLOBAL)
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
syn = Synth(**PARAMS)
g = Graph(text=syn.gfa).to_device(0)
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0

def counters():
    t = g.counters_numpy()
    return t.copy()

g.augment_reset()
b = g.load_gaf_mem(syn.gaf)
g.launch(b)
before = counters()
assert before.any()
# (1) entry points found in the process (the RCCL above was loaded RTLD_GLOBAL), block-sparse
st = g.counters_allreduce(comm.value, block_words=16)
assert (st.mode, st.world, st.block_words) == (1, 1, 16) and 0 < st.exchanged_bytes < st.counter_bytes, (st.mode, st.world, st.exchanged_bytes, st.counter_bytes)
assert np.array_equal(counters(), before)          # a sum over one rank
# (2) the same through an explicit table of entry points, every block size and dense
api = _lib.RcclApi()
api.all_reduce = _lib._ALL_REDUCE(("ncclAllReduce", rccl))
api.all_gather = _lib._ALL_GATHER(("ncclAllGather", rccl))
api.comm_count = _lib._COMM_COUNT(("ncclCommCount", rccl))
for bw in (8, 32, 64, 0):
    st = g.counters_allreduce(comm.value, block_words=bw, api=api)
    assert st.world == 1 and st.mode == (1 if bw else 0), (bw, st.mode)
    if bw == 0:
        assert st.exchanged_bytes == st.counter_bytes
    assert np.array_equal(counters(), before)
# (3) and the run ends like any other: the oracle's bytes
g.finish()
with tempfile.TemporaryFile() as f:
    g.write_gfa(f.fileno())
    f.seek(0)
    got = f.read().decode()
pre = os.path.join(tempfile.mkdtemp(), "w")
syn.write(pre)
want = io.StringIO()
augment_oracle.augment(pre + ".gaf", pre + ".gfa", want)
assert got == want.getvalue()
assert rccl.ncclCommDestroy(comm) == 0
print("OK", st.counter_bytes)
"""


def test_counters_allreduce_through_the_c_abi():
    """pantas_counters_allreduce (round 3): bitmap all-gather -> pack -> all-reduce -> unpack on a caller's RCCL communicator,
    without Python's torch.distributed in the process -- one rank here (the box has one GPU): the collectives run, the sums
    of one rank are its own counters, the run's GFA is the oracle's"""
    small = dict(PARAMS, n_genes=300, n_lines=60_000)
    out = run_child(f"ROOT = {ROOT!r}\nPARAMS = {small!r}\n" + ABI_COLLECTIVE)
    assert out.strip().splitlines()[-1].startswith("OK")
