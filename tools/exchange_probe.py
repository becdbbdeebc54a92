#!/usr/bin/env python3
"""Counter exchange on the GPU box: block size sweep of allreduce_counters against the dense all-reduce.

    python tools/exchange_probe.py --scale s4                      one rank, RCCL: touched fractions, kernel times
    python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 \
        tools/exchange_probe.py --scale s2 --backend gloo          two ranks sharing cuda:0 over gloo: the real
                                                                   exchange, result compared with the dense sum
Prints one JSON object per configuration on rank 0."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="s2")
    ap.add_argument("--backend", default="nccl")
    ap.add_argument("--lines", type=int, default=0)
    args = ap.parse_args()
    import torch
    import torch.distributed as dist
    from bench import SCALES
    from pantas_amd.distributed import allreduce_counters
    from pantas_amd.engine import Graph, Synth
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29612")
    os.environ.setdefault("RANK", "0")
    os.environ.setdefault("WORLD_SIZE", "1")
    dev = 0 if args.backend == "gloo" else int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(dev)
    if args.backend == "nccl":
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", dev))
    else:
        dist.init_process_group(backend="gloo")
    desc, kw = SCALES[args.scale]
    kw = dict(kw)
    if args.lines:
        kw["n_lines"] = args.lines
    syn = Synth(seed=7, line_offset=rank * kw["n_lines"], **kw)
    g = Graph(text=syn.gfa).to_device(dev)
    g.augment_reset()
    b = g.load_gaf_mem(syn.gaf, first_line=rank * kw["n_lines"])
    g.launch(b)

    class BE:
        def counters_tensor(self): return g.counters_tensor(dev)
        def touched_bitmap(self, bw): return g.touched_bitmap(bw, dev)
        def counters_pack(self, bm, n): return g.counters_pack(bm, n, dev)
        def counters_unpack(self): g.counters_unpack()

    mine = g.counters_tensor(dev).clone()
    dense = mine.clone()
    dist.all_reduce(dense)
    torch.cuda.synchronize()
    for block in (0, 8, 16, 32, 64, 16):
        g.counters_tensor(dev).copy_(mine)
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        info = allreduce_counters(BE(), dist, block)
        torch.cuda.synchronize()
        info["wall_ms"] = (time.perf_counter() - t0) * 1e3
        info["equal_to_dense"] = bool(torch.equal(g.counters_tensor(dev), dense))
        info.update(scale=args.scale, world=world, backend=args.backend)
        if rank == 0:
            print(json.dumps(info), flush=True)
        assert info["equal_to_dense"]
    g.finish()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
