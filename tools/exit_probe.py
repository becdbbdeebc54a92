#!/usr/bin/env python3
"""What process exit costs after `os._exit` (the tail of both command lines at S4): child processes that hold one kind of
resource each and exit; the parent times from the child's "ready" line to its exit.

    python tools/exit_probe.py [--gb 16]"""
import argparse
import json
import os
import subprocess
import sys
import time

CHILD = r'''
import os, sys, mmap, time
kind, gb = sys.argv[1], float(sys.argv[2])
n = int(gb * 1e9)
keep = []
if kind == "hbm":
    import torch
    keep.append(torch.empty(n, dtype=torch.uint8, device="cuda:0").zero_())
    torch.cuda.synchronize()
elif kind == "pinned":
    import torch
    torch.cuda.init()
    keep.append(torch.empty(n, dtype=torch.uint8).pin_memory())
elif kind == "anon4k":
    m = mmap.mmap(-1, n)
    m.madvise(mmap.MADV_NOHUGEPAGE)
    for off in range(0, n, 4096): m[off] = 1
    keep.append(m)
elif kind == "anon_thp":
    m = mmap.mmap(-1, n + (2 << 20))
    m.madvise(mmap.MADV_HUGEPAGE)
    for off in range(0, n, 4096): m[off] = 1
    keep.append(m)
elif kind == "tmpfs_map":
    path = f"/dev/shm/exit_probe_{os.getpid()}"
    with open(path, "wb") as f:
        f.truncate(n)
    fd = os.open(path, os.O_RDWR)
    m = mmap.mmap(fd, n)
    for off in range(0, n, 4096): m[off] = 1
    keep.append(m)
    os.unlink(path)
elif kind == "hip_context":
    import torch
    torch.zeros(1, device="cuda:0")
    torch.cuda.synchronize()
print("ready", flush=True)
sys.stdin.readline()
os._exit(0)
'''


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gb", type=float, default=16)
    a = ap.parse_args()
    out = {}
    for kind in ("none", "hip_context", "hbm", "pinned", "anon4k", "anon_thp", "tmpfs_map"):
        p = subprocess.Popen([sys.executable, "-c", CHILD, kind, str(a.gb)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True)
        p.stdout.readline()
        t0 = time.perf_counter()
        p.stdin.write("\n")
        p.stdin.flush()
        p.wait()
        out[kind] = round(time.perf_counter() - t0, 3)
        print(kind, out[kind], flush=True)
    print(json.dumps({"gb": a.gb, "exit_s": out}))


if __name__ == "__main__":
    main()
