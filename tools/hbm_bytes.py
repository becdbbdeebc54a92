#!/usr/bin/env python3
"""Exact memory-side bytes of the dominant augment kernel from the L2's request counters (tools/pmc_hbm.sh + tools/pmc_hbm2.sh).

    python tools/hbm_bytes.py gpurun_out/hbm_r5/summary.txt gpurun_out/hbm_r5b/summary2.txt --text-bytes N --ms 1.62 > profiles/r5_traffic.json

Reads: TCC_EA0_RDREQ counts requests of 32, 64 and 128 bytes (TCC_EA0_RDREQ_32B / _64B / _128B): bytes = 32 n32 + 64 n64 + 128 n128.
FETCH_SIZE is RDREQ * 64 B whatever the size (MI355X_MICROARCH.md "HBM"), hence the guide's "double it" for streaming reads -- a rule of
thumb this replaces.  Writes: WRITE_SIZE is exact (64-byte and 32-byte requests: 64 n64 + 32 (n - n64)).  The calibration row is
count_newlines_kernel of the same run, which streams the GAF text exactly once (--text-bytes).
"""
import argparse
import json
import re
import sys

HBM_PEAK = 8.0e12


def parse(path):
    out, k = {}, None
    for line in open(path):
        m = re.match(r"== (.*?)\s+\((\d+) dispatches\)", line)
        if m:
            k = m.group(1).replace("void ", "")
            out.setdefault(k, {})
            continue
        m = re.match(r"\s+(\S+)\s+avg\s+([\d.]+)\s+min\s+([\d.]+)\s+max\s+([\d.]+)", line)
        if m and k:
            out[k][m.group(1)] = float(m.group(2))
    return out


def find(d, needle):
    ks = [k for k in d if needle in k]
    return d[ks[0]] if ks else {}


def kernel_bytes(c):
    n, n32, n64, n128 = (c.get(f"TCC_EA0_RDREQ{s}_sum") for s in ("", "_32B", "_64B", "_128B"))
    r = {"rdreq": n, "rdreq_32B": n32, "rdreq_64B": n64, "rdreq_128B": n128}
    if None not in (n, n32, n64, n128):
        r["rdreq_other"] = n - n32 - n64 - n128
        r["read_bytes_exact"] = 32 * n32 + 64 * n64 + 128 * n128 + 64 * max(0.0, r["rdreq_other"])
    w, w64 = c.get("TCC_EA0_WRREQ_sum"), c.get("TCC_EA0_WRREQ_64B_sum")
    r.update({"wrreq": w, "wrreq_64B": w64, "atomics_to_memory": c.get("TCC_EA0_ATOMIC_sum")})
    if w is not None and w64 is not None:
        r["write_bytes_exact"] = 64 * w64 + 32 * (w - w64)
    if "FETCH_SIZE" in c:
        r["fetch_size_bytes"] = c["FETCH_SIZE"] * 1024
    if "WRITE_SIZE" in c:
        r["write_size_bytes"] = c["WRITE_SIZE"] * 1024
    for k in ("TCC_REQ_sum", "TCC_HIT_sum", "TCC_MISS_sum", "TCC_EA0_RDREQ_DRAM_sum", "TCC_EA0_WRREQ_DRAM_sum"):
        if k in c:
            r[k.lower()[:-4]] = c[k]
    if "TCC_REQ_sum" in c and "TCC_HIT_sum" in c and c["TCC_REQ_sum"]:
        r["l2_hit_rate"] = c["TCC_HIT_sum"] / c["TCC_REQ_sum"]
    return r


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("summaries", nargs="+")
    ap.add_argument("--text-bytes", type=float, default=0)
    ap.add_argument("--ms", type=float, default=0, help="average duration of the dominant kernel in the bench run")
    ap.add_argument("--scale", default="s4")
    ap.add_argument("--kernel", default="augment_short_p_kernel")
    a = ap.parse_args()
    d = {}
    for p in a.summaries:
        for k, v in parse(p).items():
            d.setdefault(k, {}).update(v)
    dom = kernel_bytes(find(d, a.kernel))
    cal = kernel_bytes(find(d, "count_newlines_kernel"))
    res = {"kernel": [k for k in d if a.kernel in k][0], "counters": dom}
    if "read_bytes_exact" in dom and "write_bytes_exact" in dom:
        res["traffic_bytes"] = dom["read_bytes_exact"] + dom["write_bytes_exact"]
        res["traffic_rule"] = "32 n32 + 64 n64 + 128 n128 read requests + 64 n64 + 32 (n - n64) write requests (TCC_EA0_*), per launch"
    if "fetch_size_bytes" in dom and "write_size_bytes" in dom:
        res["traffic_bytes_doubled_read_rule"] = 2 * dom["fetch_size_bytes"] + dom["write_size_bytes"]
        res["traffic_bytes_uncorrected"] = dom["fetch_size_bytes"] + dom["write_size_bytes"]
    if a.ms:
        res["kernel_ms"] = a.ms
        for k in ("traffic_bytes", "traffic_bytes_doubled_read_rule", "traffic_bytes_uncorrected"):
            if k in res:
                res[k.replace("traffic_bytes", "hbm_frac")] = res[k] / (a.ms * 1e-3) / HBM_PEAK
    if a.text_bytes and cal.get("rdreq"):
        res["calibration"] = {"kernel": "count_newlines_kernel (streams the GAF text once, 16 bytes per lane)", "text_bytes": a.text_bytes,
                              "counters": cal, "bytes_per_read_request": a.text_bytes / cal["rdreq"],
                              "read_bytes_exact_over_text": cal.get("read_bytes_exact", 0) / a.text_bytes if "read_bytes_exact" in cal else None}
    json.dump({a.scale: res}, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()
