#!/usr/bin/env python3
"""Wall time of the two command lines on a BASELINE-scale workload written to disk, the way the
reference's workflows time them (one process per stage under a clock:
exps/3-homo-real/workflow/rules/pantas.smk:164,181):

    python tools/e2e_scale.py --scale s4 [--lines N] [--dir /dev/shm] [--out result.json] [--cache]

Generates the scale's synthetic GFA / GTF / GAF (pantas_amd/csrc/synth.cpp, same recipe as bench.py),
writes them under --dir, then times `python -m pantas_amd.augment` and `python -m pantas_amd.call`
as child processes (process start, parsing, kernels, output writing all inside the clock), with
PANTAS_TIMING=1 so that the phase table of each stage lands in the result.  --blocks B writes the
read stream as B generated blocks of --lines each into ONE GAF file (s5 = 4 x 50 M).
"""
import argparse
import json
import os
import resource
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def timed(cmd, stdout_path, env):
    t0 = time.perf_counter()
    with open(stdout_path, "wb") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.PIPE, env=env, cwd=ROOT)
    dt = time.perf_counter() - t0
    err = r.stderr.decode(errors="replace")
    if r.returncode != 0:
        raise RuntimeError(f"{cmd}: rc {r.returncode}: {err[-3000:]}")
    ru = resource.getrusage(resource.RUSAGE_CHILDREN)
    return dt, err, ru.ru_maxrss / 1e6


def phases(stderr_text):
    out = []
    for line in stderr_text.splitlines():
        if line.startswith("[timing]"):
            out.append(line[len("[timing]"):].strip())
    return out


def same_file(p, q):
    if os.path.getsize(p) != os.path.getsize(q):
        return False
    with open(p, "rb") as f, open(q, "rb") as g:
        while True:
            x, y = f.read(1 << 26), g.read(1 << 26)
            if x != y:
                return False
            if not x:
                return True


def replicates_leg(a, pre, env, py):
    """N replicates in one process on one GPU: what the graph parse and the GFA text round trip cost per replicate."""
    import shutil
    d = pre + ".rep"
    os.makedirs(d, exist_ok=True)
    out = {"n": a.replicates}
    try:
        gafs = []
        for i in range(a.replicates):
            g = os.path.join(d, f"rep{i}.gaf")
            os.link(pre + ".gaf", g)          # same reads under N names: the launcher works per file
            gafs.append(g)
        for mode, extra in (("with_aug_gfa", []), ("events_only", ["--no-aug-gfa"])):
            t0 = time.perf_counter()
            r = subprocess.run([py, "-m", "pantas_amd.distributed", "replicates", "--out-dir", d, "--gtf", pre + ".gtf", "--call", a.flags, *extra,
                                pre + ".gfa", *gafs], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env, cwd=ROOT)
            dt = time.perf_counter() - t0
            err = r.stderr.decode(errors="replace")
            if r.returncode != 0:
                raise RuntimeError(f"replicates: rc {r.returncode}: {err[-3000:]}")
            o = {"wall_s": dt, "per_replicate_s": dt / a.replicates, "log": [l for l in err.splitlines() if l.startswith("[rank")]}
            o["events_csv_equals_cli"] = all(same_file(g + ".events.csv", pre + ".csv") for g in gafs[:1])
            if not extra:
                o["aug_gfa_equals_cli"] = all(same_file(g + ".aug.gfa", pre + ".aug.gfa") for g in gafs[:1])
            out[mode] = o
            for g in gafs:
                for ext in (".events.csv", ".aug.gfa"):
                    if os.path.exists(g + ext):
                        os.unlink(g + ext)
    finally:
        shutil.rmtree(d, ignore_errors=True)
    return out


def main():
    import bench
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", default="s2")
    ap.add_argument("--lines", type=int, default=0)
    ap.add_argument("--blocks", type=int, default=1)
    ap.add_argument("--dir", default="/dev/shm")
    ap.add_argument("--out", default=None)
    ap.add_argument("--cache", action="store_true", help="second augment run with PANTAS_GRAPH_CACHE=1 (cache written by the first)")
    ap.add_argument("--flags", default="--rca 3 --novel")
    ap.add_argument("--keep", action="store_true")
    ap.add_argument("--null-out", action="store_true", help="also time both stages with stdout on /dev/null (what the output file costs)")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--replicates", type=int, default=0,
                    help="also time N replicates of the GAF through ONE `pantas_amd.distributed replicates` process (graph parsed once, "
                         "call on the augment result without the GFA text round trip) and compare its files with the command lines'")
    ap.add_argument("--quant", type=int, default=0,
                    help="also time `pantas quant` over N + N replicates: the events CSV of this run under 2 N names (every event matches: the "
                         "join's worst case -- all rows kept, all rows printed)")
    ap.add_argument("--heartbeat", default=None, help="append a line to this file every 30 s (gpurun takes 7 silent minutes for a hang)")
    a = ap.parse_args()
    if a.heartbeat:
        import threading

        def beat():
            t0 = time.perf_counter()
            while True:
                with open(a.heartbeat, "a") as f:
                    f.write(f"{time.perf_counter() - t0:.0f} s\n")
                time.sleep(30)
        threading.Thread(target=beat, daemon=True).start()
    from pantas_amd.engine import Synth
    desc, kw, *rest = bench.SCALES[a.scale]
    if rest and a.blocks == 1:
        a.blocks = rest[0]
    kw = dict(kw)
    if a.lines:
        kw["n_lines"] = a.lines
    os.makedirs(a.dir, exist_ok=True)
    pre = os.path.join(a.dir, f"pantas_e2e_{a.scale}_{os.getpid()}")
    res = {"scale": a.scale, "workload": desc, "dir": a.dir, "cpu": os.cpu_count(), "call_flags": a.flags}
    t0 = time.perf_counter()
    gaf_lines = 0
    for b in range(a.blocks):
        syn = Synth(seed=a.seed, line_offset=b * kw["n_lines"], **kw)
        if b == 0:
            syn.write(pre)
            res["sizes"] = {"segments": syn.stats.nodes, "links": syn.stats.links, "junction_links": syn.stats.junction_links,
                            "gfa_bytes": syn.stats.gfa_bytes, "gtf_bytes": syn.stats.gtf_bytes}
        else:
            syn.write(pre + f".b{b}")
            with open(pre + ".gaf", "ab") as out, open(pre + f".b{b}.gaf", "rb") as src:
                while True:
                    chunk = src.read(1 << 26)
                    if not chunk:
                        break
                    out.write(chunk)
            for ext in (".gaf", ".gfa", ".gtf"):
                os.unlink(pre + f".b{b}" + ext)
        gaf_lines += syn.stats.gaf_lines
        syn.close()
    res["sizes"]["gaf_lines"] = gaf_lines
    res["sizes"]["gaf_bytes"] = os.path.getsize(pre + ".gaf")
    res["generate_s"] = time.perf_counter() - t0
    env = {**os.environ, "PANTAS_TIMING": "1"}
    env.pop("PANTAS_GRAPH_CACHE", None)
    py = sys.executable
    try:
        dt, err, rss = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug.gfa", env)
        res["augment"] = {"wall_s": dt, "reads_per_s": gaf_lines / dt, "phases": phases(err), "peak_rss_gb": rss,
                          "out_bytes": os.path.getsize(pre + ".aug.gfa"),
                          "stderr_tail": [l for l in err.splitlines() if not l.startswith("[timing]")][-6:]}
        if a.cache:
            env_c = {**env, "PANTAS_GRAPH_CACHE": "1"}
            dt0, _, _ = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug2.gfa", env_c)   # writes the cache
            dt1, err1, _ = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug2.gfa", env_c)
            res["augment_cached"] = {"first_wall_s": dt0, "wall_s": dt1, "phases": phases(err1)}
            os.unlink(pre + ".aug2.gfa")
        dt, err, rss = timed([py, "-m", "pantas_amd.call", *a.flags.split(), pre + ".aug.gfa", pre + ".gtf"], pre + ".csv", env)
        rows = 0
        with open(pre + ".csv", "rb") as f:
            while True:
                chunk = f.read(1 << 26)
                if not chunk:
                    break
                rows += chunk.count(b"\n")
        res["call"] = {"wall_s": dt, "events": rows - 1, "events_per_s": (rows - 1) / dt, "phases": phases(err), "peak_rss_gb": rss,
                       "out_bytes": os.path.getsize(pre + ".csv")}
        res["total_s"] = res["augment"]["wall_s"] + res["call"]["wall_s"]
        if a.replicates:
            res["replicates"] = replicates_leg(a, pre, env, py)
        if a.quant:
            t0 = time.perf_counter()
            r = subprocess.run([py, "-m", "pantas_amd.quantify", "--c1", *[pre + ".csv"] * a.quant, "--c2", *[pre + ".csv"] * a.quant],
                               stdout=open(pre + ".quant.csv", "wb"), stderr=subprocess.PIPE, env=env, cwd=ROOT)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(f"quant: rc {r.returncode}: {r.stderr.decode(errors='replace')[-2000:]}")
            res["quant"] = {"replicates_per_condition": a.quant, "input_rows": 2 * a.quant * res["call"]["events"], "wall_s": dt,
                            "rows_per_s": 2 * a.quant * res["call"]["events"] / dt, "out_bytes": os.path.getsize(pre + ".quant.csv"),
                            "what": "native join (csrc/quant.cpp) through `python -m pantas_amd.quantify`"}
            t0 = time.perf_counter()
            r = subprocess.run([py, "-m", "pantas_amd.remap", pre + ".quant.csv", pre + ".gtf"], stdout=open(pre + ".remap.csv", "wb"), stderr=subprocess.PIPE,
                               env=env, cwd=ROOT)
            dt = time.perf_counter() - t0
            if r.returncode != 0:
                raise RuntimeError(f"remap: rc {r.returncode}: {r.stderr.decode(errors='replace')[-2000:]}")
            n_rows = sum(buf.count(b"\n") for buf in iter(lambda f=open(pre + ".quant.csv", "rb"): f.read(1 << 26), b"")) - 1
            res["remap"] = {"input_rows": n_rows, "wall_s": dt, "rows_per_s": n_rows / dt, "out_bytes": os.path.getsize(pre + ".remap.csv"),
                            "phases": phases(r.stderr.decode(errors="replace")), "what": "native form (csrc/remap.cpp) through `python -m pantas_amd.remap`"}
            os.unlink(pre + ".remap.csv")
            os.unlink(pre + ".quant.csv")
        if a.null_out:
            dt, err, _ = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], "/dev/null", env)
            res["augment_to_dev_null"] = {"wall_s": dt, "phases": [p for p in phases(err) if "tokenize_gaf_device" not in p]}
            dt, err, _ = timed([py, "-m", "pantas_amd.call", *a.flags.split(), pre + ".aug.gfa", pre + ".gtf"], "/dev/null", env)
            res["call_to_dev_null"] = {"wall_s": dt, "phases": phases(err)}
    finally:
        if not a.keep:
            for ext in (".gaf", ".gfa", ".gtf", ".aug.gfa", ".csv", ".gfa.pgc", ".aug2.gfa"):
                try:
                    os.unlink(pre + ext)
                except OSError:
                    pass
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
