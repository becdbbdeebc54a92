#!/usr/bin/env python3
"""Benchmark of the augment(+call) hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scale s1|s2|s3|s4|s5|auto]

A "step" is one pass of the augment kernels (rare-key kernel, then the main kernels) over the tokenised GAF records of
one replicate share that are already resident in HBM (graph CSR and records are uploaded before the timed region).  With
N > 1 (launched by torch.distributed.run, one rank per GPU) every rank holds a replica of the graph and its own block of
the read stream of the same size (weak scaling: read shards), and the timed region ends with the ONE sum over the coverage
arrays that a sharded replicate needs: the ranks all-gather a bitmap of the counter blocks they touched and
all-reduce only the union's blocks (pantas_amd/csrc/exchange.hip).

`value` = records WALKED per second (lines the tokeniser filtered out -- low mapq, unmapped, high divergence -- never reach
the kernels and are not counted).  Because a real batch is bucketed once and launched once, the line also carries, as
first-class fields, the rate with the one-off locus bucketing inside the clock (`reads_per_s_incl_bucket`), the rate from
GAF text in host memory to counters (`reads_per_s_text_to_counters`: PCIe copy, GPU tokeniser, bucketing, kernels) and, at
N = 1, the END-TO-END wall time of the two command lines on the same workload written to disk (`e2e`: process start, GFA /
GAF / GTF parsing, kernels, GFA and CSV writing, timed the way the reference's workflows time them,
exps/3-homo-real/workflow/rules/pantas.smk:164,181), without and with the binary graph cache.

After the timed region rank 0 runs the call kernels on the accumulated coverage (fused path, no GFA text round trip) and
reports events/s beside it.  Rank 0 prints one JSON line.

Workloads are synthetic (pantas_amd/csrc/synth.cpp, SURVEY.md 8-d recipe) at the BASELINE.json scales: s2 = Drosophila
whole genome, s3 = human chr1, s4 = human whole genome (one replicate, 50 M reads), s5 = the same graph with 200 M reads
(four resident batches of 50 M); they go through the product's own GFA / GAF / GTF parsers.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0   # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)

_HUMAN = dict(n_genes=62000, n_lines=50_000_000, base_id=10**9, intron_long_max=4780, n_chrom=24, zipf_a=1.0,
              read_len=150, with_path=0)
SCALES = {
    # name: (description, synth parameters [, blocks of n_lines reads per GPU])
    "s1": ("chr4-like plumbing case", dict(n_genes=114, n_lines=500, base_id=10**6)),
    "s2": ("Drosophila whole-genome scale (exps/1-dm-sim), 1M read pairs",
           dict(n_genes=18000, n_lines=2_000_000, base_id=10**7, intron_long_max=320, n_chrom=6, zipf_a=1.0)),
    "s3": ("human chr1 + 1000G scale, 10M GAF lines",
           dict(n_genes=5500, n_lines=10_000_000, base_id=10**8, intron_long_max=4280, n_chrom=1, zipf_a=1.0,
                read_len=150, with_path=0)),
    "s4": ("human whole-genome scale (exps/3-homo-real layout), one replicate of 50M GAF lines", dict(_HUMAN)),
    "s5": ("human whole-genome scale, annotated+novel search over 200M GAF lines (four resident batches of 50M)", dict(_HUMAN), 4),
}
# reference / port ratios measured in the build container on ONE input (profiles/r1_e2e_build_container_big.json: 6 000 genes,
# 2 M lines, 48 483 junction links): what turns the port's rate on this box into an estimate for the reference on this box
CALIBRATION = {"source": "profiles/r1_e2e_build_container_big.json", "augment_ref_over_port": 61.35 / 58.95,
               "call_ref_seconds_per_J2": 1309.6 / 48483 ** 2, "call_flags": "--rca 3 --novel"}


def cpu_quota_cores():
    """cores the cgroup lets this process use at a time (cpu.max: quota / period), None when unlimited or unknown.  The GPU box shows
    256 cores and grants 16: every host-side rate of this file is a 16-core rate there."""
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] == "max":
                    return None
                return float(txt[0]) / float(txt[1])
            q = float(txt[0])
            if q <= 0:
                return None
            return q / float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
        except (OSError, ValueError, IndexError):
            pass
    return None


def host_ram_gb() -> float:
    try:
        with open("/proc/meminfo") as f:
            for line in f:
                if line.startswith("MemTotal:"):
                    return int(line.split()[1]) / 1e6
    except OSError:
        pass
    return 0.0


def cpu_baseline(scale_kw: dict, sample_lines: int, s4_lines: int, s4_junctions: int):
    """The oracle (a Python port that keeps the reference's dict-based data structures) on a bounded sample of THE SAME
    recipe as the GPU line: the scale's own parameters with fewer genes (a Python dict graph of the whole scale does not
    fit a run of seconds) and `sample_lines` reads."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import augment_oracle
    import call_oracle
    from pantas_amd.engine import Synth
    tmp = tempfile.mkdtemp(prefix="pantas_cpu_")
    pre = os.path.join(tmp, "sample")
    kw = dict(scale_kw)
    kw["n_genes"] = min(kw["n_genes"], 600)      # (the port's call takes ~4 ms per junction link: 600 genes keep it near 20 s)
    kw["n_chrom"] = min(kw.get("n_chrom", 1), 2)
    kw["n_lines"] = sample_lines
    kw["with_path"] = 1
    syn = Synth(seed=11, **kw)
    syn.write(pre)
    a = augment_oracle.Augmenter(20)
    a.load_gfa(pre + ".gfa")
    t0 = time.perf_counter()
    a.feed_file(pre + ".gaf")
    dt = time.perf_counter() - t0
    with open(pre + ".aug.gfa", "w") as f:
        a.emit(pre + ".gfa", f)
    t1 = time.perf_counter()
    rows = call_oracle.call(pre + ".aug.gfa", pre + ".gtf", None, rca=3, novel=True)
    dc = time.perf_counter() - t1
    try:
        cpp = cpp_baseline(kw, tmp)
    except Exception as e:      # a reported baseline: its absence must not take the GPU line down
        cpp = {"skipped": f"{type(e).__name__}: {e}"}
    try:
        cpp_call = cpp_call_baseline(scale_kw, tmp)
    except Exception as e:
        cpp_call = {"skipped": f"{type(e).__name__}: {e}"}
    shutil.rmtree(tmp, ignore_errors=True)
    port_rate = sample_lines / dt
    ref_rate = port_rate / CALIBRATION["augment_ref_over_port"]
    return {"value": port_rate, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": f"{sample_lines} GAF lines of the same synthetic recipe and parameters as the GPU line on a "
                      f"{kw['n_genes']}-gene instance ({syn.stats.nodes} segments), {dt:.1f} s in oracle/augment_oracle.py "
                      "(Python port of the reference's per-line loop; the rate does not depend on the graph size: dict lookups)",
            "cpp": cpp,
            "cpp_call": cpp_call,
            "call_events_per_s": len(rows) / dc,
            "call_sample": f"{len(rows)} events from {syn.stats.junction_links} junction links in {dc:.1f} s, oracle/call_oracle.py incl. "
                           "GFA/GTF parsing (indexed junction lookup: NOT the reference's O(J^2) scan)",
            # the reference itself cannot run on this box; its speed here = the port's speed here x the ratio measured where both ran
            "reference_estimate": {
                "how": "port rate on this box x (reference / port) measured in the build container on one input; call: the "
                       "reference's measured seconds per J^2 (its junction scan is quadratic, scripts/call.py:280-282)",
                "calibration": CALIBRATION,
                "augment_reads_per_s": ref_rate,
                "augment_s_for_this_workload": s4_lines / ref_rate,
                "call_s_for_this_workload": CALIBRATION["call_ref_seconds_per_J2"] * s4_junctions ** 2,
                "note": "an extrapolation, not a measurement: the reference's Python dicts would not hold a whole-genome graph in memory"}}


def cpp_baseline(kw: dict, tmp: str, lines: int = 6_000_000):
    """the same algorithm in plain C++ (oracle/augment_oracle.cpp, pinned to the reference's goldens by
    tests/test_oracle_cpp.py) on one core and on all of them.  The timed part is its per-line loop -- GAF text to counters in host
    memory, what `value` and `reads_per_s_text_to_counters` cover on the GPU side -- not the GFA load or the output."""
    import subprocess
    from pantas_amd.engine import Synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    exe = os.path.join(ROOT, "oracle", "bin", "augment_oracle_cpp")
    pre = os.path.join(tmp, "cpp_sample")
    kw = dict(kw)
    kw["n_lines"] = lines
    syn = Synth(seed=12, **kw)
    syn.write(pre)
    cores = os.cpu_count() or 1
    quota = cpu_quota_cores()
    many = cores if quota is None else max(1, min(cores, int(round(quota * 2))))      # two threads per granted core
    out = {"sample": f"{lines} GAF lines ({syn.stats.gaf_bytes / 1e9:.2f} GB of text) of the same recipe on the {kw['n_genes']}-gene instance "
                     f"({syn.stats.nodes} segments: its tables stay in cache, a whole-genome graph's would not)",
           "timed": "the per-line loop: GAF text -> parsed -> walked -> NC / IL / OL / RC tables in host memory (hash maps); GFA load, "
                    "merge of the per-thread tables and the output are listed beside it",
           "cores": cores, "cpu_quota_cores": quota}
    for name, t in (("one_thread", 1), ("all_cores", many)):
        r = subprocess.run([exe, "--threads", str(t), "--no-out", pre + ".gaf", pre + ".gfa"], capture_output=True, text=True, check=True)
        st = json.loads(r.stderr.strip().splitlines()[-1])
        out[name] = {"threads": t, "reads_per_s": st["lines"] / st["hot_loop_s"], "hot_loop_s": st["hot_loop_s"], "merge_s": st["merge_s"],
                     "load_gfa_s": st["load_gfa_s"], "reads_per_s_incl_merge": st["lines"] / (st["hot_loop_s"] + st["merge_s"])}
    for ext in (".gaf", ".gfa", ".gtf"):
        try:
            os.unlink(pre + ext)
        except OSError:
            pass
    return out


def cpp_call_baseline(scale_kw: dict, tmp: str, genes: int = 6000, lines: int = 2_000_000, flags=("--rca", "3", "--novel")):
    """SURVEY.md 8-d rows 2a / 2b for `call`: oracle/call_oracle.cpp (pinned to the reference's goldens by tests/test_oracle_cpp.py)
    on one core and on all of them, on an augmented GFA of the same recipe (made by the C++ augment oracle).  Two rates each:
    the three passes alone (what the call kernels do on the GPU) and the whole program incl. GFA / GTF parsing."""
    import subprocess
    from pantas_amd.engine import Synth
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle")], check=True)
    pre = os.path.join(tmp, "cpp_call_sample")
    kw = dict(scale_kw)
    kw["n_genes"] = min(kw["n_genes"], genes)
    kw["n_chrom"] = min(kw.get("n_chrom", 1), 2)
    kw["n_lines"] = lines
    kw["with_path"] = 1
    syn = Synth(seed=13, **kw)
    syn.write(pre)
    cores = os.cpu_count() or 1
    quota = cpu_quota_cores()
    many = cores if quota is None else max(1, min(cores, int(round(quota * 2))))
    with open(pre + ".aug.gfa", "wb") as f:
        subprocess.run([os.path.join(ROOT, "oracle", "bin", "augment_oracle_cpp"), "--threads", str(many), pre + ".gaf", pre + ".gfa"],
                       stdout=f, stderr=subprocess.DEVNULL, check=True)
    exe = os.path.join(ROOT, "oracle", "bin", "call_oracle_cpp")
    out = {"sample": f"`call {' '.join(flags)}` on a {kw['n_genes']}-gene instance of the same recipe ({syn.stats.nodes} segments, "
                     f"{syn.stats.junction_links} junction links) augmented with {lines} reads",
           "cores": cores, "cpu_quota_cores": quota}
    for name, t in (("one_thread", 1), ("all_cores", many)):
        r = subprocess.run([exe, "--threads", str(t), "--no-out", *flags, pre + ".aug.gfa", pre + ".gtf"], capture_output=True, text=True, check=True)
        st = json.loads(r.stderr.strip().splitlines()[-1])
        out[name] = {"threads": t, "events": st["rows"], "events_per_s_passes": st["rows"] / max(st["passes_s"], 1e-9),
                     "events_per_s_whole_program": st["rows"] / st["total_s"], "passes_s": st["passes_s"], "gfa_s": st["gfa_s"], "total_s": st["total_s"]}
    for ext in (".gaf", ".gfa", ".gtf", ".aug.gfa"):
        try:
            os.unlink(pre + ext)
        except OSError:
            pass
    return out


def e2e_leg(syns, scale, kw, flags="--rca 3 --novel", with_cache=True):
    """`pantas augment` + `pantas call` as child processes on the workload written to disk, wall time of each."""
    need = sum(s.stats.gaf_bytes for s in syns) * 1.0 + syns[0].stats.gfa_bytes * 3.3 + 2e9
    where = None
    for d in ("/dev/shm", os.environ.get("TMPDIR", "/tmp")):
        try:
            if shutil.disk_usage(d).free > need:
                where = d
                break
        except OSError:
            pass
    if where is None:
        return {"skipped": f"no place with {need / 1e9:.0f} GB free for the files"}
    tmp = tempfile.mkdtemp(prefix="pantas_e2e_", dir=where)
    pre = os.path.join(tmp, scale)
    res = {"dir": where, "call_flags": flags,
           "what": "wall time of `python -m pantas_amd.augment GAF GFA > aug.gfa` and `python -m pantas_amd.call FLAGS aug.gfa GTF > events.csv` "
                   "as child processes on files in " + where + " (process start, parsing, kernels, writing: all inside)"}
    try:
        t0 = time.perf_counter()
        syns[0].write(pre)
        for b, s in enumerate(syns[1:], 1):
            s.write(pre + f".b{b}")
            with open(pre + ".gaf", "ab") as out, open(pre + f".b{b}.gaf", "rb") as src:
                shutil.copyfileobj(src, out, 1 << 26)
            for ext in (".gaf", ".gfa", ".gtf"):
                try:
                    os.unlink(pre + f".b{b}" + ext)
                except OSError:
                    pass
        res["write_inputs_s"] = time.perf_counter() - t0
        lines = sum(s.stats.gaf_lines for s in syns)
        res["gaf_bytes"] = os.path.getsize(pre + ".gaf")
        res["gfa_bytes"] = os.path.getsize(pre + ".gfa")
        env = {**os.environ, "PANTAS_TIMING": "1"}
        env.pop("PANTAS_GRAPH_CACHE", None)

        def timed(cmd, out_path, env):
            t0 = time.perf_counter()
            with open(out_path, "wb") as f:
                r = subprocess.run(cmd, stdout=f, stderr=subprocess.PIPE, env=env, cwd=ROOT)
            dt = time.perf_counter() - t0
            err = r.stderr.decode(errors="replace")
            if r.returncode != 0:
                raise RuntimeError(f"{cmd}: rc {r.returncode}: {err[-2000:]}")
            return dt, [l[len("[timing]"):].strip() for l in err.splitlines() if l.startswith("[timing]")]
        py = sys.executable
        dt, ph = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug.gfa", env)
        res["augment"] = {"wall_s": dt, "reads_per_s": lines / dt, "out_bytes": os.path.getsize(pre + ".aug.gfa"), "phases": ph}
        if with_cache:
            env_c = {**env, "PANTAS_GRAPH_CACHE": "1"}
            dt0, _ = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug2.gfa", env_c)    # parses and writes the cache
            dt1, ph1 = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".aug2.gfa", env_c)
            res["augment_graph_cache"] = {"wall_s": dt1, "reads_per_s": lines / dt1, "first_run_writing_the_cache_s": dt0, "phases": ph1,
                                          "cache_bytes": os.path.getsize(pre + ".gfa.pgc")}
            os.unlink(pre + ".aug2.gfa")
            os.unlink(pre + ".gfa.pgc")
        dt, ph = timed([py, "-m", "pantas_amd.call", *flags.split(), pre + ".aug.gfa", pre + ".gtf"], pre + ".csv", env)
        rows = 0
        with open(pre + ".csv", "rb") as f:
            while True:
                chunk = f.read(1 << 26)
                if not chunk:
                    break
                rows += chunk.count(b"\n")
        res["call"] = {"wall_s": dt, "events": rows - 1, "events_per_s": (rows - 1) / dt, "out_bytes": os.path.getsize(pre + ".csv"), "phases": ph}
        res["total_s"] = res["augment"]["wall_s"] + res["call"]["wall_s"]
        if with_cache:
            res["total_graph_cache_s"] = res["augment_graph_cache"]["wall_s"] + res["call"]["wall_s"]
    except Exception as e:   # the bench line must still be printed
        res["error"] = str(e)[-1500:]
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    return res

def _sig(x, n=5):
    """numbers of the printed line rounded to n significant digits (the details file keeps them whole)"""
    if isinstance(x, float):
        return float(f"{x:.{n}g}")
    return x


def emit(out: dict, args) -> None:
    """The ONE JSON line of the contract, kept SHORT (the driver reads the tail of stdout: round 3's 22 KB line lost its head there
    and was recorded as unparsed).  Everything else -- phase lists of the command lines, per-launch times, the C++ baseline rows, the
    definitions -- goes to a details file (and to stderr), named in the line."""
    details_path = None
    for d in (os.path.join(ROOT, "gpurun_out"), tempfile.gettempdir()):
        try:
            os.makedirs(d, exist_ok=True)
            path = os.path.join(d, f"bench_details_n{out['n_gpus']}.json")
            with open(path, "w") as f:
                json.dump(out, f)
                f.write("\n")
            details_path = os.path.relpath(path, ROOT) if os.path.commonpath([os.path.abspath(path), ROOT]) == ROOT else path
            break
        except OSError:
            pass
    sys.stderr.write("[bench details] " + json.dumps(out) + "\n")
    sys.stderr.flush()
    rf, cb, call, e2e = out["roofline"], out.get("cpu_baseline"), out.get("call"), out.get("e2e")
    wl = out["config"]["workload"]
    line = collections.OrderedDict()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data"):
        line[k] = _sig(out[k], 6)
    line["config"] = {"workload": wl.split(";")[0] + f"; {out['gaf_lines_per_step'] // out['n_gpus']} GAF lines per GPU per step, "
                                                      f"{out['walked_records_per_step'] // out['n_gpus']} walked",
                      "parallelism": f"read shards x{out['n_gpus']}, graph replicated, one block-sparse RCCL all-reduce of the counters",
                      "steps": "enqueued back to back, collected" if str(out.get("step_mode", "")).startswith("the steps") else "each waited for"}
    line["roofline"] = {k: _sig(rf.get(k)) for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "hbm_frac_physical", "kernel")}
    line["roofline"]["kernel_ms"] = _sig(out["kernel_ms"]["augment_main_short_walk"] or out["kernel_ms"]["augment_main"])
    if cb is not None:
        c = {"value": _sig(cb["value"]), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"],
             "sample": f"{args.cpu_sample} GAF lines of the same recipe on a 600-gene instance, oracle/augment_oracle.py",
             "call_events_per_s": _sig(cb.get("call_events_per_s"))}
        for name, key in (("cpp", "reads_per_s"), ("cpp_call", "events_per_s_passes")):
            row = cb.get(name) or {}
            for leg in ("one_thread", "all_cores"):
                if leg in row:
                    c[f"{name}_{leg}"] = {"threads": row[leg]["threads"], key: _sig(row[leg][key])}
        line["cpu_baseline"] = c
    if call:
        line["call"] = {"events": call["events"], "events_per_s": _sig(call["events_per_s"]), "kernel_ms": _sig(call["kernel_ms"]),
                        "achieved_GBps": _sig(call.get("achieved_GBps"))}
    line["reads_per_s_incl_bucket"] = _sig(out["reads_per_s_incl_bucket"])
    line["reads_per_s_text_to_counters"] = _sig(out["reads_per_s_text_to_counters"])
    if out.get("allreduce_ms"):
        ex = out.get("allreduce") or {}
        line["allreduce_ms"] = _sig(out["allreduce_ms"])
        line["allreduce"] = {k: _sig(v) for k, v in ex.items() if isinstance(v, (int, float, str, bool)) and k in
                             ("mode", "block_words", "counter_bytes", "exchanged_bytes", "bitmap_kernel_ms", "pack_ms", "unpack_ms")}
        # per collective: [op, bytes (per rank for the all-gather), ms] -- rank 0's clock around each, inside allreduce_ms
        line["allreduce"]["collectives"] = [[c["op"], c.get("bytes", c.get("bytes_per_rank")), _sig(c["ms"], 4)] for c in ex.get("collectives", [])]
        line["world_size_seen_by_process_group"] = out["ranks"]["world_size_seen_by_process_group"]
        line["backend"] = out["ranks"]["backend"]
    if e2e:
        line["e2e_wall_s"] = ({"augment": _sig(e2e["augment"]["wall_s"]), "call": _sig(e2e["call"]["wall_s"]),
                               "call_events": e2e["call"]["events"]} if "call" in e2e else
                              {"error": (e2e.get("error") or e2e.get("skipped") or "")[:120]})
    line["details"] = details_path
    print(json.dumps(out if getattr(args, "full_line", False) else line), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # (a step is milliseconds; the kernel's first launches after the set-up run 5 % longer than its tenth -- clocks, TLBs --, so the
    # defaults are what the driver passes: 20 timed steps behind 5 warm-up steps)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--scale", default=os.environ.get("PANTAS_BENCH_SCALE", "auto"))
    ap.add_argument("--lines", type=int, default=0, help="override the scale's GAF lines per GPU (per batch)")
    ap.add_argument("--blocks", type=int, default=0, help="override the number of resident batches per GPU")
    ap.add_argument("--seed", type=int, default=7)
    ap.add_argument("--cpu-sample", type=int, default=800_000)
    ap.add_argument("--del-hot", type=float, default=0.0,
                    help="share of the deletion reads aimed at 1 000 fixed (node, position) sites: the skew of real indel variants")
    ap.add_argument("--p-general", type=float, default=0.0,
                    help="share of the mapped reads with TWO indels in their cs (synth p_two_indels): records of the general-walk class, "
                         "which the plain recipe never writes -- what an indel-rich GAF costs (kernel_ms.augment_main_general)")
    ap.add_argument("--full-line", action="store_true",
                    help="print the whole details object as the line (tools/*.sh read phase times from it); the default line is the "
                         "short one of the contract, with the details in gpurun_out/bench_details_n<N>.json")
    ap.add_argument("--sync-steps", action="store_true",
                    help="one host round trip per step (pantas_augment_launch) instead of steps enqueued back to back "
                         "(pantas_augment_launch_async, at most 64 in flight) and collected")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-call", action="store_true")
    ap.add_argument("--no-e2e", action="store_true")
    ap.add_argument("--no-e2e-cache", action="store_true")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="N > 1 ranks all on cuda:0 with gloo carrying the device tensors: exercises the sharded path on a "
                         "one-GPU box; the number it prints is not a scaling measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world > 1 and "PANTAS_THREADS" not in os.environ:
        os.environ["PANTAS_THREADS"] = str(max(4, (os.cpu_count() or 8) // world))
    dist = None
    # what the library asks the HIP runtime for when it is loaded (pantas_amd/csrc/capi.cpp ask_for_more_hardware_queues: the lanes of
    # the call passes on hardware queues of their own) -- here torch comes first and may start the runtime before the library is there
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
    import torch
    if world > 1:
        import torch.distributed as dist
        if args.rehearse_on_one_gpu:
            local_rank = 0
            torch.cuda.set_device(0)
            dist.init_process_group(backend="gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
    from pantas_amd.engine import Graph, Synth

    scale = args.scale
    if scale == "auto":
        scale = "s4" if host_ram_gb() >= 400 else "s2"
    desc, kw, *rest = SCALES[scale]
    blocks = args.blocks or (rest[0] if rest else 1)
    kw = dict(kw)
    if args.lines:
        kw["n_lines"] = args.lines
    if args.del_hot:
        kw["p_del_hot"] = args.del_hot
    if args.p_general:
        kw["p_two_indels"] = args.p_general
    n_lines = kw["n_lines"]
    t_gen = time.perf_counter()
    # every rank: the same graph (same seed), its own blocks of the read stream
    first_line = rank * blocks * n_lines
    syn = Synth(seed=args.seed, line_offset=first_line, **kw)
    t_gen = time.perf_counter() - t_gen
    g = Graph(text=syn.gfa).to_device(local_rank)
    gs = g.stats
    if world > 1 and os.environ.get("PANTAS_TRACK_TOUCHED", "1") != "0":
        g.track_touched(True)        # (the exchange that ends the replicate folds the kernels' own bitmap of touched blocks)
    g.augment_reset()
    keep_for_e2e = rank == 0 and world == 1 and not args.no_e2e
    syns = [syn]
    # The first tokenisation of a process pays for the scratch pool's growth (hipMalloc of the batch's buffers: milliseconds on most
    # leases, a quarter of a second on some); a run tokenises batch after batch out of the warm pool.  The batch is therefore tokenised
    # once untimed and dropped, the phase times of the line come from the second pass; the first call's are kept beside them.
    warm = g.load_gaf_mem(syn.gaf, first_line=first_line)
    first_call = {"tokenize_s": warm.stats.tokenize_seconds, "upload_s": warm.stats.upload_seconds, "line_index_s": warm.stats.tok_index_seconds,
                  "parse_s": warm.stats.tok_measure_seconds, "bucket_s": warm.stats.tok_bucket_seconds}
    warm.free()
    del warm
    batches = [g.load_gaf_mem(syn.gaf, first_line=first_line)]
    for b in range(1, blocks):
        more = Synth(seed=args.seed, line_offset=first_line + b * n_lines, gaf_only=1, **kw)
        batches.append(g.load_gaf_mem(more.gaf, first_line=first_line + b * n_lines))
        if keep_for_e2e:
            syns.append(more)
        else:
            more.close()
    stats = [b.stats for b in batches]

    def tot(name):
        return sum(getattr(s, name) for s in stats)
    lines, records = tot("lines"), tot("records")

    class _Exchange:   # what pantas_amd.distributed.allreduce_counters drives (zero-copy views of the library's buffers)
        def counters_tensor(self): return g.counters_tensor(local_rank)
        def touched_bitmap(self, bw): return g.touched_bitmap(bw, local_rank)
        def counters_pack(self, bm, n): return g.counters_pack(bm, n, local_rank)
        def counters_unpack(self): g.counters_unpack()
    exchange = None

    def sync():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    def step():
        """one pass over everything resident: (rare ms, main ms, short-walk ms)"""
        if len(batches) == 1:
            ls = g.launch(batches[0])      # returns after both kernels completed (HIP events on the launch stream)
            return ls.rare_ms, ls.main_ms, ls.main_short_ms
        # several batches: every rare-key kernel before the first main kernel (the library enforces it)
        r = sum(g.launch_rare(b) for b in batches)
        m = sum(g.launch_main(b) for b in batches)
        return r, m, 0.0

    warm_ms = [round(step()[1], 3) for _ in range(args.warmup)]
    if dist is not None:
        # warm-up of the two collectives of the counter exchange on buffers of their sizes (the library sets up channels and
        # protocols on first use); the counters themselves are not touched before the timed region
        dev = torch.device("cuda", local_rank)
        words = g.counters()[1]
        bm = torch.zeros(words // 512 + 1, dtype=torch.int32, device=dev)
        allbm = torch.empty(world * bm.numel(), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(allbm, bm)
        st = torch.zeros(max(1 << 20, words // 32), dtype=torch.int32, device=dev)
        dist.all_reduce(st)
        del bm, allbm, st
    sync()
    main_ms, rare_ms, short_ms = [], [], []
    # One resident batch: the K steps are enqueued back to back on the graph's stream (a streamed run launches batch after batch the
    # same way) and their HIP-event times read afterwards; with --sync-steps, or several resident batches, every launch is waited for.
    pipelined = len(batches) == 1 and not args.sync_steps
    t0 = time.perf_counter()
    if pipelined:
        done = 0
        while done < args.steps:
            k = min(64, args.steps - done)
            for _ in range(k):
                g.launch_async(batches[0])
            for r, m, s in g.launch_collect():
                rare_ms.append(r)
                main_ms.append(m)
                short_ms.append(s)
            done += k
    else:
        for _ in range(args.steps):
            r, m, s = step()
            rare_ms.append(r)
            main_ms.append(m)
            short_ms.append(s)
    allreduce_ms = 0.0
    if dist is not None:
        # the ONE sum over the coverage arrays that ends a read-sharded replicate: bitmap of touched blocks ->
        # all-reduce of the union's blocks (dense all-reduce when most blocks are touched)
        from pantas_amd.distributed import allreduce_counters
        t1 = time.perf_counter()
        exchange = allreduce_counters(_Exchange(), dist)
        torch.cuda.synchronize()
        allreduce_ms = (time.perf_counter() - t1) * 1e3
    sync()
    elapsed = time.perf_counter() - t0
    ranks = None
    if dist is not None:
        t = torch.tensor([elapsed], device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # proof of who took part: what the process group itself reports, and every rank's device
        mine = {"rank": dist.get_rank(), "local_rank": local_rank, "device": torch.cuda.get_device_name(local_rank),
                "device_index": torch.cuda.current_device(), "records": records}
        ranks = [None] * world
        dist.all_gather_object(ranks, mine)
        rec_t = torch.tensor([records, lines], dtype=torch.int64, device="cuda")
        dist.all_reduce(rec_t)
        records_all, lines_all = (int(x) for x in rec_t.tolist())
    else:
        records_all, lines_all = records, lines
    t_fin = time.perf_counter()
    g.finish()   # raises if a kernel flagged a line; downloads the counters
    t_fin = time.perf_counter() - t_fin

    call = None
    if rank == 0 and not args.no_call:
        t1 = time.perf_counter()
        g.call_prepare_fused(text=syn.gtf, device=local_rank)
        t_prep = time.perf_counter() - t1
        fd = os.open(os.devnull, os.O_WRONLY)
        # coverage has accumulated over every launch of the same batches: scale the read-count threshold with it,
        # so that the events are those of `--rca 3` on one pass over them
        rca = 3 * (args.steps + args.warmup)
        runs = []
        for _ in range(3):
            t1 = time.perf_counter()
            cs = g.call(fd, -1, rca=rca, novel=True)
            runs.append((time.perf_counter() - t1, cs))
        os.close(fd)
        wall, cs = min(runs, key=lambda x: x[0])
        # the passes run side by side since round 5 (PANTAS_CALL_SERIAL=1: in turn): their window on the device, not the sum of the lanes
        k_ms = cs.passes_ms or (cs.annotated_ms + cs.novel_ms + cs.around_ms)
        call = {"events": cs.events, "junctions": cs.junctions, "novel_links": cs.novel_links,
                "evaluated": cs.evaluated, "flags": f"--rca {rca} --novel (= --rca 3 per pass over the batch)",
                "events_per_s": cs.events / wall, "wall_s": wall, "kernel_ms": k_ms,
                "kernel_ms_by_pass": {"annotated": cs.annotated_ms, "novel_links": cs.novel_ms, "novel_around": cs.around_ms,
                                      "note": "lanes that overlap: their sum exceeds kernel_ms"},
                "junction_evals_per_s_kernel": cs.evaluated / (k_ms * 1e-3) if k_ms else None,
                # SURVEY 8-d "achieved bytes for completeness": one pass over the link flags and the CSR rows, 4 (V + A), plus a nominal
                # 64 bytes per evaluated item (a row of up to eight neighbours and four tag triples); the kernels are chains of dependent
                # loads over a few 10^5 items -- latency-bound, as the survey says: a fraction of a percent of the HBM roofline
                "alg_bytes": 4 * (gs.segments + gs.link_slots) + 64 * cs.evaluated,
                "achieved_GBps": (4 * (gs.segments + gs.link_slots) + 64 * cs.evaluated) / (k_ms * 1e-3) / 1e9 if k_ms else None,
                "prepare_s": t_prep, "tag_parse_s": cs.parse_seconds, "render_s": cs.render_seconds}

    if rank == 0:
        step_s = elapsed / args.steps
        ms_per_step = step_s * 1e3
        value = records_all / step_s
        avg_main = sum(main_ms) / len(main_ms)
        avg_short = sum(short_ms) / len(short_ms)
        alg_bytes, short_alg = tot("alg_bytes"), tot("short_alg_bytes")
        if avg_short > 0:
            # two main kernels: the short-walk kernel takes the records of its class -- the dominant one
            variant = os.environ.get("PANTAS_MAIN_KERNEL", "120")
            dom_name = {"120": "augment_short_p_kernel<512,2048,10,7,8,0>", "110": "augment_short_w_kernel<512,2048,10,7,8,1>"}.get(
                variant, f"augment_main_agg_kernel<short-walk class> (PANTAS_MAIN_KERNEL={variant})")
            dom_ms, dom_bytes, dom_recs = avg_short, short_alg, tot("short_records")
        else:
            dom_name, dom_ms, dom_bytes, dom_recs = "augment main kernels of all resident batches", avg_main, alg_bytes, records
        achieved = dom_bytes / (dom_ms * 1e-3) / 1e9
        traffic, traffic_file, traffic_rule = None, None, None
        for tf in ("r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
            try:
                tj = json.load(open(os.path.join(ROOT, "profiles", tf)))
                if scale in tj and not args.lines and not args.del_hot and not args.p_general:
                    traffic = tj[scale]["traffic_bytes"]     # measured in separate --pmc runs of this workload
                    traffic_file = tf
                    traffic_rule = tj[scale].get("traffic_rule", "2 * FETCH_SIZE + WRITE_SIZE")
                    break
            except (OSError, ValueError):
                pass
        tok_s, up_s, bucket_s = tot("tokenize_seconds"), tot("upload_seconds"), tot("tok_bucket_seconds")
        out = collections.OrderedDict()
        out.update({
            "metric": "augment GAF reads/s (+ call events/s) on synthetic spliced pangenome",
            "value": value, "unit": "reads/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"{scale}: {desc}; {gs.segments} segments, {gs.link_slots} links, "
                                   f"{kw['n_genes']} genes; {lines} GAF lines per GPU per step in {len(batches)} resident batch(es) "
                                   f"({records} walked, {tot('path_nodes') / max(1, records):.2f} nodes/line, "
                                   f"{tot('cs_ops') / max(1, records):.2f} cs ops/line), gene expression Zipf a={kw.get('zipf_a', 1.2)}"
                                   + (f", {args.del_hot:.0%} of the deletion reads on 1000 fixed sites" if args.del_hot else "")
                                   + (f", {args.p_general:.0%} of the mapped reads with two indels ({records - tot('short_records')} records of the "
                                      f"general-walk class per step)" if args.p_general else ""),
                       "parallelism": f"read shards x{world}, graph replicated, one block-sparse all-reduce of the coverage counters per replicate"},
            "value_counts": "records walked by the kernels per second, summed over ranks (GAF lines filtered out by the tokeniser are not counted)",
            "step_mode": ("the steps of the timed region are enqueued back to back on the stream (pantas_augment_launch_async, at most 64 in flight) "
                          "and collected; --sync-steps waits for every step") if pipelined else "every launch waited for (pantas_augment_launch)",
            "walked_records_per_step": records_all, "gaf_lines_per_step": lines_all,
            # first-class companions of `value`: what a batch that is bucketed once and launched once sees
            "reads_per_s_kernels_only": records_all / step_s,
            "reads_per_s_incl_bucket": records / (step_s + bucket_s) * world,
            "reads_per_s_text_to_counters": lines / (tok_s + up_s + step_s) * world,
            "reads_per_s_definitions": {
                "reads_per_s_kernels_only": "= value: resident, bucketed records; rare-key + main kernels",
                "reads_per_s_incl_bucket": "walked records / (step + the one-off locus bucketing and block ordering of the batch, "
                                           f"{bucket_s * 1e3:.1f} ms): a batch is bucketed once and launched once; phase times of the tokeniser are "
                                           "those of a process whose scratch pool is warm (tokenizer.first_call_of_the_process: the cold call's)",
                "reads_per_s_text_to_counters": "GAF lines / (PCIe copy of the text + GPU tokeniser incl. bucketing + step): from text in host memory "
                                                "to counters in HBM; never `value`"},
            "kernel_ms": {"augment_main": avg_main, "augment_main_short_walk": avg_short, "augment_main_general": avg_main - avg_short if avg_short else None,
                          "augment_rare": sum(rare_ms) / len(rare_ms),
                          "augment_main_min": min(main_ms), "augment_main_max": max(main_ms),
                          "augment_main_all": [round(x, 3) for x in main_ms],
                          # the very first launch after the upload runs several times longer (cold TLBs / caches over 20 GB of
                          # graph, batch and counters); rocprofv3's per-kernel average includes it, the timed steps do not
                          "augment_main_warmup": warm_ms},
            "allreduce_ms": allreduce_ms, "allreduce": exchange,
            "ranks": {"world_size_seen_by_process_group": dist.get_world_size() if dist is not None else 1,
                      "backend": dist.get_backend() if dist is not None else None, "per_rank": ranks},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         # what the memory system moved / (this run's kernel time) / peak: the physical fraction next to the nominal one
                         "hbm_frac_physical": (traffic / (dom_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                         "traffic_source": f"profiles/{traffic_file}: {traffic_rule}" if traffic_file else None,
                         "kernel": dom_name, "alg_bytes_per_launch": dom_bytes, "records_per_launch": dom_recs,
                         "whole_main_step": {"ms": avg_main, "alg_bytes": alg_bytes, "achieved_GBps": alg_bytes / (avg_main * 1e-3) / 1e9},
                         "traffic_note": "HBM bytes per launch from profiles/r*_traffic.json (separate rocprofv3 --pmc passes "
                                         "of this workload), null when no profile of this scale is committed"},
            "call": call,
            "host": {"generate_s": t_gen, "gfa_load_s": gs.load_seconds, "gaf_tokenize_s": tok_s,
                     "gaf_upload_s": up_s, "finish_download_s": t_fin,
                     "gaf_text_bytes": tot("text_bytes"), "gfa_text_bytes": gs.text_bytes,
                     "tokenize_reads_per_s": lines / tok_s if tok_s else None,
                     "cores": os.cpu_count(), "cpu_quota_cores": cpu_quota_cores(), "ram_gb": host_ram_gb(),
                     "peak_rss_gb": __import__("resource").getrusage(__import__("resource").RUSAGE_SELF).ru_maxrss / 1e6},
            # the step in front of the path: GAF text -> records; never part of `value`
            "tokenizer": {"where": {1: "host threads (gaf.cpp)", 2: "GPU kernels (tokenize.hip)"}.get(stats[0].tokenizer),
                          "text_to_hbm_s": up_s if stats[0].tokenizer == 2 else None,
                          "line_index_s": tot("tok_index_seconds"), "measure_and_scans_s": tot("tok_measure_seconds"),
                          "write_s": tot("tok_write_seconds"), "bucket_s": bucket_s,
                          "first_call_of_the_process": first_call,
                          "text_GBps_kernels": (tot("text_bytes") / 1e9 / (tot("tok_index_seconds") + tot("tok_measure_seconds") +
                                                                           tot("tok_write_seconds") + bucket_s)
                                                if stats[0].tokenizer == 2 else None)},
        })
        if keep_for_e2e:
            # The command lines are timed as processes of their own.  This process gives its device memory back first and waits for the
            # driver to scrub it (freed memory is wiped before it is handed out again, and the hipMalloc that gets it waits: 16 GB 1.9 s,
            # tools/hipmalloc_probe.hip) -- otherwise the children's first allocations would pay for what THIS process held.
            for b in batches:
                b.free()
            g.release_device()
            time.sleep(6.0)
            out["e2e"] = e2e_leg(syns, scale, kw, with_cache=not args.no_e2e_cache)
        if not args.no_cpu and world == 1:
            out["cpu_baseline"] = cpu_baseline(kw, args.cpu_sample, lines, syn.stats.junction_links)
        emit(out, args)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
