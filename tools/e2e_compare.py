#!/usr/bin/env python3
"""End-to-end wall time of `augment` + `call` on one mid-size synthetic workload:

  * the real reference scripts (only where /root/reference exists, i.e. the build container),
  * the oracle port (everywhere: calibrates one machine against the other),
  * this engine's command lines (`python -m pantas_amd.augment|call`, process start, GFA parse and
    GFA/CSV writing included) where a GPU is present,

and checks the engine's outputs against the oracle's (GFA bytes, CSV row multiset).

    python tools/e2e_compare.py [--genes 2000] [--lines 300000] [--out result.json]
"""
import argparse
import collections
import json
import os
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def timed(cmd, stdout_path, env=None):
    t0 = time.perf_counter()
    with open(stdout_path, "wb") as f:
        r = subprocess.run(cmd, stdout=f, stderr=subprocess.PIPE, env=env, cwd=ROOT)
    dt = time.perf_counter() - t0
    if r.returncode != 0:
        raise RuntimeError(f"{cmd}: {r.stderr.decode()[-2000:]}")
    return dt


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--genes", type=int, default=2000)
    ap.add_argument("--lines", type=int, default=300000)
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    from pantas_amd.engine import Synth
    from pantas_amd import _lib
    import call_oracle
    tmp = tempfile.mkdtemp(prefix="pantas_e2e_")
    pre = os.path.join(tmp, "w")
    syn = Synth(seed=21, n_genes=a.genes, n_lines=a.lines, base_id=10**7, intron_long_max=200, zipf_a=1.0)
    syn.write(pre)
    flags = ["--rca", "3", "--novel"]
    res = {"workload": {"genes": a.genes, "gaf_lines": a.lines, "segments": syn.stats.nodes, "links": syn.stats.links,
                        "junction_links": syn.stats.junction_links, "call_flags": " ".join(flags)},
           "cpu": os.cpu_count()}
    py = sys.executable
    # ---- oracle port
    t_aug = timed([py, os.path.join(ROOT, "oracle", "augment_oracle.py"), pre + ".gaf", pre + ".gfa"], pre + ".or.gfa")
    t_call = timed([py, os.path.join(ROOT, "oracle", "call_oracle.py"), *flags, pre + ".or.gfa", pre + ".gtf"], pre + ".or.csv")
    res["oracle_port"] = {"augment_s": t_aug, "call_s": t_call, "total_s": t_aug + t_call}
    # ---- the real reference
    if os.path.isdir(os.path.join(a.ref, "scripts")):
        t_aug = timed([py, os.path.join(a.ref, "scripts", "alignments_augmentation_from_gaf.py"), pre + ".gaf", pre + ".gfa"], pre + ".ref.gfa")
        t_call = timed([py, os.path.join(a.ref, "scripts", "call.py"), *flags, pre + ".ref.gfa", pre + ".gtf"], pre + ".ref.csv",
                       env={**os.environ, "PYTHONHASHSEED": "0"})
        # the reference's rows against the oracle's as multisets (after the one normalisation of SURVEY 8-P #2): the oracle fixes one
        # of the reference's hash-seed-dependent behaviours, so rows may differ -- they are counted and shown, not hidden
        norm = call_oracle.normalise_row
        ref_rows = collections.Counter(map(norm, open(pre + ".ref.csv").read().splitlines()[1:]))
        or_rows = collections.Counter(map(norm, open(pre + ".or.csv").read().splitlines()[1:]))
        only_ref, only_or = ref_rows - or_rows, or_rows - ref_rows
        res["reference"] = {"augment_s": t_aug, "call_s": t_call, "total_s": t_aug + t_call,
                            "gfa_equals_oracle": open(pre + ".ref.gfa", "rb").read() == open(pre + ".or.gfa", "rb").read(),
                            "csv_rows": sum(ref_rows.values()), "oracle_csv_rows": sum(or_rows.values()),
                            "csv_equals_oracle_multiset": not only_ref and not only_or,
                            "rows_only_reference_seed0": sum(only_ref.values()), "rows_only_oracle": sum(only_or.values()),
                            "variants": {"only_reference_seed0": sorted(only_ref.elements())[:20], "only_oracle": sorted(only_or.elements())[:20]}}
    # ---- the engine
    if _lib.device_count() > 0:
        runs = []
        for _ in range(3):
            t_aug = timed([py, "-m", "pantas_amd.augment", pre + ".gaf", pre + ".gfa"], pre + ".gpu.gfa")
            t_call = timed([py, "-m", "pantas_amd.call", *flags, pre + ".gpu.gfa", pre + ".gtf"], pre + ".gpu.csv")
            runs.append((t_aug, t_call))
        t_aug, t_call = min(runs, key=lambda x: x[0] + x[1])
        norm = call_oracle.normalise_row
        same_csv = (collections.Counter(map(norm, open(pre + ".gpu.csv").read().splitlines())) ==
                    collections.Counter(map(norm, open(pre + ".or.csv").read().splitlines())))
        res["engine_cli"] = {"augment_s": t_aug, "call_s": t_call, "total_s": t_aug + t_call, "runs": runs,
                             "gfa_equals_oracle": open(pre + ".gpu.gfa", "rb").read() == open(pre + ".or.gfa", "rb").read(),
                             "csv_equals_oracle_multiset": same_csv}
        res["speedup_vs_oracle_port_same_machine"] = res["oracle_port"]["total_s"] / (t_aug + t_call)
    print(json.dumps(res, indent=1))
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
