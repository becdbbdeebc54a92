#!/usr/bin/env python3
"""Build tests/golden/ by running the REAL reference scripts in this container.

Usage:  python tools/make_golden.py [--ref /root/reference] [--seeds 16]

The reference is stdlib-only Python and runs here; it never travels to the GPU box
(it is read from ``--ref`` at generation time only).  For every case this script
writes the inputs it synthesised itself (GFA / GAF / GTF -- data, not reference
code) and what the reference printed for them:

  <case>/in.gfa in.gaf in.gtf        inputs
  <case>/expected.aug.gfa             stdout of alignments_augmentation_from_gaf.py (bytes)
  <case>/expected.json                {"rejected": N, "calls": [{"flags": [...],
                                        "stable": [rows], "variants": [rows]} ...]}
  crashes.json                        inputs on which the reference raises, with the
                                      exception class it raised

``stable`` = rows (as a sorted multiset, after the one normalisation of
``oracle/call_oracle.normalise_row``) that every PYTHONHASHSEED produced; ``variants`` =
rows that only some seeds produced (the reference iterates Python sets of strings).
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

from pantas_amd import synth  # noqa: E402
import call_oracle  # noqa: E402


# --------------------------------------------------------------------------------------
# hand-made cases
# --------------------------------------------------------------------------------------
def seq(n, ch="A"):
    return ch * n


def gaf_line(name, nodes, lens, start, span, cs, rev=False, mapq=60, dv="0", extra="", sep="\t",
             tags=None):
    """One GAF record.  ``nodes`` in READ order; start/span are on the read-order path."""
    plen = sum(lens[n] for n in nodes)
    p = "".join(("<" if rev else ">") + str(n) for n in nodes)
    end = start + span
    cols = [name, "100", "0", "100", "+", p, str(plen), str(start), str(end), str(span), str(span), str(mapq)]
    if tags is None:
        tags = [f"AS:i:{span}"]
        if cs is not None:
            tags.append(f"cs:Z:{cs}")
        if dv is not None:
            tags.append(f"dv:f:{dv}")
    return sep.join(cols + tags) + extra


MICRO_LENS = {1001: 10, 1002: 20, 1003: 30, 1004: 25, 1005: 30, 1006: 40}


def micro_case():
    """SURVEY.md Appendix A."""
    L = MICRO_LENS
    gfa = ["H\tVN:Z:1.1"]
    ex = {1001: "T1_R1.1,T2_R1.1", 1002: "T1_R1.1,T2_R1.1", 1004: "T1_R1.2", 1006: "T1_R1.3,T2_R1.2"}
    for n in sorted(L):
        gfa.append(f"S\t{n}\t{seq(L[n])}" + (f"\tEX:Z:{ex[n]}" if n in ex else ""))
    jn = {(1002, 1004): "T1_R1.1.2", (1002, 1006): "T2_R1.1.2", (1004, 1006): "T1_R1.2.3"}
    for u, v in [(1001, 1002), (1002, 1003), (1002, 1004), (1002, 1006), (1003, 1004), (1004, 1005),
                 (1004, 1006), (1005, 1006)]:
        gfa.append(f"L\t{u}\t+\t{v}\t+\t*" + (f"\tJN:Z:{jn[(u, v)]}" if (u, v) in jn else ""))
    gfa.append("P\tchr4\t" + ",".join(f"{n}+" for n in sorted(L)) + "\t*")
    gtf = ['chr4\tX\tgene\t1\t155\t.\t+\t.\tgene_id "G1";',
           'chr4\tX\tmRNA\t1\t155\t.\t+\t.\tgene_id "G1"; transcript_id "T1";',
           'chr4\tX\tmRNA\t1\t155\t.\t+\t.\tgene_id "G1"; transcript_id "T2";']
    gaf = [
        gaf_line("r1", [1001, 1002, 1004, 1006], L, 5, 60, ":60"),
        gaf_line("r2", [1002, 1006], L, 3, 50, ":20*ag:29"),
        gaf_line("r3", [1006, 1004, 1002], L, 10, 60, ":60", rev=True),
        gaf_line("r4", [1002, 1003, 1004], L, 0, 60, ":30-acg:27"),
        gaf_line("r5", [1001, 1002, 1004], L, 0, 40, ":40", mapq=5),
        gaf_line("r6", [1001, 1004, 1006], L, 0, 70, ":70"),
        "r7\t100\t0\t0\t+\t*\t0\t0\t0\t0\t0\t60",
    ]
    calls = [["--rca", "0", "--novel"], [], ["--rca", "2"], ["--rca", "0", "--novel", "--no-annotated"]]
    return "micro", gfa, gaf, gtf, calls


QL = {2001: 10, 2002: 20, 2003: 1, 2004: 25, 2005: 30, 2006: 40, 2007: 5, 2008: 12}


def quirks_case():
    L = QL
    ex = {2001: "Q1T1_R1.1,Q1T2_R1.1", 2002: "Q1T1_R1.1,Q1T2_R1.1", 2004: "Q1T1_R1.2",
          2006: "Q1T1_R1.3,Q1T2_R1.2", 2008: "Q1T1_R1.4,Q1T2_R1.3"}
    jn = {(2002, 2004): "Q1T1_R1.1.2", (2004, 2006): "Q1T1_R1.2.3", (2002, 2006): "Q1T2_R1.1.2",
          (2006, 2008): "Q1T1_R1.3.4,Q1T2_R1.2.3"}
    gfa = ["H\tVN:Z:1.1", "# a comment line is echoed"]
    for n in sorted(L):
        s = f"S\t{n}\t{seq(L[n], 'C')}"
        if n == 2005:
            s += "\tLN:i:30"                      # pre-existing tag, more than 3 tokens
        if n in ex:
            s += f"\tEX:Z:{ex[n]}"
        gfa.append(s)
    links = [(2001, 2002), (2002, 2003), (2002, 2004), (2003, 2004), (2004, 2005), (2005, 2006),
             (2004, 2006), (2002, 2006), (2006, 2007), (2007, 2008), (2006, 2008)]
    for u, v in links:
        gfa.append(f"L\t{u}\t+\t{v}\t+\t*" + (f"\tJN:Z:{jn[(u, v)]}" if (u, v) in jn else ""))
    gfa.append("L\t2004\t+\t2005\t+\t*")           # duplicate link line: gets RC:i:0
    gfa.append("L")                                # bare L line: dropped by augment
    gfa.append("")                                 # blank line: echoed as blank
    gfa.append("  P\tchrQ\t" + ",".join(f"{n}+" for n in sorted(L)) + "\t*  ")  # stripped
    gtf = ["#!quirks", 'chrQ\tX\tgene\t1\t143\t.\t-\t.\tgene_id "Q1";',
           'chrQ\tX\ttranscript\t1\t143\t.\t-\t.\tgene_id "Q1"; transcript_id "Q1T1";',
           'chrQ\tX\tncRNA\t1\t143\t.\t-\t.\ttranscript_id "Q1T2"; gene_id "Q1";',
           'chrQ\tX\texon\t1\t30\t.\t-\t.\tgene_id "Q1"; transcript_id "Q1T1";']
    g = []
    a = g.append
    a(gaf_line("q01_leadstar", [2001, 2002], L, 0, 15, "*ac:14"))
    a(gaf_line("q02_staronly", [2002, 2003, 2004], L, 15, 16, ":5*ag:10"))
    a(gaf_line("q03_lone_del", [2006, 2007, 2008], L, 30, 21, ":10-acgta:6"))
    a(gaf_line("q04_lone_ins", [2006, 2007, 2008], L, 30, 21, ":10+acgta:6"))
    a(gaf_line("q05_clip_front", [2005, 2006], L, 2, 53, "+acg:50"))
    a(gaf_line("q06_clip_back", [2005, 2006], L, 5, 50, ":50+acg"))
    a(gaf_line("q07_noclip3", [2005, 2006], L, 5, 52, "+ac:20:30"))
    a(gaf_line("q08_dv_exp", [2001, 2002], L, 0, 15, ":15", dv="2e-05"))
    a(gaf_line("q09_nocs", [2003], L, 0, 1, None))
    a(gaf_line("q10_duprun", [2001, 2001, 2002], L, 0, 15, ":15").replace("\t20\t", "\t40\t", 0))
    a(gaf_line("q11_eq", [2001, 2002], L, 0, 15, "=CCCCCCCCCC:5"))
    a(gaf_line("q12_del_at_start", [2001, 2002], L, 0, 20, ":10-ac:8"))
    a(gaf_line("q12b_del_at_end", [2001, 2002], L, 0, 18, ":7-acg:8"))
    a(gaf_line("q13_del_at_start_rev", [2002, 2001], L, 0, 28, ":20-ac:6", rev=True))
    a(gaf_line("q13b_del_at_end_rev", [2002, 2001], L, 0, 28, ":17-acg:8", rev=True))
    a(gaf_line("q14_del_across", [2001, 2002], L, 0, 20, ":8-acgt:8"))
    a(gaf_line("q14b_del_across_rev", [2004, 2002], L, 10, 25, ":12-acgtac:7", rev=True))
    a(gaf_line("q15_zero_budget", [2003, 2004], L, 1, 10, ":10"))
    a(gaf_line("q15b_neg_budget", [2003, 2004], L, 3, 10, ":10"))
    a(gaf_line("q16_mapq19", [2001, 2002], L, 0, 15, ":15", mapq=19))
    a(gaf_line("q16b_mapq20", [2001, 2002], L, 0, 15, ":15", mapq=20))
    a("q17_unmapped_low\t100\t0\t0\t+\t*\t0\t0\t0\t0\t0\t3")
    a("q18_unmapped_ok\t100\t0\t0\t+\t*\t0\t0\t0\t0\t0\t60")
    a(gaf_line("q19_multistar", [2004, 2005], L, 20, 9, ":3*ac*gt:4"))
    a(gaf_line("q20_revisit", [2001, 2002, 2001], L, 0, 35, ":35"))
    a(gaf_line("q21_same_keys", [2001, 2002], L, 0, 20, ":10-ac:8"))
    a(gaf_line("q22_order_a", [2004, 2005], L, 20, 12, ":5:7"))
    a(gaf_line("q22_order_b", [2004, 2005], L, 20, 12, ":5-ac:5"))
    a(gaf_line("q22_order_c", [2005, 2006], L, 25, 12, ":5-ac:5"))
    a(gaf_line("q22_order_d", [2005, 2006], L, 25, 12, ":5:7"))
    a(gaf_line("q24_embedded", [2001, 2002], L, 0, 15, None, tags=["xcs:Z::15", "dv:f:0.05"]))
    a(gaf_line("q25_two_cs", [2001, 2002], L, 0, 15, None, tags=["cs:Z::15", "cs:Z::3", "zzdv:f:0.1", "dv:f:0.9"]))
    a(gaf_line("q26_spaces", [2001, 2002], L, 0, 15, ":15", sep=" "))
    a(gaf_line("q27_dv_high", [2001, 2002], L, 0, 15, ":15", dv="0.11"))
    a(gaf_line("q27b_dv_edge", [2001, 2002], L, 0, 15, ":15", dv="0.1"))
    a(gaf_line("q28_ins_mid", [2004, 2005, 2006], L, 20, 45, ":5+ac:30:10"))
    a(gaf_line("q29_rev_long", [2008, 2006, 2004, 2002], L, 2, 80, ":30*ct:20-ac:27", rev=True))
    a(gaf_line("q30_single", [2006], L, 3, 30, ":30"))
    a(gaf_line("q31_single_rev", [2006], L, 3, 30, ":12-a:17", rev=True))
    a(gaf_line("q32_zero_len_op", [2001, 2002], L, 0, 15, "::15"))
    a(gaf_line("q33_trailing_ws", [2001, 2002], L, 0, 15, ":15", extra="  \t "))
    a(gaf_line("q34_ops_left_over", [2001, 2002], L, 0, 15, ":15:200*ac"))
    a(gaf_line("q35_novel_rev", [2008, 2005, 2002], L, 0, 50, ":50", rev=True))
    a(gaf_line("q36_novel_rev_again", [2008, 2005], L, 0, 30, ":30", rev=True))
    a(gaf_line("q37_del_whole_mid", [2004, 2005, 2006], L, 20, 45, ":5-" + "a" * 30 + ":10"))
    a(gaf_line("q38_star_then_del", [2001, 2002], L, 0, 20, ":10*ac-ac:7"))
    calls = [[], ["--rca", "0", "--novel", "--minintronsize", "1"], ["--events", "SS", "--rca", "1"],
             ["--events", "ES", "SS", "--novel", "--isw", "1", "--minintronsize", "5"]]
    return "quirks", gfa, g, gtf, calls


def mixedwidth_case():
    """ids 97..104: '100' < '98' as strings -- pins the reference's string comparisons."""
    L = {97: 32, 98: 32, 99: 32, 100: 32, 101: 32, 102: 32, 103: 32, 104: 20}
    ex = {97: "MT1_R1.1,MT2_R1.1", 98: "MT1_R1.1,MT2_R1.1", 99: "MT2_R1.1", 100: "MT2_R1.1",
          101: "MT1_R1.2,MT2_R1.1", 102: "MT1_R1.2,MT2_R1.1", 104: "MT1_R1.3,MT2_R1.2"}
    jn = {(98, 101): "MT1_R1.1.2", (102, 104): "MT1_R1.2.3,MT2_R1.1.2"}
    gfa = ["H\tVN:Z:1.1"]
    for n in sorted(L):
        gfa.append(f"S\t{n}\t{seq(L[n], 'G')}" + (f"\tEX:Z:{ex[n]}" if n in ex else ""))
    for u, v in [(97, 98), (98, 99), (98, 101), (99, 100), (100, 101), (101, 102), (102, 103), (102, 104),
                 (103, 104)]:
        gfa.append(f"L\t{u}\t+\t{v}\t+\t*" + (f"\tJN:Z:{jn[(u, v)]}" if (u, v) in jn else ""))
    gtf = ['chrM\tX\tgene\t1\t250\t.\t+\t.\tgene_id "MG";',
           'chrM\tX\tmRNA\t1\t250\t.\t+\t.\tgene_id "MG"; transcript_id "MT1";',
           'chrM\tX\tmRNA\t1\t250\t.\t+\t.\tgene_id "MG"; transcript_id "MT2";']
    gaf = [gaf_line("m1", [97, 98, 101, 102], L, 2, 100, ":100"),
           gaf_line("m2", [98, 99, 100, 101], L, 10, 100, ":100"),
           gaf_line("m3", [101, 102, 104], L, 0, 80, ":80"),
           gaf_line("m4", [102, 103, 104], L, 5, 70, ":70"),
           gaf_line("m5", [104, 102, 101, 98], L, 0, 100, ":100", rev=True)]
    calls = [[], ["--rca", "0"]]
    return "mixedwidth", gfa, gaf, gtf, calls


def crash_cases():
    """Inputs on which the reference dies.  Each: (name, stage, gfa, gaf, gtf, flags)."""
    _, gfa, _, gtf, _ = micro_case()
    L = MICRO_LENS
    ok = gaf_line("ok", [1001, 1002], L, 0, 15, ":15")
    out = []
    out.append(("aug_missing_dv", "augment", gfa, [ok, gaf_line("x", [1001, 1002], L, 0, 15, ":15", dv=None)], gtf, []))
    out.append(("aug_ops_exhausted", "augment", gfa, [ok, gaf_line("x", [1001, 1002, 1004], L, 0, 40, ":12")], gtf, []))
    out.append(("aug_unknown_node", "augment", gfa, [ok, gaf_line("x", [1001, 9999], {**L, 9999: 5}, 0, 12, ":12")], gtf, []))
    out.append(("aug_tilde_first", "augment", gfa, [gaf_line("x", [1001, 1002], L, 0, 15, "~ac5gt:10")], gtf, []))
    out.append(("aug_short_line", "augment", gfa, [ok, "x\t100\t0\t100\t+\t>1001"], gtf, []))
    out.append(("aug_bad_mapq", "augment", gfa, [ok.replace("\t60\t", "\tsixty\t")], gtf, []))
    out.append(("aug_no_orient", "augment", gfa, [ok.replace(">1001>1002", "1001")], gtf, []))
    out.append(("aug_mixed_orient", "augment", gfa, [ok.replace(">1001>1002", "<1002>1001")], gtf, []))
    out.append(("aug_empty_cs", "augment", gfa, [gaf_line("x", [1001, 1002], L, 0, 15, None, tags=["cs:Z:", "dv:f:0"])], gtf, []))
    # call-stage crashes run on an augmented micro graph made below
    out.append(("call_unaugmented", "call", gfa, None, gtf, []))
    bad_tag = [l + "\tXX:Z:a:b" if l.startswith("S\t1003") else l for l in gfa]
    out.append(("call_tag_extra_colon", "call_aug", bad_tag, [ok], gtf, []))
    out.append(("call_tx_not_in_gtf", "call_aug", gfa, [ok], gtf[:2], []))
    out.append(("call_bad_gtf_id", "call_aug", gfa, [ok], gtf[:2] + ['chr4\tX\tmRNA\t1\t5\t.\t+\t.\tgene_id "G1"; transcript_id "T-2";'], []))
    out.append(("call_short_p_line", "call_aug", gfa + ["P\tx\t1001+"], [ok], gtf, []))
    return out


# --------------------------------------------------------------------------------------
def run_ref(ref, script, args, seed=None, stdin=None):
    env = dict(os.environ)
    if seed is not None:
        env["PYTHONHASHSEED"] = str(seed)
    return subprocess.run([sys.executable, os.path.join(ref, "scripts", script), *args],
                          capture_output=True, text=True, env=env)


def write_lines(path, lines):
    with open(path, "w") as f:
        for l in lines:
            f.write(l + "\n")


def build_case(ref, outdir, name, calls, nseeds):
    d = os.path.join(outdir, name)
    r = run_ref(ref, "alignments_augmentation_from_gaf.py", [f"{d}/in.gaf", f"{d}/in.gfa"])
    assert r.returncode == 0, (name, r.stderr)
    with open(f"{d}/expected.aug.gfa", "w") as f:
        f.write(r.stdout)
    rej = [l for l in r.stderr.splitlines() if l.startswith("Rejected alignments:")]
    meta = {"rejected": int(rej[0].split(":")[1]), "stderr": r.stderr.splitlines(), "calls": []}
    for flags in calls:
        runs = []
        for s in range(nseeds):
            c = run_ref(ref, "call.py", [*flags, f"{d}/expected.aug.gfa", f"{d}/in.gtf"], seed=s)
            assert c.returncode == 0, (name, flags, c.stderr)
            rows = c.stdout.splitlines()
            assert rows[0] == call_oracle.HEADER
            runs.append(collections.Counter(call_oracle.normalise_row(x) for x in rows[1:]))
        stable = runs[0]
        for x in runs[1:]:
            stable = stable & x
        variants = set()
        for x in runs:
            variants |= set((x - stable).keys())
        meta["calls"].append({"flags": flags, "n_rows": sum(runs[0].values()),
                              "stable": sorted(stable.elements()), "variants": sorted(variants)})
        print(f"  {name} {' '.join(flags) or '(default)'}: {sum(runs[0].values())} rows, "
              f"{len(variants)} seed-dependent alternatives")
    with open(f"{d}/expected.json", "w") as f:
        json.dump(meta, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--seeds", type=int, default=16)
    ap.add_argument("--out", default=os.path.join(ROOT, "tests", "golden"))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)

    for case in (micro_case(), quirks_case(), mixedwidth_case()):
        name, gfa, gaf, gtf, calls = case
        d = os.path.join(a.out, name)
        os.makedirs(d, exist_ok=True)
        write_lines(f"{d}/in.gfa", gfa)
        write_lines(f"{d}/in.gaf", gaf)
        write_lines(f"{d}/in.gtf", gtf)
        build_case(a.ref, a.out, name, calls, a.seeds)

    synth_cases = [
        ("synth_a", dict(seed=1, n_genes=14, n_lines=1500),
         [[], ["--rca", "3", "--novel", "--minintronsize", "25"],
          ["--rca", "0", "--novel", "--no-annotated", "--isw", "2"],
          ["--events", "ES", "IR", "--rca", "1", "--novel", "--minintronsize", "1"]]),
        ("synth_b", dict(seed=2, n_genes=10, n_lines=1200, p_indel=0.25, p_sub=0.4, bubble_rate=0.12,
                         p_novel_skip=0.08, p_novel_ce=0.05, p_intron=0.12, p_novel_ss=0.08),
         [["--rca", "2", "--novel", "--minintronsize", "40"], ["--rca", "5"],
          ["--events", "SS", "--novel", "--isw", "8"]]),
        ("synth_c", dict(seed=3, n_genes=8, n_lines=800, bubble_rate=0.0, base_id=10_000_000),
         [["--rca", "1", "--novel", "--minintronsize", "30"]]),
    ]
    for name, kw, calls in synth_cases:
        d = os.path.join(a.out, name)
        os.makedirs(d, exist_ok=True)
        synth.write_all(f"{d}/in", **kw)
        build_case(a.ref, a.out, name, calls, a.seeds)

    crashes = []
    cd = os.path.join(a.out, "crashes")
    os.makedirs(cd, exist_ok=True)
    for name, stage, gfa, gaf, gtf, flags in crash_cases():
        write_lines(f"{cd}/{name}.gfa", gfa)
        write_lines(f"{cd}/{name}.gtf", gtf)
        if gaf is not None:
            write_lines(f"{cd}/{name}.gaf", gaf)
        if stage == "augment":
            r = run_ref(a.ref, "alignments_augmentation_from_gaf.py", [f"{cd}/{name}.gaf", f"{cd}/{name}.gfa"])
            call_input = None
        elif stage == "call":
            r = run_ref(a.ref, "call.py", [*flags, f"{cd}/{name}.gfa", f"{cd}/{name}.gtf"], seed=0)
            call_input = f"{name}.gfa"
        else:
            r0 = run_ref(a.ref, "alignments_augmentation_from_gaf.py", [f"{cd}/{name}.gaf", f"{cd}/{name}.gfa"])
            assert r0.returncode == 0, (name, r0.stderr)
            with open(f"{cd}/{name}.aug.gfa", "w") as f:
                f.write(r0.stdout)
            r = run_ref(a.ref, "call.py", [*flags, f"{cd}/{name}.aug.gfa", f"{cd}/{name}.gtf"], seed=0)
            call_input = f"{name}.aug.gfa"
        assert r.returncode != 0, f"{name}: reference did not crash"
        last = r.stderr.strip().splitlines()[-1]
        exc = last.split(":")[0].strip()
        crashes.append({"name": name, "stage": "augment" if stage == "augment" else "call",
                        "call_input": call_input, "flags": flags, "exception": exc,
                        "stdout_bytes": len(r.stdout)})
        print(f"  crash {name}: {last[:100]}")
    with open(os.path.join(a.out, "crashes.json"), "w") as f:
        json.dump(crashes, f, indent=1)


if __name__ == "__main__":
    main()
