#!/usr/bin/env python3
"""Round-2 goldens: inputs at the edge of what the engine accepts, with what the REAL reference prints for them.

    python tools/make_golden_r2.py [--ref /root/reference]

Run in the build container only (the reference never travels).  Writes tests/golden/edge/<case>.{gfa,gaf,gtf},
the reference's stdout as <case>.expected (augment: the GFA bytes; call: the CSV rows under PYTHONHASHSEED=0 ..) and
tests/golden/edge/cases.json = [{name, stage, flags, rc, exception, engine}], where `engine` says what this
repo's engine does with the input: "same" (must reproduce the reference, output or exception class) or
"refuses" (a documented deviation, DESIGN.md section 4: the engine fails loudly where the reference succeeds) --
the tests hold the oracle to the reference on every case and the engine on the "same" ones, and keep the
"refuses" ones as expected failures with the reference's bytes beside them.
"""
from __future__ import annotations

import argparse
import collections
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tools"))

import call_oracle  # noqa: E402
import make_golden as mg  # noqa: E402


def cases():
    _, gfa, gaf, gtf, _ = mg.micro_case()
    L = mg.MICRO_LENS
    out = []
    # ---- '~' ops (alignments_augmentation_from_gaf.py:228-233 never assigns a length for them) -------------
    tilde = [
        mg.gaf_line("t1_mid", [1001, 1002, 1004], L, 0, 50, ":10~ac5gt:25"),              # laid out with the ':' length before it
        mg.gaf_line("t2_split", [1002, 1003, 1004], L, 3, 60, ":4~ac9gt:30*ag:20"),       # stale 4 < room: whole; then ':'
        mg.gaf_line("t3_rev", [1006, 1004, 1002], L, 10, 60, ":30~ct3ag:20", rev=True),
        mg.gaf_line("t4_over_nodes", [1001, 1002, 1003], L, 2, 50, ":40~aa1cc:3"),        # stale 40 > room: the '~' piece is cut at the node end
        mg.gaf_line("t5_after_star", [1002, 1004], L, 5, 30, ":3*ag~gt7ag:20"),
        mg.gaf_line("t6_after_del", [1002, 1003, 1004], L, 0, 60, ":25-acg~gt7ag:30"),
        mg.gaf_line("t7_twice", [1004, 1005, 1006], L, 1, 70, ":6~gt1ag~gt2ag:50"),
        mg.gaf_line("t8_last", [1001, 1002], L, 0, 25, ":9~gt1ag"),
    ]
    out.append(dict(name="tilde_mid", stage="augment", gfa=gfa, gaf=tilde, gtf=gtf, flags=[], engine="same"))
    leak = [mg.gaf_line("ok", [1001, 1002], L, 0, 15, ":15"),
            mg.gaf_line("x_first_op", [1001, 1002], L, 0, 15, "~ac5gt:10"),                # length leaks in from the line before
            mg.gaf_line("rejected", [1001, 1002], L, 0, 15, ":7", mapq=3),
            mg.gaf_line("y_first_op", [1002, 1004], L, 4, 20, "~ac5gt:30")]
    out.append(dict(name="tilde_leak", stage="augment", gfa=gfa, gaf=leak, gtf=gtf, flags=[], engine="same"))
    # ---- S line behind leading whitespace --------------------------------------------------------------------
    ws = [("  " + l if l.startswith("S\t1003") else l) for l in gfa]
    out.append(dict(name="ws_s_augment", stage="augment", gfa=ws, gaf=gaf, gtf=gtf, flags=[], engine="same"))
    out.append(dict(name="ws_s_call", stage="call_aug_ws", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0", "--novel"], engine="same"))
    # ---- duplicate S line in call ------------------------------------------------------------------------------
    out.append(dict(name="dup_s_call", stage="call_aug_dup_s", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0", "--novel"], engine="refuses"))
    out.append(dict(name="dup_s_call_annotated", stage="call_aug_dup_s", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0"], engine="same"))
    # ---- duplicate L lines that disagree on JN ------------------------------------------------------------------
    out.append(dict(name="dup_l_jn_last", stage="call_aug_dup_l_last", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0"], engine="refuses"))
    out.append(dict(name="dup_l_jn_first", stage="call_aug_dup_l_first", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0"], engine="same"))
    # ---- a lone carriage return: Python's universal newlines make it a line break --------------------------------
    cr_mid = [gaf[0], gaf[1].replace("\tcs:Z", "\r\tcs:Z"), gaf[2]]
    out.append(dict(name="cr_mid_line", stage="augment", gfa=gfa, gaf=cr_mid, gtf=gtf, flags=[], engine="same"))
    cr_sep = [gaf[0] + "\r" + gaf[1], gaf[2]]
    out.append(dict(name="cr_between_records", stage="augment", gfa=gfa, gaf=cr_sep, gtf=gtf, flags=[], engine="same"))
    # the same in the other two inputs: two S lines of the GFA / two rows of the GTF with a lone '\r' between them
    k3 = [k for k, l in enumerate(gfa) if l.startswith("S\t1003")][0]
    gfa_cr = gfa[:k3] + [gfa[k3] + "\r" + gfa[k3 + 1]] + gfa[k3 + 2:]
    out.append(dict(name="cr_in_gfa", stage="augment", gfa=gfa_cr, gaf=gaf, gtf=gtf, flags=[], engine="same"))
    out.append(dict(name="cr_in_gfa_call", stage="call_aug_cr", gfa=gfa, gaf=gaf, gtf=gtf, flags=["--rca", "0", "--novel"], engine="same"))
    gtf_cr = [gtf[0] + "\r" + gtf[1]] + gtf[2:]
    out.append(dict(name="cr_in_gtf", stage="call_aug_plain", gfa=gfa, gaf=gaf, gtf=gtf_cr, flags=["--rca", "0", "--novel"], engine="same"))
    crlf = [l + "\r" for l in gaf]
    out.append(dict(name="crlf_gaf", stage="augment", gfa=[l + "\r" for l in gfa], gaf=crlf, gtf=gtf, flags=[], engine="same"))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    ap.add_argument("--seeds", type=int, default=8)
    a = ap.parse_args()
    d = os.path.join(ROOT, "tests", "golden", "edge")
    os.makedirs(d, exist_ok=True)
    meta = []
    for c in cases():
        n = c["name"]
        mg.write_lines(f"{d}/{n}.gfa", c["gfa"])
        mg.write_lines(f"{d}/{n}.gaf", c["gaf"])
        mg.write_lines(f"{d}/{n}.gtf", c["gtf"])
        rec = dict(name=n, stage="augment" if c["stage"] == "augment" else "call", flags=c["flags"], engine=c["engine"])
        if c["stage"] == "augment":
            r = mg.run_ref(a.ref, "alignments_augmentation_from_gaf.py", [f"{d}/{n}.gaf", f"{d}/{n}.gfa"])
            rec["stdout_bytes_before_crash"] = len(r.stdout) if r.returncode else None
            if r.returncode == 0:
                open(f"{d}/{n}.expected", "w").write(r.stdout)
                rec["rejected"] = int([l for l in r.stderr.splitlines() if l.startswith("Rejected")][0].split(":")[1])
        else:
            # augment the plain micro case with the reference, then damage the AUGMENTED graph the way the case asks
            r0 = mg.run_ref(a.ref, "alignments_augmentation_from_gaf.py", [f"{d}/{n}.gaf", f"{d}/{n}.gfa"])
            assert r0.returncode == 0, r0.stderr
            lines = r0.stdout.splitlines()
            kind = c["stage"]
            if kind == "call_aug_ws":
                lines = [("  " + l if l.startswith("S\t1003") else l) for l in lines]
            elif kind == "call_aug_dup_s":
                i = [k for k, l in enumerate(lines) if l.startswith("L\t1002\t+\t1004")][0]
                s1002 = [l for l in lines if l.startswith("S\t1002")][0]
                lines.insert(i + 1, s1002)            # a second S line for 1002 after some of its links
            elif kind == "call_aug_cr":
                i = [k for k, l in enumerate(lines) if l.startswith("S\t1003")][0]
                lines[i:i + 2] = [lines[i] + "\r" + lines[i + 1]]      # a lone '\r' between two S lines of the augmented graph
            elif kind in ("call_aug_dup_l_last", "call_aug_dup_l_first"):
                i = [k for k, l in enumerate(lines) if l.startswith("L\t1002\t+\t1004")][0]
                plain = "L\t1002\t+\t1004\t+\t*\tRC:i:2"
                if kind == "call_aug_dup_l_last":
                    lines.insert(i + 1, plain)        # the later line has no JN tag
                else:
                    lines.insert(i, plain)            # the earlier line has no JN tag
            mg.write_lines(f"{d}/{n}.aug.gfa", lines)
            runs = []
            r = None
            for s in range(a.seeds):
                r = mg.run_ref(a.ref, "call.py", [*c["flags"], f"{d}/{n}.aug.gfa", f"{d}/{n}.gtf"], seed=s)
                if r.returncode:
                    break
                rows = r.stdout.splitlines()
                runs.append(collections.Counter(call_oracle.normalise_row(x) for x in rows[1:]))
            if r.returncode == 0:
                stable = runs[0]
                for x in runs[1:]:
                    stable = stable & x
                variants = set()
                for x in runs:
                    variants |= set((x - stable).keys())
                json.dump({"stable": sorted(stable.elements()), "variants": sorted(variants)}, open(f"{d}/{n}.expected", "w"), indent=1)
        rec["rc"] = r.returncode
        rec["exception"] = r.stderr.strip().splitlines()[-1].split(":")[0].strip() if r.returncode else None
        meta.append(rec)
        print(f"{n}: rc {r.returncode} {rec['exception'] or ''}")
    json.dump(meta, open(f"{d}/cases.json", "w"), indent=1)
    # ---- --debug: the "Checking ..." lines (scripts/call.py:254 ... :1002) as a sorted multiset (the order in which the
    # reference visits junctions is that of a Python set), per golden case and flag set
    dbg = {}
    gold = os.path.join(ROOT, "tests", "golden")
    for case, flag_sets in (("micro", [["--rca", "0", "--novel"], []]),
                            ("synth_a", [["--rca", "3", "--novel", "--minintronsize", "25"], ["--events", "ES", "IR", "--rca", "1", "--novel", "--minintronsize", "1"]]),
                            ("synth_b", [["--events", "SS", "--novel", "--isw", "8"], ["--rca", "5"]])):
        for flags in flag_sets:
            r = mg.run_ref(a.ref, "call.py", ["--debug", *flags, f"{gold}/{case}/expected.aug.gfa", f"{gold}/{case}/in.gtf"], seed=0)
            assert r.returncode == 0, r.stderr
            # with --debug the reference prints the lines with print(..., file=sys.stderr): all on stderr, between the
            # two timestamped "Parsing" lines and the rows on stdout
            lines = [l for l in r.stderr.splitlines() if l.startswith(("Checking", "Skipping"))]
            dbg.setdefault(case, []).append({"flags": flags, "stderr_sorted": sorted(lines)})
            print(f"debug {case} {' '.join(flags)}: {len(lines)} lines")
    json.dump(dbg, open(f"{d}/debug_lines.json", "w"), indent=0)


if __name__ == "__main__":
    main()
