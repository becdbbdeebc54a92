"""`pantas quant`: PSI / dPSI of AS events over the replicates of two conditions.

Mirror of the reference's ``scripts/quantify.py`` (same command line, same CSV, same row order, same float
printing) with the replicate merge done as a hash join: the reference finds the partner of an event by a linear
``eq_event`` scan (``quantify.py:263,281,312,344``), which is quadratic in the number of events; here the identity
that ``eq_event`` (``:218-231``) compares -- type, chrom, gene and the node tuples of the event / canonical
junctions (only the event junction for IR) -- is a dict key.  The numbers are computed with the same Python
operations (``statistics.mean``, true and floor division of floats, ``max(0, psi)``), so they print alike.

    python -m pantas_amd.quantify [-w W] [--both] --c1 C1.R1 [C1.R2 ...] --c2 C2.R1 [C2.R2 ...]
"""
from __future__ import annotations

import argparse
import sys
from statistics import mean

ETYPES = ("ES", "CE", "IR", "A3", "A5")
COLUMNS = 14   # call.py's CSV (scripts/call.py:121-129)
HEADER = ("etype,annotation_type,chrom,gene,strand,junction1_name,junction2_name,junction3_name,"
          "junction1_nodes,junction2_nodes,junction3_nodes,W1,W2,psi_c1,psi_c2,dpsi")


def node_ids(text: str) -> tuple:
    """quantify.py:9-12 -- ids of a '>'-joined path; '..' (elided IR interior) and '?' are skipped"""
    return tuple(int(x) for x in text.split(">") if x != ".." and x != "?")


def coverage(text: str) -> float:
    return float(text) if text != "." else -1


def _int_mean(vals: list) -> int:
    """int(statistics.mean(vals)) -- coverages are whole numbers held in floats: their sum is exact as an int, and the true division of
    two ints is correctly rounded like float(Fraction(sum, n)), so the Fraction arithmetic of statistics.mean (more than half of the
    run time of this script) is only needed for anything else"""
    try:
        ints = [int(v) for v in vals]
        if all(i == v for i, v in zip(ints, vals)):
            return int(sum(ints) / len(ints))
    except (OverflowError, ValueError):
        pass
    return int(mean(vals))


class Event:
    """One row of a call CSV with its junctions sorted into 'event' and 'canonical' (quantify.py:65-160)."""
    __slots__ = ("etype", "annotation_type", "chrom", "gene", "strand", "event_cov", "canonic_cov", "key",
                 "event_names", "canonic_names", "event_nodes", "canonic_nodes", "replicates")

    def __init__(self, fields: list, replicate: int):
        if len(fields) != COLUMNS:
            raise TypeError(f"an event row has {COLUMNS} columns, this one has {len(fields)}")
        (self.etype, self.annotation_type, self.chrom, self.gene, self.strand, n1, n2, n3,
         p1, c1, p2, c2, p3, c3) = fields
        c1, c2, c3 = coverage(c1), coverage(c2), coverage(c3)
        et = self.etype
        if et == "ES" or et == "CE":
            pair_cov = (c2 + c3) // 2
            pair_ids = (node_ids(p2), node_ids(p3))
            pair_nodes = sorted([p2, p3])
            if et == "ES":          # skipping junction = event, the two inclusion junctions = canonical
                single_ids = node_ids(p1)
                self.event_cov, self.canonic_cov = c1, pair_cov
                event_ids, canonic_ids = single_ids, pair_ids
                self.event_nodes, self.canonic_nodes = [p1], pair_nodes
                self.event_names, self.canonic_names = [n1], [n2, n3]
            else:                   # cassette exon: the other way round (ids parsed in the reference's order)
                self.event_cov, self.canonic_cov = pair_cov, c1
                event_ids, canonic_ids = pair_ids, node_ids(p1)
                self.event_nodes, self.canonic_nodes = pair_nodes, [p1]
                self.event_names, self.canonic_names = [n2, n3], [n1]
        elif et in ("A5", "A3", "IR"):
            if et == "A5":
                first_is_event = self.strand == "+"
            elif et == "A3":
                first_is_event = self.strand != "+"
            else:
                first_is_event = not (self.annotation_type == "novel" and n2 == "?")
            ev, ca = ((n1, p1, c1), (n2, p2, c2)) if first_is_event else ((n2, p2, c2), (n1, p1, c1))
            self.event_cov = ev[2]
            event_ids = node_ids(ev[1])
            self.canonic_cov = ca[2]
            canonic_ids = node_ids(ca[1])
            self.event_nodes, self.canonic_nodes = [ev[1], "."], [ca[1]]
            self.event_names, self.canonic_names = [ev[0], "."], [ca[0]]
        else:
            raise AttributeError(f"unknown event type {et!r}")
        # what eq_event compares (quantify.py:218-231)
        self.key = (self.chrom, self.gene, event_ids) if et == "IR" else (self.chrom, self.gene, event_ids, canonic_ids)
        self.replicates = []
        self.add_replicate(replicate, self.event_cov, self.canonic_cov)

    def add_replicate(self, replicate: int, event_cov, canonic_cov):
        while len(self.replicates) <= replicate:     # replicates this event was missing from count as (0, 0)
            self.replicates.append([0, 0])
        self.replicates[replicate] = [event_cov, canonic_cov]

    def psi(self):
        """mean over the replicates that have reads of canonical / (canonical + event); -1 if none has"""
        vals = []
        for ev, ca in self.replicates:
            den = ca + ev
            if den != 0:
                vals.append(float(ca) / den)
        # the reference marks "no reads" with -1 and then drops every value equal to -1 (quantify.py:205-209); with a
        # negative -w a '.' coverage (-1) can make a real ratio come out as exactly -1.0, which is dropped alike
        vals = [v for v in vals if v != -1]
        return mean(vals) if vals else -1

    def weights(self) -> str:
        return f"{_int_mean([r[1] for r in self.replicates])}/{_int_mean([r[0] for r in self.replicates])}"

    def prefix(self) -> str:
        return ",".join([self.etype, self.annotation_type, self.chrom, self.gene, self.strand,
                         ",".join(self.canonic_names), ",".join(self.event_names),
                         ",".join(self.canonic_nodes), ",".join(self.event_nodes)])


def read_condition(paths, w: int) -> dict:
    """{etype: {key: Event}} in first-seen order (dicts keep insertion order), replicates merged"""
    events = {t: {} for t in ETYPES}
    for i, path in enumerate(paths):
        per_type = {t: [] for t in ETYPES}
        with open(path) as f:
            for line in f:
                fields = line.strip().split(",")
                if fields[0] == "event_type":
                    continue
                e = Event(fields, i)
                per_type[e.etype].append(e)     # the reference takes a file type by type (quantify.py:258-259)
        for t in ETYPES:
            table = events[t]
            for e in per_type[t]:
                if e.canonic_cov < w or e.event_cov < w:
                    continue
                first = table.get(e.key)
                if first is None:
                    table[e.key] = e
                else:
                    first.add_replicate(i, e.event_cov, e.canonic_cov)
    return events


_RAISES = {"TypeError": TypeError, "ValueError": ValueError, "AttributeError": AttributeError, "FileNotFoundError": FileNotFoundError,
           "OSError": OSError}


def quantify(c1, c2, w: int = 0, both: bool = False, out=sys.stdout):
    """the native join of libpantas_gpu (csrc/quant.cpp: all host threads, hash join sharded by key) -- what the command line runs;
    `quantify_python` below is the same computation in the interpreter (4e4 rows/s: fine for a gene panel, hours for six
    whole-genome replicates) and what the native code is held to (tests/test_quant_remap.py)."""
    import ctypes as C
    import os
    import tempfile
    try:
        from ._lib import PantasError, lib
    except (ImportError, OSError) as e:
        # pure host work, but the native form lives in libpantas_gpu.so (HIP-linked): say what the way out is instead of a bare
        # loader error -- the interpreted form is an explicit choice, never a silent substitute
        raise ImportError(f"{e}; this command is host-only: PANTAS_QUANT=python runs its interpreted form without the library") from e
    n = len(c1)
    if len(c2) != n:
        raise ValueError("the same number of replicates for each condition")
    a = (C.c_char_p * n)(*[os.fsencode(p) for p in c1])
    b = (C.c_char_p * n)(*[os.fsencode(p) for p in c2])
    try:
        fd = out.fileno()
        direct = True
    except (AttributeError, OSError, ValueError):
        direct = False
    if direct:
        out.flush()
        rc = lib.pantas_quant_run(a, b, n, int(w), int(bool(both)), fd)
    else:                       # a text stream without a descriptor (tests): through a temporary file
        with tempfile.TemporaryFile() as f:
            rc = lib.pantas_quant_run(a, b, n, int(w), int(bool(both)), f.fileno())
            f.seek(0)
            if rc == 0:
                out.write(f.read().decode())
    if rc != 0:
        kind = (lib.pantas_last_error_kind() or b"").decode()
        msg = (lib.pantas_last_error() or b"").decode()
        raise _RAISES.get(kind, PantasError)(msg) if kind in _RAISES else PantasError(rc, kind, msg)


def quantify_python(c1, c2, w: int = 0, both: bool = False, out=sys.stdout):
    ev1 = read_condition(c1, w)
    ev2 = read_condition(c2, w)
    print(HEADER, file=out)
    for t in ETYPES:
        t1, t2 = ev1[t], ev2[t]
        for key, e1 in t1.items():
            e2 = t2.get(key)
            if e2 is not None:
                psi1, psi2 = e1.psi(), e2.psi()
                dpsi = max(0, psi1) - max(0, psi2)
                if psi1 == -1 and psi2 == -1:
                    dpsi = -1
                print(e1.prefix(), e1.weights(), e2.weights(), psi1, psi2, dpsi, sep=",", file=out)
            elif not both:
                psi1 = e1.psi()
                if not psi1 == -1:
                    print(e1.prefix(), e1.weights(), ".", psi1, 1, max(0, psi1) - max(0, 1), sep=",", file=out)
        if not both:
            for key, e2 in t2.items():
                if key not in t1:
                    psi2 = e2.psi()
                    if not psi2 == -1:
                        print(e2.prefix(), ".", e2.weights(), 1, psi2, max(0, 1) - max(0, psi2), sep=",", file=out)


def build_parser():
    p = argparse.ArgumentParser(prog="Quantify", description="")
    p.add_argument("--c1", help="Replicates of condition 1", dest="c1", type=str, required=True, nargs="+")
    p.add_argument("--c2", help="Replicates of condition 2", dest="c2", type=str, required=True, nargs="+")
    p.add_argument("--both", dest="both", help="Report only events present in both conditions", default=False, action="store_true")
    p.add_argument("-w", dest="w", help="Minimum value of read counts on each junction of the event to be valid (Default: 0)",
                   type=int, default=0)
    return p


def _interpreted() -> bool:
    import os
    return os.environ.get("PANTAS_QUANT", "") == "python"


def main(argv=None):
    args = build_parser().parse_args(argv)
    if not len(args.c1) == len(args.c2):
        print("Provide the same number of replicates for each condition")
        return 0
    (quantify_python if _interpreted() else quantify)(args.c1, args.c2, args.w, args.both)
    return 0


if __name__ == "__main__":
    sys.exit(main())
