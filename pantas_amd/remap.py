"""`pantas remap`: reference-genome coordinates for the junctions of a quantification CSV.

Same command line, columns, row order and number printing as the reference's ``scripts/remap.py``: `remap` runs the native form
(csrc/remap.cpp, `pantas_remap_run`), `remap_python` is the line-by-line mirror the native form is tested against.
For each of the three junction names of a row the reference transcript copy (``<tid>_R<k>.<exon>[.<exon>]``) is
looked up in the GTF's exon table (``remap.py:45-63``); one exon index = retained intron reported as the whole
exon, two = the intron between them (``:118-142``).  Annotated alternative-splice-site rows are put shorter
junction first and IR rows are moved to the first slot (``:144-186``), rows are labelled reference / haplotype
(``:188-209``) and novel rows with a region shorter than ``-i`` are dropped (``:211-219``).

    python -m pantas_amd.remap [-i MIN_INTRON] quant.csv annotation.gtf
"""
from __future__ import annotations

import argparse
import re
import sys

HEADER = ("etype,annotation_type,haplotype_type,chrom,gene,strand,junction1_name,junction2_name,junction3_name,"
          "junction1_nodes,junction2_nodes,junction3_nodes,junction1_positions,junction2_positions,junction3_positions,"
          "W1,W2,psi_c1,psi_c2,dpsi")
TRANSCRIPT_ID = re.compile('transcript_id "[A-Za-z0-9_]+";')


def reference_copy(names: str) -> str:
    """first '|'-separated name whose last '_' piece starts with R (the _R<k> copy), '?' if none (remap.py:21-26)"""
    for x in names.split("|"):
        if x.split("_")[-1][0] == "R":
            return x
    return "?"


def transcript_of(name: str) -> str:
    return "_".join(name.split(".")[0].split("_")[:-1])


def span(region: str):
    """(start, end) of 'chrom:start-end' with an optional trailing '?' (remap.py:4-12)"""
    if region.endswith("?"):
        region = region[:-1]
    s, e = region.split(":")[1].split("-")
    return int(s), int(e)


def length_strict(region: str) -> int:
    """end - start; a trailing '?' is not accepted here, as in remap.py:15-18"""
    s, e = region.split(":")[1].split("-")
    return int(e) - int(s)


def data_rows(path):
    with open(path) as f:
        for line in f:
            if not line.startswith("etype"):
                yield line.strip("\n").split(",")


def exon_table(csv_path: str, gtf_path: str) -> dict:
    wanted = {}
    for fields in data_rows(csv_path):
        names1, names2, names3 = fields[5:8]
        for names in (names1, names2, names3):
            t = reference_copy(names)
            if t != "?":
                wanted[transcript_of(t)] = []
    with open(gtf_path) as f:
        for line in f:
            if line.startswith("#"):
                continue
            cols = line.strip("\n").split("\t")
            if cols[2] != "exon":
                continue
            tid = TRANSCRIPT_ID.search(cols[-1]).group(0).split('"')[-2]
            if tid in wanted:
                wanted[tid].append((int(cols[3]), int(cols[4])))
    for exons in wanted.values():
        exons.sort()
    return wanted


def position(name: str, chrom: str, exons_of: dict) -> str:
    if name == "?":
        return "?"
    head = name.split(".")[0]
    imprecise = head.split("_")[-1][0] == "H"
    exons = exons_of[transcript_of(name)]
    idx = [int(n) for n in name.split(".")[1:]]
    assert len(idx) <= 2
    if not idx:
        return "?"
    if len(idx) == 1:                       # retained intron: the whole exon
        s, e = exons[idx[0] - 1]
    else:                                   # the intron between the two exons
        s = exons[idx[0] - 1][1] + 1
        e = exons[idx[1] - 1][0] - 1
    assert s <= e
    return f"{chrom}:{s}-{e}" + ("?" if imprecise else "")


_RAISES = {"ValueError": ValueError, "IndexError": IndexError, "AttributeError": AttributeError, "AssertionError": AssertionError,
           "KeyError": KeyError, "FileNotFoundError": FileNotFoundError, "OSError": OSError}


def remap(csv_path: str, gtf_path: str, min_intron_size: int = 100, out=sys.stdout):
    """the native form in libpantas_gpu (csrc/remap.cpp: rows on all host threads) -- what the command line runs; `remap_python`
    below is the same computation in the interpreter and what the native code is held to (tests/test_quant_remap.py).
    PANTAS_REMAP=python runs the interpreted form."""
    import os
    if os.environ.get("PANTAS_REMAP") == "python":
        return remap_python(csv_path, gtf_path, min_intron_size, out)
    import tempfile
    try:
        from ._lib import PantasError, lib
    except (ImportError, OSError) as e:
        # pure host work, but the native form lives in libpantas_gpu.so (HIP-linked): say what the way out is instead of a bare
        # loader error -- the interpreted form is an explicit choice, never a silent substitute
        raise ImportError(f"{e}; this command is host-only: PANTAS_REMAP=python runs its interpreted form without the library") from e
    try:
        fd = out.fileno()
        direct = True
    except (AttributeError, OSError, ValueError):
        direct = False
    if direct:
        out.flush()
        rc = lib.pantas_remap_run(os.fsencode(csv_path), os.fsencode(gtf_path), int(min_intron_size), fd)
    else:                       # a text stream without a descriptor (tests): through a temporary file
        with tempfile.TemporaryFile() as f:
            rc = lib.pantas_remap_run(os.fsencode(csv_path), os.fsencode(gtf_path), int(min_intron_size), f.fileno())
            f.seek(0)
            out.write(f.read().decode())      # (also of a failed run: the rows in front of the failing one, as the script prints them)
    if rc != 0:
        kind = (lib.pantas_last_error_kind() or b"").decode()
        msg = (lib.pantas_last_error() or b"").decode()
        raise _RAISES[kind](msg) if kind in _RAISES else PantasError(rc, kind, msg)


def remap_python(csv_path: str, gtf_path: str, min_intron_size: int = 100, out=sys.stdout):
    exons_of = exon_table(csv_path, gtf_path)
    print(HEADER, file=out)
    for fields in data_rows(csv_path):
        (etype, annotation_type, chrom, gene, strand, names1, names2, names3,
         nodes1, nodes2, nodes3, w1, w2, psi1, psi2, dpsi) = fields
        name = [reference_copy(names1), reference_copy(names2), reference_copy(names3)]
        pos = [position(n, chrom, exons_of) for n in name]
        nodes = [nodes1, nodes2, nodes3]

        def swap_conditions():
            nonlocal w1, w2, psi1, psi2, dpsi
            w1, w2 = w2, w1
            psi1, psi2 = psi2, psi1
            dpsi = -float(dpsi)

        if annotation_type == "annotated":
            if etype[0] == "A":
                if pos[0] == "?":
                    # the second junction moves to the first slot -- except its position, which the reference
                    # overwrites again with the slot it has just cleared (remap.py:150-156): both end up '?'
                    name[0], name[1] = name[1], "?"
                    pos[0], pos[1] = "?", "?"
                    nodes[0], nodes[1] = nodes[1], "."
                    swap_conditions()
                elif pos[1] != "?":
                    a, b = span(pos[0]), span(pos[1])
                    if a[1] - a[0] > b[1] - b[0]:       # shorter junction first
                        name[0], name[1] = name[1], name[0]
                        pos[0], pos[1] = pos[1], pos[0]
                        nodes[0], nodes[1] = nodes[1], nodes[0]
                        swap_conditions()
            elif etype == "IR":
                name[0], name[1] = name[1], "?"
                pos[0], pos[1] = pos[1], "?"
                nodes[0], nodes[1] = nodes[1], "."
                swap_conditions()

        unknown = [p == "?" for p in pos]
        if annotation_type == "annotated":
            hap = ((etype == "ES" and any(unknown)) or (etype[0] == "A" and (unknown[0] or unknown[1]))
                   or (etype == "IR" and unknown[0]))
        else:
            hap = ((etype == "ES" and all(unknown)) or (etype[0] == "A" and unknown[0] and unknown[1])
                   or (etype == "IR" and unknown[0] and unknown[1]))
        if annotation_type == "novel" and any([length_strict(p) < min_intron_size for p in pos if p != "?"]):
            continue
        print(etype, annotation_type, "haplotype" if hap else "reference", chrom, gene, strand, name[0], name[1], name[2],
              nodes[0], nodes[1], nodes[2], pos[0], pos[1], pos[2], w1, w2, psi1, psi2, dpsi, sep=",", file=out)


def build_parser():
    p = argparse.ArgumentParser(prog="Remap", description="")
    p.add_argument("CSV", help="Quantification in CSV format")
    p.add_argument("GTF", help="Annotation in GTF format")
    p.add_argument("-i", dest="min_intron_size", help="Minimum intron size for a novel junction (default: 100)", type=int, default=100)
    return p


def main(argv=None):
    a = build_parser().parse_args(argv)
    remap(a.CSV, a.GTF, a.min_intron_size)
    return 0


if __name__ == "__main__":
    sys.exit(main())
