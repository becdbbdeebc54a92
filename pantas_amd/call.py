"""`pantas call` front end: the argparse interface, stdout (14-column CSV) and stderr of the
reference script ``scripts/call.py`` (:1108-1178), with the event enumeration done by
libpantas_gpu on the MI355X.

    python -m pantas_amd.call [--rca N] [--novel] [--no-annotated] [--events ES SS IR]
                              [--isw N] [--minintronsize N] [--debug] [--junction a-b] GFA GTF
"""
from __future__ import annotations

import argparse
import datetime
import os
import sys
import time

_T0 = time.perf_counter()

from .engine import Graph, PantasError  # noqa: E402


def _lap(what: str) -> None:
    if os.environ.get("PANTAS_TIMING"):
        print(f"[timing] cli call: {what:<28} at {time.perf_counter() - _T0:8.3f} s since interpreter start of the module", file=sys.stderr)

CSV_HEADER = ("event_type,annotated/novel,chrom,gene,strand,transcripts1,transcripts2,transcripts3,"
              "nodes1,coverage1,nodes2,coverage2,nodes3,coverage3")      # scripts/call.py:218-234


def build_parser() -> argparse.ArgumentParser:
    # same options, defaults and help as the reference parser (scripts/call.py:1111-1174)
    p = argparse.ArgumentParser(prog="Caller", description="")
    p.add_argument("GFA", help="Spliced pangenome in GFA format")
    p.add_argument("GTF", help="Annotation in GTF format")
    p.add_argument("--rca", help="Minimum read count for annotated events (default: -1)", dest="rca",
                   type=int, required=False, default=-1)
    p.add_argument("--novel", dest="novel", help="Call novel events (default: False)", action="store_true",
                   default=False)
    p.add_argument("--no-annotated", dest="annotated", help="Do not call known annotated events (default: False)",
                   action="store_true", default=False)
    p.add_argument("--events", help="Events to call (default: [ES, SS, IR])", dest="events", nargs="+",
                   required=False, default=["ES", "SS", "IR"])
    p.add_argument("--isw", dest="isw", type=int, default=5,
                   help="Intronic search window for novel events, larger values reduce FP but increase time (default: 5)")
    p.add_argument("--minintronsize", dest="minintronsize", help="Minimum intron size (default: 100)", type=int,
                   default=100)
    p.add_argument("--debug", dest="debug", help="Debug (default: False)", action="store_true", default=False)
    p.add_argument("--junction", dest="junction", help="Junction to check, in the form '1-2' (default: None)",
                   type=str, default=None)
    return p


def _stamp(msg: str) -> None:
    print(datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"), msg, file=sys.stderr)


def run(args) -> int:
    device = int(os.environ.get("PANTAS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    try:
        _stamp("Parsing GFA..")
        _lap("imports done")
        from ._lib import lib
        lib.pantas_device_init(device)         # HIP runtime start-up and the streams of the run, while the process is small
        g = Graph(args.GFA, for_call=True)
        _lap("graph loaded")
        _stamp("Parsing GTF..")
        g.call_prepare(args.GTF, device)
        _lap("tables built and uploaded")
        sys.stdout.flush()
        sys.stderr.flush()
        g.call(sys.stdout.fileno(), sys.stderr.fileno(), rca=args.rca, novel=args.novel,
               no_annotated=args.annotated, events=args.events, isw=args.isw,
               minintronsize=args.minintronsize, junction=args.junction, debug=args.debug,
               release_device=True)     # one run, then exit: the device memory is scrubbed beside the rendering, not in front of the next process
        _lap("events written")
    except PantasError as e:
        print(f"{e.kind}: {e}", file=sys.stderr)
        return 1
    return 0


def main(argv=None) -> int:
    return run(build_parser().parse_args(argv))


if __name__ == "__main__":
    from .augment import _leave
    _leave(main())      # (no page-by-page teardown of the graph arrays; PANTAS_CLEAN_EXIT=1 for the ordinary exit)
