"""`pantas augment` front end: same argv, stdout and stderr as the reference script
``scripts/alignments_augmentation_from_gaf.py`` (``main(argv)`` at :110-113; stderr lines
:120,134,375,376), with the work done by libpantas_gpu on the MI355X.

    python -m pantas_amd.augment <GAF> <GFA> [thr=20] > augmented.gfa
"""
from __future__ import annotations

import os
import sys
import time

_T0 = time.perf_counter()

from .engine import Graph, PantasError, graph_cache_path  # noqa: E402


def _lap(what: str) -> None:
    if os.environ.get("PANTAS_TIMING"):
        print(f"[timing] cli augment: {what:<28} at {time.perf_counter() - _T0:8.3f} s since interpreter start of the module", file=sys.stderr)


def main(argv) -> int:
    gaf_file = argv[0]
    gfa_file = argv[1]
    thr = int(argv[2]) if len(argv) > 2 else 20
    device = int(os.environ.get("PANTAS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    print("Read GFA", file=sys.stderr)
    try:
        _lap("imports done")
        from ._lib import lib
        # HIP runtime start-up before the GFA parse (0.25 s): on a thread beside it, it was starved by the parse's page faults and
        # left 0.9 s for the moment the graph goes to the device; `call` has its table build between parse and upload and keeps the thread
        lib.pantas_device_init(device)
        g = Graph(gfa_file, cache=graph_cache_path(gfa_file))   # PANTAS_GRAPH_CACHE: the graph is shared by all replicates
        _lap("graph loaded")
        g.to_device(device)
        print("Augmentation by GAF alignments", file=sys.stderr)
        bs, _ = g.augment(gaf_file, thr, fd_out=-1)
        _lap("reads walked, counters on host")
        print(f"Rejected alignments: {bs.rejected}", file=sys.stderr)
        print("Annotating GFA", file=sys.stderr)
        sys.stdout.flush()
        g.release_device()      # the device's part is over: its memory is scrubbed beside the writing, not in front of the next process
        g.write_gfa(sys.stdout.fileno())
        _lap("GFA written")
    except PantasError as e:
        # the reference dies with a traceback and exit status 1, stdout still empty
        print(f"{e.kind}: {e}", file=sys.stderr)
        return 1
    return 0


def _leave(rc: int) -> None:
    """The output is complete: leave without tearing down 60 GB of graph arrays page by page (seconds at whole-genome scale).
    PANTAS_CLEAN_EXIT=1 takes the ordinary way out (atexit handlers and library finalisers run: what a profiler or a coverage
    tool around this process needs to write its own output)."""
    for stream in (sys.stdout, sys.stderr):
        try:
            stream.flush()
        except BrokenPipeError:       # the reader went away (`| head`): nothing left to deliver, no traceback either
            pass
    if os.environ.get("PANTAS_CLEAN_EXIT"):
        sys.exit(rc)
    os._exit(rc)


if __name__ == "__main__":
    _leave(main(sys.argv[1:]))
