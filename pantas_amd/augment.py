"""`pantas augment` front end: same argv, stdout and stderr as the reference script
``scripts/alignments_augmentation_from_gaf.py`` (``main(argv)`` at :110-113; stderr lines
:120,134,375,376), with the work done by libpantas_gpu on the MI355X.

    python -m pantas_amd.augment <GAF> <GFA> [thr=20] > augmented.gfa
"""
from __future__ import annotations

import os
import sys

from .engine import Graph, PantasError, graph_cache_path


def main(argv) -> int:
    gaf_file = argv[0]
    gfa_file = argv[1]
    thr = int(argv[2]) if len(argv) > 2 else 20
    device = int(os.environ.get("PANTAS_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    print("Read GFA", file=sys.stderr)
    try:
        g = Graph(gfa_file, cache=graph_cache_path(gfa_file))   # PANTAS_GRAPH_CACHE: the graph is shared by all replicates
        g.to_device(device)
        print("Augmentation by GAF alignments", file=sys.stderr)
        bs, _ = g.augment(gaf_file, thr, fd_out=-1)
        print(f"Rejected alignments: {bs.rejected}", file=sys.stderr)
        print("Annotating GFA", file=sys.stderr)
        sys.stdout.flush()
        g.write_gfa(sys.stdout.fileno())
    except PantasError as e:
        # the reference dies with a traceback and exit status 1, stdout still empty
        print(f"{e.kind}: {e}", file=sys.stderr)
        return 1
    return 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
