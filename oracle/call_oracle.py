"""CPU oracle for the AS-event calling path.  TEST INFRASTRUCTURE ONLY.

Independent restatement of the algorithm of the reference script ``scripts/call.py``
(``main`` at :132-1105).  Only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it; the product package never does.

Parity status: PINNED -- ``tests/test_oracle_golden.py`` checks this module against
outputs of the real reference script stored under ``tests/golden/`` (made by
``tools/make_golden.py``).  The reference's row order and a few of its choices depend on
``PYTHONHASHSEED`` (set iteration); the goldens therefore hold, per case,
the rows every hash seed agrees on plus the set of seed-dependent alternatives, and this
module resolves each such choice with a fixed rule that is one of the reference's
possible behaviours:

  R1  rows are emitted in a fixed order (pass, then junction in file order);
  R2  "pick any exon of a set" (:484-486) picks the tag that appears first in the GFA;
  R3  ``set.pop()`` in the intronic search (:743, :796) works level by level -- the members
      the set held when ``pvisitl`` was last set are popped, in GFA order, before anything
      added since (round 5; until round 4: always the member that appears first in the GFA,
      which spends the ``--isw`` budget on the lowest branch and misses rows the reference
      prints under 61 of 64 hash seeds); "any node of visit & targets" (:752, :805) is the
      one that appears first in the GFA;
  R4  '|'-joined set contents (:876, :1072-1077) are listed in first-appearance order.

Differences from the reference by design: junctions sharing a head/tail are found through
an index instead of a linear scan (:280-282, :945-947) -- same sets, not the same cost.
"""
from __future__ import annotations

import re
import sys

from augment_oracle import ReferenceCrash  # noqa: F401  (same crash vocabulary)

HEADER = ("event_type,annotated/novel,chrom,gene,strand,transcripts1,transcripts2,transcripts3,"
          "nodes1,coverage1,nodes2,coverage2,nodes3,coverage3")

TX_FEATURES = {"mRNA", "transcript", "miRNA", "ncRNA", "pre_miRNA", "snoRNA", "pseudogene"}


def tx_of_junction_tag(tag: str) -> str:
    """[:98-104] 'T_H.k.k2' -> 'T'"""
    ht = ".".join(tag.split(".")[:-2])
    return "_".join(ht.split("_")[:-1])


def tx_of_exon_tag(tag: str) -> str:
    """[:107-121] 'T_H.k' -> 'T'"""
    ht = ".".join(tag.split(".")[:-1])
    return "_".join(ht.split("_")[:-1])


def tx_for_filter(tag: str) -> str:
    """the inline filter at :356, :364, :595, :603"""
    return "_".join(tag.split("_")[:-1])


def ceil_mean(total: int, n: int) -> int:
    """[:517-520] ceil(sum / len) -- exact in integers for the magnitudes involved."""
    return -((-total) // n)


class Graph:
    def __init__(self):
        self.seg = {}        # name -> dict(LN, NC?, EX?, I, O, rank)
        self.link = {}       # (u, v) -> dict(RC?, JN?, ID?)
        self.junctions = {}  # ordered set
        self.novel = {}      # ordered set
        self.exon_rank = {}  # EX tag -> first-appearance rank (rule R2/R4)
        self.by_head = {}
        self.by_tail = {}
        self.nov_by_head = {}
        self.nov_by_tail = {}


def _position_clusters_raise(v, ctx):
    """What is left of the IL/OL clustering [:37-66, collapse_linkcounts :15-21]: the exceptions.

    One entry: only ``max(_v, key=x[1])`` [:66] looks at it.  Several: one cluster when first minimum and first maximum
    position are closer than 3 [:43-45], else a low and a high cluster [:47-61] seeded with those two; entries whose whole
    list equals one of the seeds are left out [:49-50].  Every cluster reads x[1] of its members, then divides by their sum;
    the low cluster is summed first.
    """
    def total(members):
        if any(len(x) < 2 for x in members):
            raise ReferenceCrash("IndexError", f"{ctx}: entry without a count")
        if sum(x[1] for x in members) == 0:
            raise ReferenceCrash("ZeroDivisionError", f"{ctx}: counts of a cluster sum to 0")

    if len(v) == 1:
        if len(v[0]) < 2:
            raise ReferenceCrash("IndexError", f"{ctx}: entry without a count")
        return
    lo = min(v, key=lambda x: x[0])
    hi = max(v, key=lambda x: x[0])
    if hi[0] - lo[0] < 3:
        total(v)
        return
    rest = [x for x in v if x != lo and x != hi]
    total([lo] + [x for x in rest if x[0] - lo[0] < hi[0] - x[0]])
    total([hi] + [x for x in rest if not x[0] - lo[0] < hi[0] - x[0]])


def _attrs(fields, crash_ctx):
    """[:25-69] tag list -> dict."""
    out = {}
    for f in fields:
        parts = f.split(":")
        if len(parts) != 3:
            raise ReferenceCrash("ValueError", f"{crash_ctx}: tag {f!r} does not split in 3 on ':'")
        name, _, value = parts
        if name in ("LN", "RC", "NC"):
            try:
                out[name] = int(value)
            except ValueError:
                raise ReferenceCrash("ValueError", f"{crash_ctx}: {f!r}") from None
        elif name in ("IL", "OL"):
            # parsed and clustered by the reference [:33-66]; nothing reads the result, but the clustering can raise
            try:
                out[name] = [list(map(int, x.split("."))) for x in value.split(",")]
            except ValueError:
                raise ReferenceCrash("ValueError", f"{crash_ctx}: {f!r}") from None
            _position_clusters_raise(out[name], f"{crash_ctx}: {f!r}")
        else:
            out[name] = value.split(",")
    return out


def load_gfa(path: str) -> Graph:
    """[:145-181]"""
    g = Graph()
    with open(path) as fh:
        for lineno, line in enumerate(fh, start=1):
            line = line.strip()
            if line.startswith("S"):
                tok = line.split()
                if len(tok) < 3:
                    raise ReferenceCrash("ValueError", f"GFA line {lineno}: short S line")
                a = _attrs(tok[3:], f"GFA line {lineno}")
                a["LN"] = len(tok[2])
                a["I"] = []
                a["O"] = []
                a["rank"] = len(g.seg)
                g.seg[tok[1]] = a
                for e in a.get("EX", ()):
                    g.exon_rank.setdefault(e, len(g.exon_rank))
            elif line.startswith("P"):
                if len(line.split()) != 4:
                    raise ReferenceCrash("ValueError", f"GFA line {lineno}: P line needs 4 fields")
            elif line.startswith("L"):
                tok = line.split()
                if len(tok) < 6:
                    raise ReferenceCrash("ValueError", f"GFA line {lineno}: short L line")
                u, v = tok[1], tok[3]
                a = _attrs(tok[6:], f"GFA line {lineno}")
                g.link[(u, v)] = a
                if u not in g.seg or v not in g.seg:
                    raise ReferenceCrash("KeyError", f"GFA line {lineno}: link end without S line")
                g.seg[u]["O"].append(v)
                g.seg[v]["I"].append(u)
                if "JN" in a:
                    g.junctions[(u, v)] = True
                if "ID" in a:
                    g.novel[(u, v)] = True
    for j in g.junctions:
        g.by_head.setdefault(j[0], []).append(j)
        g.by_tail.setdefault(j[1], []).append(j)
    for j in g.novel:
        g.nov_by_head.setdefault(j[0], []).append(j)
        g.nov_by_tail.setdefault(j[1], []).append(j)
    return g


_GID = re.compile('gene_id "[A-Za-z0-9_]+";')
_TID = re.compile('transcript_id "[A-Za-z0-9_]+";')


def load_gtf(path: str):
    """[:188-216]"""
    t2g, strand, chrom = {}, {}, {}
    with open(path) as fh:
        for lineno, line in enumerate(fh, start=1):
            if line.startswith("#"):
                continue
            col = line.strip("\n").split("\t")
            if len(col) < 3:
                raise ReferenceCrash("IndexError", f"GTF line {lineno}")
            if col[2] not in TX_FEATURES:
                continue
            mg = _GID.search(col[-1])
            mt = _TID.search(col[-1])
            if mg is None or mt is None:
                raise ReferenceCrash("AttributeError", f"GTF line {lineno}: gene_id/transcript_id")
            gid = mg.group(0).split('"')[-2]
            tid = mt.group(0).split('"')[-2]
            if len(col) < 7:
                raise ReferenceCrash("IndexError", f"GTF line {lineno}")
            t2g[tid] = gid
            strand[gid] = col[6]
            chrom[gid] = col[0]
    return t2g, strand, chrom


class Caller:
    def __init__(self, g: Graph, t2g, strand, chrom, rca=-1, novel=False, no_annotated=False,
                 events=("ES", "SS", "IR"), isw=5, minintronsize=100, junction=None, err=None):
        self.g, self.t2g, self.strand, self.chrom = g, t2g, strand, chrom
        self.rca, self.do_novel, self.no_annotated = rca, novel, no_annotated
        self.events = set(events)
        self.isw, self.minintron, self.junction = isw, minintronsize, junction
        self.err = err
        self.rows = []

    # ---- small helpers ----------------------------------------------------------------
    def ex(self, n):
        return set(self.g.seg[n].get("EX", ()))

    def gene_of(self, tx, ctx):
        if tx not in self.t2g:
            raise ReferenceCrash("KeyError", f"{ctx}: transcript {tx!r} not in GTF")
        return self.t2g[tx]

    def jn(self, j):
        return self.g.link[j]["JN"]

    def rc(self, j):
        a = self.g.link[j]
        if "RC" not in a:
            raise ReferenceCrash("KeyError", f"link {j}: no RC tag")
        return a["RC"]

    def nc(self, n):
        a = self.g.seg[n]
        if "NC" not in a:
            raise ReferenceCrash("KeyError", f"segment {n}: no NC tag")
        return a["NC"]

    def jn_txs(self, j):
        return {tx_of_junction_tag(t) for t in self.jn(j)}

    def jn_genes(self, j):
        return {self.gene_of(t, f"junction {j}") for t in self.jn_txs(j)}

    def ex_txs(self, exons):
        return {tx_of_exon_tag(e) for e in exons}

    def by_rank(self, exons):
        return sorted(exons, key=lambda e: self.g.exon_rank[e])

    def first_node(self, nodes):
        return min(nodes, key=lambda n: self.g.seg[n]["rank"])

    def row(self, *cols):
        self.rows.append(",".join(str(c) for c in cols))

    def skip_msg(self, what):
        if self.err is not None:
            print(f"Skipping {what} due to multiple genes", file=self.err)

    def same_gene_exons(self, n, genes, ctx):
        return {e for e in self.ex(n) if self.gene_of(tx_of_exon_tag(e), ctx) in genes}

    def mean_nc(self, path):
        return ceil_mean(sum(self.nc(x) if x in self.g.seg else 0 for x in path), len(path))

    def walk_exon(self, start, goal, ok, what):
        """[:488-499, :843-852] follow the first out-neighbour that sorts <= goal (as a
        string) and satisfies ``ok``; when none does, the reference's loop variable is
        left on the last out-neighbour and that one is taken."""
        path = [start]
        guard = len(self.g.seg) + 2
        while path[-1] != goal:
            outs = self.g.seg[path[-1]]["O"]
            if not outs:
                raise ReferenceCrash("AssertionError", f"Error while reconstrucing {what} path")
            nxt = outs[-1]
            for nn in outs:
                if nn <= goal and ok(nn):
                    nxt = nn
                    break
            path.append(nxt)
            guard -= 1
            if guard == 0:
                raise ReferenceCrash("Hang", f"{what} walk does not terminate")
        return path

    # ---- annotated junctions ---------------------------------------------------------
    def annotated(self):
        g = self.g
        for j in g.junctions:
            if self.junction is not None and "f{_j[0]}-{_j[1]}" != self.junction:  # [:241] sic
                continue
            if self.rc(j) < self.rca:
                continue
            u, v = j
            txs = self.jn_txs(j)
            genes = {self.gene_of(t, f"junction {j}") for t in txs}
            if len(genes) > 1:
                continue
            gene = next(iter(genes))
            ex_u = self.same_gene_exons(u, genes, f"junction {j}")
            ex_v = self.same_gene_exons(v, genes, f"junction {j}")
            if not ex_u or not ex_v:
                raise ReferenceCrash("AssertionError", f"junction {j}: no exon of its gene on an end")
            heads = [x for x in g.by_head.get(u, ()) if x != j and self.jn_genes(x) & genes]
            tails = [x for x in g.by_tail.get(v, ()) if x != j and self.jn_genes(x) & genes]
            chrom, strand = self.chrom[gene], self.strand[gene]
            if "ES" in self.events:                                     # [:316-375]
                for j1 in heads:
                    for j2 in tails:
                        inc = (self.jn_txs(j1) & self.jn_txs(j2)) - txs
                        if not inc:
                            continue
                        if (self.ex(j1[1]) | self.ex(j2[0])) - (ex_u | ex_v):
                            self.row("ES", "annotated", chrom, gene, strand, "|".join(self.jn(j)),
                                     "|".join(t for t in self.jn(j1) if tx_for_filter(t) in inc),
                                     "|".join(t for t in self.jn(j2) if tx_for_filter(t) in inc),
                                     ">".join(j), self.rc(j), ">".join(j1), self.rc(j1),
                                     ">".join(j2), self.rc(j2))
            if "SS" in self.events:                                     # [:377-463]
                if tails:
                    for n in g.seg[u]["O"]:
                        if n == v:
                            continue
                        cont = self.ex(n) & ex_u
                        if not cont:
                            continue
                        for j2 in tails:
                            if j2[0] == u:
                                continue
                            shared = self.ex(j2[0]) & cont
                            if not shared:
                                continue
                            if not (self.ex_txs(shared) - txs):
                                continue
                            if set(self.jn(j)) & set(self.jn(j2)):
                                raise ReferenceCrash("AssertionError", f"junctions {j} {j2} share a JN tag")
                            self.row("A5" if strand == "+" else "A3", "annotated", chrom, gene, strand,
                                     "|".join(self.jn(j)), "|".join(self.jn(j2)), ".",
                                     ">".join(j), self.rc(j), ">".join(j2), self.rc(j2), ".", ".")
                if heads:
                    for n in g.seg[v]["I"]:
                        if n == u:
                            continue
                        cont = self.ex(n) & ex_v
                        if not cont:
                            continue
                        for j1 in heads:
                            if j1[1] == v:
                                continue
                            shared = self.ex(j1[1]) & cont
                            if not shared:
                                continue
                            if not (self.ex_txs(shared) - txs):
                                continue
                            self.row("A3" if strand == "+" else "A5", "annotated", chrom, gene, strand,
                                     "|".join(self.jn(j)), "|".join(self.jn(j1)), ".",
                                     ">".join(j), self.rc(j), ">".join(j1), self.rc(j1), ".", ".")
            if "IR" in self.events:                                     # [:465-524]
                both = ex_u & ex_v
                found = None
                for n1 in g.seg[u]["O"]:
                    for n2 in g.seg[v]["I"]:
                        if n1 == v or n2 == u:
                            continue
                        if n2 < n1:
                            continue
                        cand = self.ex(n1) & self.ex(n2) & both
                        if not cand:
                            continue
                        e = self.by_rank(cand)[0]                        # rule R2
                        found = self.walk_exon(n1, n2, lambda nn, e=e: e in self.ex(nn), "IR")
                        break
                    if found is not None:
                        break
                if found is not None:
                    self.row("IR", "annotated", chrom, gene, strand, "|".join(self.jn(j)), ".", ".",
                             ">".join(j), self.rc(j), ">".join(found), self.mean_nc(found), ".", ".")

    # ---- novel links ---------------------------------------------------------------------
    def search(self, start, targets, key):
        """[:734-750, :787-803] the level-bounded frontier search, with rule R3 for pop().

        `visit` is the reference's set; `level_nodes` are the `pvisitl` members it held when the level began, in GFA order --
        they are popped before anything the level adds (a node popped and added again belongs to the next level).  The loop
        condition is looked at before every pop, as there: a hit on the level's LAST pop is seen with the level already counted.
        """
        visit = {start}
        level_nodes = [start]
        left = 1
        level = 0
        while not (visit & targets) and level < self.isw:
            if not visit:
                return None
            n = level_nodes.pop(0)
            visit.discard(n)
            left -= 1
            visit |= set(self.g.seg[n][key])
            if left == 0:
                level += 1
                left = len(visit)
                level_nodes = sorted(visit, key=lambda x: self.g.seg[x]["rank"])
        if level < self.isw:
            return self.first_node(visit & targets)
        return None

    def novel_links(self):
        g = self.g
        for j in g.novel:
            if self.junction is not None and f"{j[0]}-{j[1]}" != self.junction:
                continue
            if self.rc(j) < self.rca:
                continue
            u, v = j
            ex0, ex1 = self.ex(u), self.ex(v)
            if not ex0 and not ex1:
                continue
            cap = self.ex_txs(ex0) & self.ex_txs(ex1)
            nxt, prv = g.seg[u]["O"], g.seg[v]["I"]
            if "ES" in self.events:                                     # [:550-614]
                if ex0 and ex1 and len(ex0 & ex1) != len(ex0):
                    n1 = [n for n in nxt if (u, n) in g.junctions]
                    n2 = [p for p in prv if (p, v) in g.junctions]
                    for n in n1:
                        for p in n2:
                            j1, j2 = (u, n), (p, v)
                            inc = self.jn_txs(j1) & self.jn_txs(j2)
                            if not inc:
                                continue
                            genes = {self.gene_of(t, f"novel {j}") for t in inc}
                            if len(genes) > 1:
                                self.skip_msg("ES")
                                continue
                            gene = next(iter(genes))
                            self.row("ES", "novel", self.chrom[gene], gene, self.strand[gene], "?",
                                     "|".join(t for t in self.jn(j1) if tx_for_filter(t) in inc),
                                     "|".join(t for t in self.jn(j2) if tx_for_filter(t) in inc),
                                     ">".join(j), self.rc(j), ">".join(j1), self.rc(j1),
                                     ">".join(j2), self.rc(j2))
            if "SS" in self.events:
                if cap:                                                  # [:617-726]
                    for side in (0, 1):
                        spliced = ex1 if side == 0 else ex0
                        if not spliced:
                            continue
                        hits = []
                        for n in (nxt if side == 0 else prv):
                            shared = self.ex(n) & spliced
                            if not shared:
                                continue
                            jj = (u, n) if side == 0 else (n, v)
                            keep = self.ex_txs(shared) & cap
                            if jj in g.junctions and keep:
                                hits.append((jj, keep))
                        for j1, keep in hits:
                            genes = {self.gene_of(t, f"novel {j}") for t in keep}
                            if len(genes) > 1:
                                self.skip_msg("SS")
                                continue
                            gene = next(iter(genes))
                            plus = self.strand[gene] == "+"
                            label = ("A3" if plus else "A5") if side == 0 else ("A5" if plus else "A3")
                            self.row(label, "novel", self.chrom[gene], gene, self.strand[gene], "?",
                                     "|".join(self.jn(j1)), ".", ">".join(j), self.rc(j),
                                     ">".join(j1), self.rc(j1), ".", ".")
                # intronic splice site [:730-835]
                j1 = None
                if ex0 and not ex1:
                    targets = {n for n in nxt if (u, n) in g.junctions}
                    hit = self.search(v, targets, "O")
                    if hit is not None:
                        j1 = (u, hit)
                elif not ex0 and ex1:
                    targets = {p for p in prv if (p, v) in g.junctions}
                    hit = self.search(u, targets, "I")
                    if hit is not None:
                        j1 = (hit, v)
                if j1 is not None:
                    genes = self.jn_genes(j1)
                    if len(genes) > 1:
                        self.skip_msg("SS")
                    else:
                        gene = next(iter(genes))
                        self.row("A3" if self.strand[gene] == "+" else "A5", "novel", self.chrom[gene],
                                 gene, self.strand[gene], "|".join(self.jn(j1)), "?", ".",
                                 ">".join(j1), self.rc(j1), ">".join(j), self.rc(j), ".", ".")
            if "IR" in self.events and cap:                              # [:837-893]
                inside = ex0 & ex1
                if inside:
                    path = self.walk_exon(u, v, lambda nn: bool(inside & self.ex(nn)), "novel IR")
                    if sum(g.seg[x]["LN"] for x in path) >= self.minintron:
                        genes = {self.gene_of(t, f"novel {j}") for t in self.ex_txs(inside)}
                        if len(genes) > 1:
                            self.skip_msg("novel IR")
                        else:
                            gene = next(iter(genes))
                            self.row("IR", "novel", self.chrom[gene], gene, self.strand[gene], "?",
                                     "|".join(self.by_rank(inside)), ".", ">".join(j), self.rc(j),
                                     ">".join(path), self.mean_nc(path), ".", ".")

    # ---- annotated junctions again: cassette exons and read-through introns -----------------
    def novel_around_annotated(self):
        g = self.g
        if not ({"IR", "ES"} & self.events):
            return
        for j in g.junctions:
            if self.junction is not None and f"{j[0]}-{j[1]}" != self.junction:
                continue
            if self.rc(j) < self.rca:
                continue
            u, v = j
            genes = self.jn_genes(j)
            if len(genes) > 1:
                continue
            gene = next(iter(genes))
            ex0 = self.same_gene_exons(u, genes, f"junction {j}")
            ex1 = self.same_gene_exons(v, genes, f"junction {j}")
            if not ex0 or not ex1:
                raise ReferenceCrash("AssertionError", f"junction {j}: no exon of its gene on an end")
            chrom, strand = self.chrom[gene], self.strand[gene]
            if "ES" in self.events:                                     # [:954-984]
                h = [x for x in g.nov_by_head.get(u, ()) if x != j]
                t = [x for x in g.nov_by_tail.get(v, ()) if x != j]
                cas = {}
                for j1 in h:
                    for j2 in t:
                        if j1[1] <= j2[0]:
                            cas[(j1[1], j2[0])] = True
                for es, ee in cas:
                    a, b = (u, es), (ee, v)
                    self.row("CE", "novel", chrom, gene, strand, "|".join(self.jn(j)), "?", "?",
                             ">".join(j), self.rc(j), ">".join(a), self.rc(a), ">".join(b), self.rc(b))
            if "IR" in self.events:                                     # [:985-1102]
                try:
                    iu, iv = int(u), int(v)
                except ValueError:
                    raise ReferenceCrash("ValueError", f"junction {j}: non-integer segment name") from None
                left = [self.rc((u, str(x))) >= self.rca for x in range(iu + 1, iu + 4)
                        if (u, str(x)) in g.link and (u, str(x)) not in g.junctions]
                if not any(left):
                    continue
                right = [self.rc((str(x), v)) >= self.rca for x in range(iv - 3, iv)
                         if (str(x), v) in g.link and (str(x), v) not in g.junctions]
                if not any(right):
                    continue
                ends = [e for e in self.by_rank(ex0) if all(e not in self.ex(x) for x in g.seg[u]["O"])]
                starts = [e for e in self.by_rank(ex1) if all(e not in self.ex(x) for x in g.seg[v]["I"])]
                pairs = [(a, b) for a in ends for b in starts if tx_of_exon_tag(a) == tx_of_exon_tag(b)]
                if not pairs:
                    continue
                path = [u]
                guard = len(g.seg) + 2
                while path[-1] != v:
                    outs = [x for x in g.seg[path[-1]]["O"] if x <= v]
                    if len(path) == 1:
                        outs.remove(v)
                    if v in outs:
                        path.append(v)
                        continue
                    outs = [x for x in outs if g.seg[x]["O"]]
                    if not outs:
                        raise ReferenceCrash("ValueError", f"junction {j}: max() arg is an empty sequence")
                    best = outs[0]
                    for x in outs[1:]:
                        if self.nc(x) > self.nc(best):
                            best = x
                    self.nc(best)
                    path.append(best)
                    guard -= 1
                    if guard == 0:
                        raise ReferenceCrash("Hang", "novel IR (2) walk does not terminate")
                if sum(g.seg[x]["LN"] for x in path[1:-1]) < self.minintron:
                    continue
                genes2 = {self.gene_of(t, f"junction {j}") for t in self.ex_txs(a for a, _ in pairs)}
                if len(genes2) > 1:
                    self.skip_msg("novel IR")
                    continue
                gene2 = next(iter(genes2))
                jann = "|".join(a + "." + b.split(".")[-1] for a, b in pairs)
                self.row("IR", "novel", self.chrom[gene2], gene2, self.strand[gene2], jann, "?", ".",
                         ">".join(path), self.mean_nc(path), ">".join(j), self.rc(j), ".", ".")

    def run(self):
        if not self.no_annotated:
            self.annotated()
        if self.do_novel:
            self.novel_links()
            self.novel_around_annotated()
        return self.rows


def call(gfa_path, gtf_path, out=None, **opts):
    g = load_gfa(gfa_path)
    t2g, strand, chrom = load_gtf(gtf_path)
    c = Caller(g, t2g, strand, chrom, **opts)
    if out is not None:
        out.write(HEADER + "\n")
    try:
        c.run()
    finally:
        if out is not None:
            for r in c.rows:
                out.write(r + "\n")
    return c.rows


def normalise_row(row: str) -> str:
    """The parity comparator's only normalisation: on 'IR,novel' rows the
    '|'-joined contents of columns 6 and 7 come from Python set iteration -- sort them."""
    c = row.split(",")
    if len(c) == 14 and c[0] == "IR" and c[1] == "novel":
        c[5] = "|".join(sorted(c[5].split("|")))
        c[6] = "|".join(sorted(c[6].split("|")))
    return ",".join(c)


if __name__ == "__main__":
    import argparse
    ap = argparse.ArgumentParser()
    ap.add_argument("GFA")
    ap.add_argument("GTF")
    ap.add_argument("--rca", type=int, default=-1)
    ap.add_argument("--novel", action="store_true")
    ap.add_argument("--no-annotated", dest="no_annotated", action="store_true")
    ap.add_argument("--events", nargs="+", default=["ES", "SS", "IR"])
    ap.add_argument("--isw", type=int, default=5)
    ap.add_argument("--minintronsize", type=int, default=100)
    ap.add_argument("--junction", default=None)
    a = ap.parse_args()
    call(a.GFA, a.GTF, sys.stdout, rca=a.rca, novel=a.novel, no_annotated=a.no_annotated,
         events=a.events, isw=a.isw, minintronsize=a.minintronsize, junction=a.junction, err=sys.stderr)
