"""Deterministic synthetic spliced-pangenome generator (text form).

Produces the three text inputs of the augment+call hot path from ``(seed, sizes)``:

* an *annotated* GFA whose tags are shaped the way the reference's graph annotator
  writes them (``EX:Z:T_R1.k,...`` on segments, ``JN:Z:T_R1.k.k+1,...`` on links that
  leave the haplotype path -- reference ``build/annotate.cpp:163-190,236-253`` and
  ``docs/README.md:5-37``),
* a GTF with ``gene`` / ``mRNA`` / ``exon`` rows (the columns the reference reads at
  ``scripts/call.py:188-216``),
* a GAF with the 12 mandatory columns and ``cs:Z:`` / ``dv:f:`` tags read at
  ``scripts/alignments_augmentation_from_gaf.py:142-180``.

Nothing here is derived from reference code; it is the recipe of SURVEY.md section 8-d.
Node ids are fixed-width decimals (string order == numeric order).

The graph model: one linear backbone of nodes in topological id order with SNP
bubbles (ref/alt single-base nodes); genes are disjoint backbone windows; a gene has
exons (runs of backbone positions) and introns; per gene up to six transcript shapes
(full, exon-skip, alt-donor, alt-acceptor, intron-retaining, a second skip), each also
as an ``_H1`` haplotype copy that takes the alt allele of every bubble.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field


@dataclass
class Transcript:
    name: str                 # e.g. G3T1
    gene: str
    exons: list               # list of (first_slot, last_slot) inclusive backbone slots
    haps: tuple = ("R1", "H1")


@dataclass
class Gene:
    name: str
    chrom: str
    strand: str
    exons: list               # canonical exon slots [(a, b), ...]
    transcripts: list = field(default_factory=list)


@dataclass
class Model:
    base_id: int
    slots: list               # per backbone slot: ("n", id) or ("b", ref_id, alt_id)
    node_len: dict            # id -> int
    node_seq: dict            # id -> str
    links: list               # [(u, v)] in emission order
    ex_tags: dict             # id -> [tag, ...]
    jn_tags: dict             # (u, v) -> [tag, ...]
    genes: list
    chrom: str = "chrS"

    def slot_nodes(self, slot: int, hap: str) -> int:
        s = self.slots[slot]
        if s[0] == "n":
            return s[1]
        return s[1] if hap == "R1" else s[2]

    def tx_path(self, tx: Transcript, hap: str) -> list:
        """Node path of a haplotype-aware transcript and the exon number of every node."""
        path, exno = [], []
        for k, (a, b) in enumerate(tx.exons, start=1):
            for s in range(a, b + 1):
                path.append(self.slot_nodes(s, hap))
                exno.append(k)
        return path, exno


_ACGT = "ACGT"


def make_model(seed: int = 1, n_genes: int = 20, base_id: int = 1_000_000,
               bubble_rate: float = 0.06, max_exons: int = 9,
               intergenic: tuple = (4, 12)) -> Model:
    rng = random.Random(seed)
    slots: list = []
    node_len: dict = {}
    node_seq: dict = {}
    next_id = base_id

    def new_node(length: int) -> int:
        nonlocal next_id
        nid = next_id
        next_id += 1
        node_len[nid] = length
        node_seq[nid] = "".join(rng.choice(_ACGT) for _ in range(length))
        return nid

    def add_plain() -> int:
        ln = 32 if rng.random() < 0.8 else rng.randint(1, 31)
        slots.append(("n", new_node(ln)))
        return len(slots) - 1

    def add_slots(n: int, allow_bubbles: bool = True) -> tuple:
        a = len(slots)
        i = 0
        while i < n:
            # a bubble needs a plain node on both sides inside the run
            if allow_bubbles and 0 < i < n - 1 and slots[-1][0] == "n" and rng.random() < bubble_rate:
                r = new_node(1)
                al = new_node(1)
                slots.append(("b", r, al))
            else:
                add_plain()
            i += 1
        return a, len(slots) - 1

    genes: list = []
    add_slots(rng.randint(*intergenic), allow_bubbles=False)
    for g in range(n_genes):
        gname = f"G{g + 1}"
        strand = "+" if g % 2 == 0 else "-"
        n_ex = min(max_exons, max(3, int(rng.expovariate(1 / 5.0)) + 3))
        exons = []
        for e in range(n_ex):
            ex_len = rng.randint(2, 6) if rng.random() < 0.85 else rng.randint(7, 12)
            a, b = add_slots(ex_len)
            # exon boundaries must be plain nodes (junction endpoints are unambiguous)
            exons.append((a, b))
            if e != n_ex - 1:
                in_len = rng.randint(3, 9) if rng.random() < 0.7 else rng.randint(10, 40)
                add_slots(in_len)
        gene = Gene(gname, "chrS", strand, exons)
        # --- transcript shapes -------------------------------------------------
        t = 1
        full = Transcript(f"{gname}T{t}", gname, list(exons)); t += 1
        gene.transcripts.append(full)
        # exon skip (drop one internal exon)
        if n_ex >= 3:
            k = rng.randint(1, n_ex - 2)
            gene.transcripts.append(Transcript(f"{gname}T{t}", gname,
                                               [x for i, x in enumerate(exons) if i != k])); t += 1
        # alt donor: exon k ends early (needs >= 3 slots)
        cand = [i for i, (a, b) in enumerate(exons[:-1]) if b - a >= 2]
        if cand and rng.random() < 0.8:
            k = rng.choice(cand)
            a, b = exons[k]
            cut = rng.randint(a, b - 1)
            while slots[cut][0] != "n":
                cut -= 1
            if cut >= a:
                ex2 = list(exons)
                ex2[k] = (a, cut)
                gene.transcripts.append(Transcript(f"{gname}T{t}", gname, ex2)); t += 1
        # alt acceptor: exon k starts late
        cand = [i for i, (a, b) in enumerate(exons) if i > 0 and b - a >= 2]
        if cand and rng.random() < 0.8:
            k = rng.choice(cand)
            a, b = exons[k]
            cut = rng.randint(a + 1, b)
            while slots[cut][0] != "n":
                cut += 1
            if cut <= b:
                ex2 = list(exons)
                ex2[k] = (cut, b)
                gene.transcripts.append(Transcript(f"{gname}T{t}", gname, ex2)); t += 1
        # intron retention: exons k and k+1 fused with the intron between them
        if n_ex >= 2 and rng.random() < 0.8:
            k = rng.randint(0, n_ex - 2)
            ex2 = list(exons[:k]) + [(exons[k][0], exons[k + 1][1])] + list(exons[k + 2:])
            gene.transcripts.append(Transcript(f"{gname}T{t}", gname, ex2)); t += 1
        # a second, double exon skip
        if n_ex >= 5 and rng.random() < 0.5:
            k = rng.randint(1, n_ex - 3)
            gene.transcripts.append(Transcript(f"{gname}T{t}", gname,
                                               [x for i, x in enumerate(exons) if i not in (k, k + 1)])); t += 1
        genes.append(gene)
        add_slots(rng.randint(*intergenic), allow_bubbles=False)

    # ---- links: backbone edges in topological order --------------------------------
    links: list = []
    seen = set()

    def add_link(u: int, v: int) -> None:
        if (u, v) not in seen:
            seen.add((u, v))
            links.append((u, v))

    def outs(slot):
        s = slots[slot]
        return [s[1]] if s[0] == "n" else [s[1], s[2]]

    for i in range(len(slots) - 1):
        for u in outs(i):
            for v in outs(i + 1):
                add_link(u, v)

    model = Model(base_id, slots, node_len, node_seq, links, {}, {}, genes)
    # ---- annotation (EX on nodes, JN on non-backbone hops) ---------------------------
    backbone = set(links)
    for gene in genes:
        for tx in gene.transcripts:
            for hap in tx.haps:
                tag = f"{tx.name}_{hap}"
                path, exno = model.tx_path(tx, hap)
                for n, k in zip(path, exno):
                    model.ex_tags.setdefault(n, []).append(f"{tag}.{k}")
                for (u, ku), (v, kv) in zip(zip(path, exno), zip(path[1:], exno[1:])):
                    if ku != kv:
                        if (u, v) not in seen:
                            seen.add((u, v))
                            links.append((u, v))
                        model.jn_tags.setdefault((u, v), []).append(f"{tag}.{ku}.{kv}")
    # annotate.cpp emits links grouped by source node in id order; do the same
    order = {n: i for i, n in enumerate(sorted(node_len))}
    first_seen = {l: i for i, l in enumerate(links)}
    links.sort(key=lambda l: (order[l[0]], first_seen[l]))
    _ = backbone
    return model


def write_gfa(model: Model, path: str, with_path: bool = True) -> None:
    with open(path, "w") as f:
        f.write("H\tVN:Z:1.1\n")
        for nid in sorted(model.node_len):
            line = f"S\t{nid}\t{model.node_seq[nid]}"
            if nid in model.ex_tags:
                line += "\tEX:Z:" + ",".join(model.ex_tags[nid])
            f.write(line + "\n")
        for (u, v) in model.links:
            line = f"L\t{u}\t+\t{v}\t+\t*"
            if (u, v) in model.jn_tags:
                line += "\tJN:Z:" + ",".join(model.jn_tags[(u, v)])
            f.write(line + "\n")
        if with_path:
            ref = [model.slot_nodes(s, "R1") for s in range(len(model.slots))]
            f.write(f"P\t{model.chrom}\t" + ",".join(f"{n}+" for n in ref) + "\t*\n")


def write_gtf(model: Model, path: str) -> None:
    # coordinates are cumulative reference-allele lengths (only cols 1,3,7,9 are read by call)
    pos = []
    acc = 1
    for s in range(len(model.slots)):
        n = model.slot_nodes(s, "R1")
        pos.append((acc, acc + model.node_len[n] - 1))
        acc += model.node_len[n]
    with open(path, "w") as f:
        f.write("#!synthetic annotation\n")
        for g in model.genes:
            a = pos[g.exons[0][0]][0]
            b = pos[g.exons[-1][1]][1]
            f.write(f'{g.chrom}\tsynth\tgene\t{a}\t{b}\t.\t{g.strand}\t.\tgene_id "{g.name}";\n')
            for tx in g.transcripts:
                a = pos[tx.exons[0][0]][0]
                b = pos[tx.exons[-1][1]][1]
                f.write(f'{g.chrom}\tsynth\tmRNA\t{a}\t{b}\t.\t{g.strand}\t.\t'
                        f'gene_id "{g.name}"; transcript_id "{tx.name}";\n')
                for (ea, eb) in tx.exons:
                    f.write(f'{g.chrom}\tsynth\texon\t{pos[ea][0]}\t{pos[eb][1]}\t.\t{g.strand}\t.\t'
                            f'gene_id "{g.name}"; transcript_id "{tx.name}";\n')


def _cs_for_span(rng: random.Random, span: int, p_sub: float, p_indel: float) -> str:
    """A cs string whose reference-consuming ops (':' '*' '-') sum to exactly ``span``."""
    r = rng.random()
    if r >= p_sub + p_indel or span < 12:
        return f":{span}"
    bases = "acgt"
    if r < p_sub:
        nsub = 1 if rng.random() < 0.7 else 2
        cuts = sorted(rng.sample(range(1, span - 1), nsub))
        out, prev = "", 0
        for c in cuts:
            if c - prev > 0:
                out += f":{c - prev}"
            a = rng.choice(bases)
            b = rng.choice([x for x in bases if x != a])
            out += f"*{a}{b}"
            prev = c + 1
        if span - prev > 0:
            out += f":{span - prev}"
        return out
    # small insertion or deletion
    c = rng.randint(3, span - 8)
    n = rng.randint(1, 4)
    ins = "".join(rng.choice(bases) for _ in range(n))
    if rng.random() < 0.5:
        return f":{c}+{ins}:{span - c}"
    return f":{c}-{ins}:{span - c - n}"


def make_gaf_lines(model: Model, seed: int = 1, n_lines: int = 500, read_len: int = 100,
                   p_lowmapq: float = 0.05, p_unmapped: float = 0.01, p_sub: float = 0.30,
                   p_indel: float = 0.05, p_highdv: float = 0.01, p_novel_skip: float = 0.04,
                   p_novel_ce: float = 0.02, p_intron: float = 0.06, p_novel_ss: float = 0.04,
                   p_clip: float = 0.03, zipf_a: float = 1.2):
    """Yield GAF text lines (without newline)."""
    rng = random.Random(seed * 7919 + 13)
    genes = model.genes
    weights = [1.0 / (i + 1) ** zipf_a for i in range(len(genes))]
    perm = list(range(len(genes)))
    rng.shuffle(perm)
    all_links = set(model.links)
    slot_of = {}
    for i, s in enumerate(model.slots):
        for n in s[1:]:
            slot_of[n] = i

    def emit(name, path, start, span, rev, mapq, cs, dv):
        plen = sum(model.node_len[n] for n in path)
        if rev:
            pstr = "".join(f"<{n}" for n in reversed(path))
            start = plen - (start + span)
        else:
            pstr = "".join(f">{n}" for n in path)
        end = start + span
        return (f"{name}\t{read_len}\t0\t{read_len}\t+\t{pstr}\t{plen}\t{start}\t{end}\t"
                f"{span}\t{span}\t{mapq}\tAS:i:{span}\tcs:Z:{cs}\tdv:f:{dv}")

    for li in range(n_lines):
        name = f"read{li // 2}/{li % 2 + 1}"
        if rng.random() < p_unmapped:
            yield f"{name}\t{read_len}\t0\t0\t+\t*\t0\t0\t0\t0\t0\t{rng.choice([0, 60])}"
            continue
        g = genes[perm[rng.choices(range(len(genes)), weights)[0]]]
        tx = rng.choice(g.transcripts)
        hap = rng.choice(tx.haps)
        path, _ = model.tx_path(tx, hap)
        r = rng.random()
        kind = "plain"
        acc = 0.0
        for k, p in (("skip", p_novel_skip), ("ce", p_novel_ce), ("intron", p_intron), ("ss", p_novel_ss)):
            acc += p
            if r < acc:
                kind = k
                break
        if kind == "skip" and len(tx.exons) >= 3:
            # novel exon skip: drop one internal exon of this transcript
            k = rng.randint(1, len(tx.exons) - 2)
            tx2 = Transcript(tx.name, tx.gene, [x for i, x in enumerate(tx.exons) if i != k])
            path, _ = model.tx_path(tx2, hap)
            jpos = sum(b - a + 1 for a, b in tx2.exons[:k])  # index of first node after the hop
        elif kind == "ce" and len(tx.exons) >= 2:
            # cassette: head -> 1-2 intron slots -> tail of an annotated junction
            k = rng.randint(0, len(tx.exons) - 2)
            ia, ib = tx.exons[k][1] + 1, tx.exons[k + 1][0] - 1
            if ib - ia >= 2:
                cs_a = rng.randint(ia + 1, ib - 1)
                cs_b = min(ib - 1, cs_a + rng.randint(0, 1))
                ex2 = list(tx.exons[:k + 1]) + [(cs_a, cs_b)] + list(tx.exons[k + 1:])
                tx2 = Transcript(tx.name, tx.gene, ex2)
                path, _ = model.tx_path(tx2, hap)
                jpos = sum(b - a + 1 for a, b in ex2[:k + 1])
            else:
                kind = "plain"
        elif kind == "intron" and len(tx.exons) >= 2:
            # read through an intron along the reference path
            k = rng.randint(0, len(tx.exons) - 2)
            ex2 = list(tx.exons[:k]) + [(tx.exons[k][0], tx.exons[k + 1][1])] + list(tx.exons[k + 2:])
            tx2 = Transcript(tx.name, tx.gene, ex2)
            path, _ = model.tx_path(tx2, hap)
            jpos = sum(b - a + 1 for a, b in tx.exons[:k + 1])
        elif kind == "ss" and len(tx.exons) >= 2:
            # novel splice site: shift one exon boundary by 1-3 slots (into exon or intron)
            k = rng.randint(0, len(tx.exons) - 2)
            ex2 = list(tx.exons)
            d = rng.choice([-3, -2, -1, 1, 2, 3])
            if rng.random() < 0.5:
                a, b = ex2[k]
                nb = b + d
                if a <= nb < tx.exons[k + 1][0] - 1:
                    ex2[k] = (a, nb)
            else:
                a, b = ex2[k + 1]
                na = a + d
                if tx.exons[k][1] + 1 < na <= b:
                    ex2[k + 1] = (na, b)
            tx2 = Transcript(tx.name, tx.gene, ex2)
            path, _ = model.tx_path(tx2, hap)
            jpos = sum(b - a + 1 for a, b in ex2[:k + 1])
        else:
            kind = "plain"
        lens = [model.node_len[n] for n in path]
        total = sum(lens)
        span = min(read_len, total)
        if kind == "plain":
            off = rng.randint(0, total - span)
        else:
            # centre the window on the hop so that it is actually traversed
            before = sum(lens[:jpos])
            lo = max(0, before - span + 5)
            hi = min(total - span, max(0, before - 5))
            off = min(max(0, total - span), rng.randint(min(lo, hi), max(lo, hi)))
        # window -> node sub-path
        acc = 0
        i0 = 0
        while acc + lens[i0] <= off:
            acc += lens[i0]
            i0 += 1
        start = off - acc
        i1 = i0
        acc2 = acc
        while acc2 + lens[i1] < off + span:
            acc2 += lens[i1]
            i1 += 1
        sub = path[i0:i1 + 1]
        rev = rng.random() < 0.5
        mapq = rng.randint(0, 19) if rng.random() < p_lowmapq else 60
        cs = _cs_for_span(rng, span, p_sub, p_indel)
        if rng.random() < p_clip and span > 20:
            n = rng.randint(1, 6)
            ins = "".join(rng.choice("acgt") for _ in range(n))
            cs = f"+{ins}:{span}" if rng.random() < 0.5 else f":{span}+{ins}"
        dv = "0.2" if rng.random() < p_highdv else rng.choice(["0", "0.01", "0.03", "0.005"])
        _ = all_links, slot_of
        yield emit(name, sub, start, span, rev, mapq, cs, dv)


def write_gaf(model: Model, path: str, **kw) -> int:
    n = 0
    with open(path, "w") as f:
        for line in make_gaf_lines(model, **kw):
            f.write(line + "\n")
            n += 1
    return n


def write_all(prefix: str, seed: int = 1, n_genes: int = 20, n_lines: int = 500, **kw) -> Model:
    model_kw = {k: kw.pop(k) for k in list(kw) if k in ("base_id", "bubble_rate", "max_exons", "intergenic")}
    read_seed = kw.pop("read_seed", seed)     # other reads over the same graph (replicates)
    m = make_model(seed=seed, n_genes=n_genes, **model_kw)
    write_gfa(m, prefix + ".gfa")
    write_gtf(m, prefix + ".gtf")
    write_gaf(m, prefix + ".gaf", seed=read_seed, n_lines=n_lines, **kw)
    return m


if __name__ == "__main__":
    import sys
    pre = sys.argv[1]
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    ng = int(sys.argv[3]) if len(sys.argv) > 3 else 20
    nl = int(sys.argv[4]) if len(sys.argv) > 4 else 500
    write_all(pre, seed, ng, nl)
