"""A hand-built augmented GFA in which the intronic splice-site search (reference scripts/call.py:728-835)
has to hold more nodes than the engine's in-register frontier: a novel link lands on an intron node that
fans out over `fan` parallel alleles before the annotated acceptor."""


def write_hub_case(prefix: str, fan: int = 40):
    ids = iter(range(10000, 99999))
    e1a, e1b = next(ids), next(ids)          # exon 1
    w = next(ids)                            # intron node the novel link lands on
    alleles = [next(ids) for _ in range(fan)]
    join = next(ids)                         # intron node after the fan
    e2a, e2b = next(ids), next(ids)          # exon 2
    pre = next(ids)                          # intron node before w (keeps w off the exon)
    lines = ["H\tVN:Z:1.1"]

    def seg(n, ex=None, nc=5):
        lines.append(f"S\t{n}\t{'A' * 20}" + (f"\tEX:Z:{ex}" if ex else "") + f"\tNC:i:{nc}")

    seg(e1a, "HT1_R1.1"); seg(e1b, "HT1_R1.1"); seg(pre); seg(w)
    for a in alleles:
        seg(a)
    seg(join); seg(e2a, "HT1_R1.2"); seg(e2b, "HT1_R1.2")

    def link(u, v, extra=""):
        lines.append(f"L\t{u}\t+\t{v}\t+\t*{extra}")

    link(e1a, e1b, "\tRC:i:9")
    link(e1b, pre, "\tRC:i:1")
    link(e1b, e2a, "\tJN:Z:HT1_R1.1.2\tRC:i:7")     # the annotated junction
    link(pre, w, "\tRC:i:1")
    for a in alleles:
        link(w, a, "\tRC:i:1")
    for a in alleles:
        link(a, join, "\tRC:i:1")
    link(join, e2a, "\tRC:i:4")
    link(e2a, e2b, "\tRC:i:9")
    # novel links (as augment appends them): exon 1 -> w (intronic acceptor) and join -> ... nothing else
    link(e1b, w, "\tRC:i:6\tID:Z:N")
    open(prefix + ".gfa", "w").write("\n".join(lines) + "\n")
    open(prefix + ".gtf", "w").write('chrH\tX\tgene\t1\t900\t.\t+\t.\tgene_id "HG";\n'
                                      'chrH\tX\tmRNA\t1\t900\t.\t+\t.\tgene_id "HG"; transcript_id "HT1";\n')
