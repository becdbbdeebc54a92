"""Random GAF lines on a small graph: cs strings with every op kind but '~' (whose handling in the reference
depends on a stale variable) -- zero lengths, '=', runs of '*',
leading/trailing indels), arbitrary start / end offsets, both strands, revisits.  Many of them make the
reference raise; the checker must raise on the same lines."""
import random

LENS = {3001: 7, 3002: 32, 3003: 1, 3004: 1, 3005: 12, 3006: 32, 3007: 3, 3008: 20, 3009: 1, 3010: 9}
LINKS = [(3001, 3002), (3002, 3003), (3002, 3004), (3003, 3005), (3004, 3005), (3005, 3006), (3006, 3007),
         (3007, 3008), (3008, 3009), (3009, 3010), (3002, 3005), (3005, 3008), (3006, 3010)]


def write_graph(path):
    with open(path, "w") as f:
        f.write("H\tVN:Z:1.1\n")
        for n, l in LENS.items():
            f.write(f"S\t{n}\t{'ACGT' * 8}"[:len(f"S\t{n}\t") + l] + "\n")
        for u, v in LINKS:
            f.write(f"L\t{u}\t+\t{v}\t+\t*\n")


def random_cs(rng):
    ops = []
    for _ in range(rng.choice([1, 1, 1, 2, 2, 3, 4, 6])):
        k = rng.random()
        if k < 0.45:
            ops.append(f":{rng.choice([0, 1, 2, 3, 5, 8, 13, 30, 60])}")
        elif k < 0.65:
            ops.append("*" + rng.choice("acgt") + rng.choice("acgt"))
        elif k < 0.78:
            ops.append("-" + "".join(rng.choice("acgt") for _ in range(rng.choice([0, 1, 2, 5, 12, 33]))))
        elif k < 0.9:
            ops.append("+" + "".join(rng.choice("acgt") for _ in range(rng.choice([0, 1, 2, 4]))))
        else:
            ops.append("=" + "".join(rng.choice("ACGT") for _ in range(rng.choice([1, 3, 9]))))
    return "".join(ops)


def random_line(rng, i):
    nodes = sorted(LENS)
    k = rng.choice([1, 1, 2, 2, 3, 3, 4, 5, 7])
    start_idx = rng.randrange(len(nodes))
    path = [nodes[start_idx]]
    adj = {}
    for u, v in LINKS:
        adj.setdefault(u, []).append(v)
    while len(path) < k:
        nxt = adj.get(path[-1])
        if not nxt or rng.random() < 0.08:
            path.append(rng.choice(nodes))          # a hop that is no link (novel) or a revisit
        else:
            path.append(rng.choice(nxt))
    rev = rng.random() < 0.5
    listed = list(reversed(path)) if rev else path
    plen = sum(LENS[n] for n in listed)
    start = rng.choice([0, 0, 1, 2, 5, LENS[listed[0]] - 1, LENS[listed[0]], LENS[listed[0]] + 2])
    end = rng.choice([plen, plen, plen - 1, plen - 3, plen - LENS[listed[-1]], plen - LENS[listed[-1]] - 1, start + 1])
    p = "".join(("<" if rev else ">") + str(n) for n in listed)
    mapq = rng.choice([60, 60, 60, 20, 19, 0])
    tags = [f"cs:Z:{random_cs(rng)}"] if rng.random() < 0.95 else []
    tags.append("dv:f:" + rng.choice(["0", "0.01", "0.1", "0.11", "1", "3e-05"]))
    if rng.random() < 0.03:
        tags = tags[:1]                                   # no dv tag
    return "\t".join([f"f{i}", "100", "0", "100", "+", p, str(plen), str(start), str(end), "50", "50", str(mapq)] + tags)
