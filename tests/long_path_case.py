"""Lines around the path-length bounds of the short-walk class (31 nodes; 29 with an indel) on a chain of tiny nodes
with SNP-like side nodes: ':' runs, '*' and one '+' / '-' at every position class (inside a node, on a node boundary,
spanning nodes, at the read ends), both strands.  Small enough for the oracle."""
import random

N = 60            # chain nodes 5001 .. 5060
LEN = {5001 + i: (3 if i % 5 else 1) for i in range(N)}
ALT = {5001 + i: 7001 + i for i in range(3, N - 3, 4)}      # node -> a 1-base alternative beside it


def write_graph(path):
    with open(path, "w") as f:
        f.write("H\tVN:Z:1.1\n")
        for n, l in LEN.items():
            f.write(f"S\t{n}\t{'ACGT'[:l] if l <= 4 else 'A' * l}\n")
        for n, a in ALT.items():
            f.write(f"S\t{a}\tT\n")
        for i in range(N - 1):
            u, v = 5001 + i, 5002 + i
            f.write(f"L\t{u}\t+\t{v}\t+\t*\n")
            if v in ALT:
                f.write(f"L\t{u}\t+\t{ALT[v]}\t+\t*\n")
                f.write(f"L\t{ALT[v]}\t+\t{v + 1}\t+\t*\n")


def random_line(rng, i):
    k = rng.choice([1, 2, 5, 27, 28, 29, 30, 31, 32, 33, 40])
    s = rng.randrange(0, N - k)
    path, lens = [], []
    for j in range(k):
        n = 5001 + s + j
        if n in ALT and 0 < j < k - 1 and rng.random() < 0.3:
            path.append(ALT[n]); lens.append(1)
        else:
            path.append(n); lens.append(LEN[n])
    plen = sum(lens)
    rev = rng.random() < 0.5
    first_len, last_len = (lens[-1], lens[0]) if rev else (lens[0], lens[-1])      # start / end refer to the path as listed
    start = rng.choice([0, 0, 1, first_len - 1]) if first_len > 1 else 0
    end = plen - (rng.choice([0, 0, 1]) if last_len > 1 else 0)
    span = end - start
    if span < 4:
        start, end, span = 0, plen, plen
    kind = rng.random()
    ops = []
    if kind < 0.35:
        ops = [(":", span)]
    elif kind < 0.55:                       # mismatches
        cut = sorted(rng.sample(range(span), min(span, rng.choice([1, 2, 3, 4, 5]))))
        prev = 0
        for c in cut:
            if c > prev: ops.append((":", c - prev))
            ops.append(("*", 1)); prev = c + 1
        if span > prev: ops.append((":", span - prev))
    else:                                   # one indel, maybe with mismatches around it
        ind = rng.choice("+-")
        m = rng.choice([1, 1, 2, 3, 5])
        body = span - (m if True else 0)
        if body < 2:
            ops = [(":", span)]
        else:
            c = rng.choice([0, 1, body // 2, body - 1, body]) if rng.random() < 0.4 else rng.randrange(0, body + 1)
            left, right = c, body - c
            for seg, tail in ((left, False), (right, True)):
                if tail: ops.append((ind, m))
                if seg:
                    if seg > 2 and rng.random() < 0.4:
                        q = rng.randrange(0, seg)
                        if q: ops.append((":", q))
                        ops.append(("*", 1))
                        if seg - q - 1: ops.append((":", seg - q - 1))
                    else:
                        ops.append((":", seg))
    if len(ops) == 2 and any(o == "+" for o, _ in ops):
        ops = [(":", span)]          # the reference clips a 2-op list with a '+' (:40-50) and would then run out of ops
    cs = "".join(f":{l}" if o == ":" else ("*ac" if o == "*" else o + "acgt"[:1] * l) for o, l in ops)
    listed = list(reversed(path)) if rev else path
    p = "".join(("<" if rev else ">") + str(n) for n in listed)
    return "\t".join([f"r{i}", "100", "0", "100", "+", p, str(plen), str(start), str(end), "50", "50", "60", f"cs:Z:{cs}", "dv:f:0.01"])


def write_case(prefix, seed, lines):
    rng = random.Random(seed)
    write_graph(prefix + ".gfa")
    with open(prefix + ".gaf", "w") as f:
        for i in range(lines):
            f.write(random_line(rng, i) + "\n")
