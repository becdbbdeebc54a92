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


_DV_MID = "100000000000000012490009027033011079765856266021728515625"


def mutate_text(rng, line):
    """Byte-level damage to a well-formed line: what the two tokenisers (host gaf.cpp, device gaf_device.h) must
    read alike -- odd whitespace, repeated / embedded tags, integer spellings, dv values around 0.1, odd paths."""
    cols = line.split("\t")
    k = rng.randrange(22)
    if k == 0:
        cols = cols[:rng.randrange(0, 13)]
    elif k == 1:
        cols[11] = rng.choice(["+60", "6_0", "60_", "_60", "6__0", "", "x", "-1", "20", "19", "0x14", "9" * 25, "\xd9\xa1"])
    elif k == 2:
        cols[rng.choice([6, 7, 8])] = rng.choice(["+5", "1_0", "-3", "abc", "5.0", "1e3", str(2 ** 31), str(2 ** 40), str(-2 ** 31 - 1), "9" * 20])
    elif k == 3:
        cols[5] = rng.choice(["*", "**", ">", "<", "3001", ">3001<3002", "<3002>3001", ">3001>3001>3002", ">03001", ">3001>>3002",
                              ">99999", ">3001>x", "<3010<3009<3009", ">" + "9" * 19, ">3001>", ">-1"])
    elif k == 4 and len(cols) > 12:
        cols[12] = rng.choice(["cs:Z:", "cs:Z:cs:Z::5", "cs:i:7", "xcs:Z::9", "cs:Z::5cs:Z:*ac", "cs::3", "cs:Z::0", "cs:Z:~ac3gt", "cs:Z::12:7",
                               "cs:Z:*ac*gt", "cs:Z::536870911", "cs:Z::536870912", "cs:Z:+ac:50", "cs:Z::50+ac", "cs:Z:+ac-gt", "cs:Z:-" + "a" * 40,
                               "cs:Z::5dv:f:0.9", "cs:Z:5:5", "cs:Z:=ACGT:3", "cs:Z::3*ac*ac*ac*ac*ac:2", "cs:Z:+:5", "ccs:cs:Z::4"])
    elif k == 5:
        i = rng.randrange(12, max(13, len(cols)))
        v = rng.choice(["0.1", "0.10", "0.09", "0.100000000000000005", "0." + _DV_MID, "0." + _DV_MID + "1", "0." + _DV_MID[:-1] + "4",
                        "0.1000000000000000125", "0.2", "1", "00.05", ".5", "5.", "0.1e5", "1e-3", "0.0999999999999999999999", "a", ""])
        tag = rng.choice(["dv:f:", "xdv:f:", "dv:f:dv:f:"]) + v
        if i < len(cols):
            cols[i] = tag
        else:
            cols.append(tag)
    elif k == 6:
        cols.insert(rng.randrange(12, len(cols) + 1), rng.choice(["NM:i:3", "dv:f:x", "tp:A:P", "cs:Z::1", "dv:f:0.5", "dv:f:0.0"]))
    elif k == 7:
        line2 = "\t".join(cols)
        pos = rng.randrange(len(line2) + 1)
        return line2[:pos] + rng.choice([" ", "  ", "\t\t", "\r", "\x0b", "\x0c", "\x1c", "\x1f", " \t "]) + line2[pos:]
    elif k == 8:
        return rng.choice([" ", "\t", ""]) + "\t".join(cols) + rng.choice([" ", "\r", "\t", ""])
    elif k == 9:
        return " ".join(cols)
    elif k == 10:
        return ""
    elif k == 11 and len(cols) > 12:
        rng.shuffle(cols[12:])
        tail = cols[12:]
        rng.shuffle(tail)
        cols = cols[:12] + tail
    elif k == 12:
        line2 = "\t".join(cols)
        pos = rng.randrange(len(line2) + 1)
        return line2[:pos] + chr(rng.choice([0, 1, 127, 200, 255, 0x85, 0xa0])) + line2[pos:]
    return "\t".join(cols)
