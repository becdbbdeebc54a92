#!/usr/bin/env python3
"""Per-kernel durations out of a rocprofv3 results database (rocprofv3 --kernel-trace -d DIR -o NAME writes NAME_results.db when the
csv output is not asked for): name, calls, average and minimum in microseconds; --each lists every launch of the kernels that match."""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
each = sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--each" else None
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
if each:
    q = f"select s.kernel_name, d.start, d.end - d.start from {kd} d join {ks} s on d.kernel_id = s.id order by d.start"
    for name, _, dur in db.execute(q):
        if each in name:
            print(name[:60], round(dur / 1e3, 1))
else:
    q = f"select s.kernel_name, count(*), avg(d.end - d.start), min(d.end - d.start) from {kd} d join {ks} s on d.kernel_id = s.id group by s.kernel_name order by 3 desc"
    for name, calls, avg, mn in db.execute(q):
        if "augment" in name:
            print(name[:80], calls, round(avg / 1e3, 1), round(mn / 1e3, 1))
