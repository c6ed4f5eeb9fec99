#!/usr/bin/env python3
"""Per-block kernel timeline of a rocprofv3 rocpd database: the k-th launch of every kernel belongs to block k.
usage: timeline2.py results.db [first_block] [n_blocks]"""
import sqlite3
import sys
from collections import defaultdict

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels where name like '%aisk%' order by start").fetchall()
t0 = rows[0][1]
cnt = defaultdict(int)
blocks = defaultdict(list)
for name, st, en, q in rows:
    nm = name.split("aisk::")[1].split("(")[0].split("<")[0]
    k = cnt[nm]
    cnt[nm] += 1
    blocks[k].append((nm, (st - t0) / 1e3, (en - t0) / 1e3, q))
b0 = int(sys.argv[2]) if len(sys.argv) > 2 else 10
nb = int(sys.argv[3]) if len(sys.argv) > 3 else 3
for b in range(b0, b0 + nb):
    print("block", b)
    for nm, st, en, q in blocks[b]:
        print("   %-18s q%-2s %9.1f -> %9.1f (%6.1f)" % (nm, q, st, en, en - st))
