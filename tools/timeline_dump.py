#!/usr/bin/env python3
"""Every aisk kernel of a rocprofv3 rocpd database as one line: short name, start us, end us, duration us, queue (sorted by start).
usage: timeline_dump.py results.db > file.txt   (analysis happens off the GPU box)"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, queue_id from kernels where name like '%aisk%' order by start").fetchall()
t0 = rows[0][1]
for name, st, en, q in rows:
    nm = name.split("aisk::")[1].split("(")[0]
    print("%-40s %12.1f %12.1f %9.1f q%s" % (nm[:40], (st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, q))
