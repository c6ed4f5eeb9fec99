#!/usr/bin/env python3
"""Print the kernel timeline (start/end in us relative to the first aisk kernel) of a rocprofv3 rocpd database."""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, start, end, stream_id, queue_id from kernels where name like '%aisk%' order by start").fetchall()
t0 = rows[0][1]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
n = int(sys.argv[3]) if len(sys.argv) > 3 else 60
for r in rows[skip:skip + n]:
    nm = r[0].split("aisk::")[1].split("(")[0][:22]
    print("%-22s q%-3s %10.1f -> %10.1f  (%7.1f us)" % (nm, r[4], (r[1] - t0) / 1e3, (r[2] - t0) / 1e3, (r[2] - r[1]) / 1e3))
