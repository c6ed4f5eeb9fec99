#!/bin/bash
# HBM-side traffic of EVERY kernel of a step (separate --pmc passes for FETCH_SIZE / WRITE_SIZE, serial mode so that kernels
# do not overlap), per launch, as a table.  FETCH_SIZE is doubled (gfx950 reports 1/2 for wide coalesced reads, see
# MI355X_MICROARCH.md); for kernels that read narrow pieces the doubled value is an upper bound.  usage: tools/pmc_traffic_all.sh [extra bench args]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_all_$c
  AISGPU_SERIAL=1 timeout 300 rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_all_$c -o p -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline "$@" > /dev/null 2>&1
done
python - <<PY
import sqlite3
def per_launch(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(distinct dispatch_id) from counters_collection where counter_name=? and kernel_name like '%aisk%' group by kernel_name", (counter,)).fetchall()
    return {r[0]: (r[1] / r[2], r[2]) for r in rows}
f = per_launch("/tmp/pmc_all_FETCH_SIZE/p_results.db", "FETCH_SIZE")
w = per_launch("/tmp/pmc_all_WRITE_SIZE/p_results.db", "WRITE_SIZE")
print("%-58s %8s %14s %14s" % ("kernel", "launches", "read MB (x2)", "written MB"))
steps = max(v[1] for k, v in f.items() if "k1_dpp" in k)   # one launch of the front end per step
tr = tw = 0.0
for k in sorted(f, key=lambda k: -(f[k][0] * 2 + w.get(k, (0, 0))[0])):
    rd, wr = f[k][0] * 2 * 1024 / 1e6, w.get(k, (0, 0))[0] * 1024 / 1e6
    print("%-58s %8d %14.1f %14.1f" % (k.split("aisk::")[1].split("(")[0][:58], f[k][1], rd, wr))
    # per step: a kernel's total over the run divided by the number of steps (the template instances of the derotation / FIR
    # kernel take turns: each has fewer launches than there are steps, together one per step)
    tr += rd * f[k][1] / steps; tw += wr * w.get(k, (0, f[k][1]))[1] / steps
print("per step (every kernel's bytes over the run / %d steps): read %.1f MB, written %.1f MB, together %.1f MB" % (steps, tr, tw, tr + tw))
PY
