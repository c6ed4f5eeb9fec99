#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (--kernel-trace --stats) into the plain-text per-kernel summary that is
committed under profiles/.  usage: rocprof_summary.py results.db [> profiles/xyz.txt]"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(grid_x), max(grid_y), max(workgroup_x), max(lds_size), max(vgpr_count), max(sgpr_count) "
        "from kernels group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-64s %6s %12s %12s %12s %12s %6s  %s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "%", "grid/wg/lds/vgpr/sgpr"))
    for r in rows:
        print("%-64s %6d %12.1f %12.2f %12.2f %12.2f %6.2f  %dx%d/%d/%d/%d/%d" % (
            r[0][:64], r[1], r[2] / 1e3, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / total,
            r[6], r[7], r[8], r[9], r[10], r[11]))
    try:
        pmc = c.execute("select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection "
                        "group by kernel_name, counter_name order by kernel_name, counter_name").fetchall()
        if pmc:
            print("\ncounters (value summed over all instances, per dispatch average):")
            for r in pmc:
                print("%-64s %-24s %20.1f" % (r[0][:64], r[1], r[2] / max(r[3], 1)))
    except sqlite3.Error:
        pass


if __name__ == "__main__":
    main(sys.argv[1])
