#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace and/or PMC) as text.

    python tools/rocpd_summary.py gpurun_out/prof/xyz_results.db > profiles/r01_xyz.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    print("# rocprofv3 summary of %s" % path)
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    rows = cur.execute("select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       "from kernels group by name order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("\n## kernel-trace stats (ns)\n")
    print("%-72s %7s %14s %14s %14s %14s %7s" % ("kernel", "calls", "total_ns", "avg_ns", "min_ns", "max_ns", "pct"))
    for name, n, tot, avg, mn, mx in rows:
        print("%-72s %7d %14d %14.0f %14d %14d %6.2f%%" % (name[:72], n, tot, avg, mn, mx, 100.0 * tot / total))
    # resources of the named kernels
    want = [c for c in ("name", "vgpr_count", "accum_vgpr_count", "sgpr_count", "lds_size", "scratch_size", "workgroup_size", "grid_size") if c in cols]
    if len(want) > 1:
        print("\n## dispatch resources (first dispatch per kernel)\n")
        seen = set()
        for row in cur.execute("select %s from kernels" % ", ".join(want)):
            if row[0] in seen:
                continue
            seen.add(row[0])
            print("  " + ", ".join("%s=%s" % (k, v) for k, v in zip(want, row)))
    try:
        pmc = cur.execute("select k.name, p.counter_name, count(*), sum(p.value), avg(p.value) from pmc_events p "
                          "join kernels k on k.id = p.event_id group by k.name, p.counter_name").fetchall()
    except sqlite3.Error:
        pmc = []
    if not pmc:
        try:
            pmc = cur.execute("select kernel_name, counter_name, count(*), sum(value), avg(value) from counters_collection "
                              "group by kernel_name, counter_name").fetchall()
        except sqlite3.Error:
            pmc = []
    if pmc:
        print("\n## PMC counters (per dispatch average)\n")
        for name, cname, n, tot, avg in pmc:
            print("%-60s %-28s n=%-5d avg=%.6g sum=%.6g" % (name[:60], cname, n, avg, tot))


if __name__ == "__main__":
    main(sys.argv[1])
