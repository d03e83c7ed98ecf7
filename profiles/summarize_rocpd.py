#!/usr/bin/env python3
"""Turns rocprofv3's rocpd sqlite output (ROCm 7.2 default format) into the small CSV summaries committed here.

    python profiles/summarize_rocpd.py stats  <results.db> <out.csv>      # --kernel-trace --stats run
    python profiles/summarize_rocpd.py pmc    <results.db> <out.csv>      # --pmc run: per-kernel mean counter values
"""
import csv
import sqlite3
import sys


def stats(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "calls", "total_us", "avg_us", "percent"])
        for n, c, t, a, p in rows:
            w.writerow([n, c, round(t, 3), round(a, 3), round(p, 3)])


def pmc(db, out):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration), "
                       "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_block_size), max(workgroup_size), max(grid_size) "
                       "from counters_collection group by kernel_name, counter_name").fetchall()
    with open(out, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["kernel", "counter", "dispatches", "mean", "min", "max", "mean_duration_ns_profiled", "vgpr", "agpr",
                    "sgpr", "lds_bytes", "workgroup", "grid"])
        for r in rows:
            w.writerow(r)


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2], sys.argv[3])
