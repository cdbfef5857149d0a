#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace --stats run (rocpd sqlite .db) as a plain-text per-kernel table.
usage: rocprof_summary.py <results.db> [more.db ...]   (durations in microseconds)"""
import sqlite3
import sys


def main():
    for path in sys.argv[1:]:
        db = sqlite3.connect(path)
        print("# %s" % path)
        print("%-78s %7s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
        rows = db.execute("select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc").fetchall()
        scale = 1.0
        # rocpd stores ns in rocpd_kernel_dispatch; the top_kernels view is already aggregated -- detect its unit from the raw table
        try:
            raw = db.execute("select sum(end-start) from rocpd_kernel_dispatch").fetchone()[0]
            tot = sum(r[2] for r in rows)
            if tot and raw:
                scale = (raw / 1e3) / tot            # -> microseconds
        except Exception:
            pass
        for name, calls, total, avg, pct in rows:
            print("%-78s %7d %14.1f %12.2f %6.1f%%" % (name[:78], calls, total * scale, avg * scale, pct))
        print()


if __name__ == "__main__":
    main()
