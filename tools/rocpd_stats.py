#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) kernel trace: per-kernel calls, total, avg, share.
usage: python tools/rocpd_stats.py results.db [top_n]"""
import re
import sqlite3
import sys


def main():
    db = sys.argv[1]
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
    c = sqlite3.connect(db)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scol = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in scol else ("kernel_name" if "kernel_name" in scol else scol[-1])
    rows = c.execute(
        "select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id group by s.%s order by 3 desc"
        % (name_col, kd, ks, name_col)
    ).fetchall()
    total = sum(r[2] for r in rows)
    span = c.execute("select min(start), max(end) from %s" % kd).fetchone()
    print("# kernels: %d distinct, %d dispatches, busy %.3f ms, span %.3f ms" % (len(rows), sum(r[1] for r in rows), total / 1e6, (span[1] - span[0]) / 1e6))
    print("%-6s %9s %11s %10s %10s %10s  %s" % ("pct", "calls", "total_ms", "avg_us", "min_us", "max_us", "kernel"))
    for name, n, tot, mn, mx in rows[:top]:
        short = re.sub(r"\s+", " ", name)[:150]
        print("%5.1f%% %9d %11.3f %10.1f %10.1f %10.1f  %s" % (100.0 * tot / total, n, tot / 1e6, tot / n / 1e3, mn / 1e3, mx / 1e3, short))


if __name__ == "__main__":
    main()
