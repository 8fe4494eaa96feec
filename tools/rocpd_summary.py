#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) into a per-kernel stats table
(the equivalent of `--stats`' kernel_stats.csv).  usage: rocpd_summary.py results.db [out.csv]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute(
        "select name, count(*), sum(end - start), avg(end - start), min(end - start), max(end - start) "
        "from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows) or 1
    lines = ["Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage"]
    for n, c, s, a, mn, mx in rows:
        lines.append(f'"{n}",{c},{s},{a:.1f},{mn},{mx},{100.0 * s / tot:.2f}')
    out = "\n".join(lines) + "\n"
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(out)
    else:
        sys.stdout.write(out)


if __name__ == "__main__":
    main()
