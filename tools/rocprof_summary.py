#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd sqlite (.db) kernel trace: per-kernel calls, total, average.
Usage: python tools/rocprof_summary.py gpurun_out/prof/x_results.db [top_n] > profiles/rNN_x.txt"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by sum(end-start) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}")
    print(f"# total kernel time {total/1e6:.3f} ms over {sum(r[1] for r in rows)} dispatches")
    print(f"{'calls':>8} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9} {'pct':>6}  kernel")
    for n, c, s, a, mn, mx in rows[:top]:
        print(f"{c:8d} {s/1e3:12.1f} {a/1e3:10.2f} {mn/1e3:9.2f} {mx/1e3:9.2f} {100*s/total:6.2f}  {n[:150]}")


if __name__ == "__main__":
    main()
