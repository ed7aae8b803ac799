#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (sqlite) result: per-kernel calls / total / average / min / max duration.

    python tools/rocpd_stats.py gpurun_out/prof_xxx/name_results.db > profiles/rNN_kernel_stats.txt
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = "name" if "name" in cols else "kernel_name"
    rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                       f"from kernels group by {name_col} order by 3 desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print(f"{'kernel':<70} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>10} {'max_us':>10} {'pct':>6}")
    for name, calls, tot, avg, mn, mx in rows:
        short = name if len(name) <= 70 else name[:67] + "..."
        print(f"{short:<70} {calls:>7} {tot / 1e3:>12.1f} {avg / 1e3:>10.2f} {mn / 1e3:>10.2f} {mx / 1e3:>10.2f} {100.0 * tot / total:>6.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
