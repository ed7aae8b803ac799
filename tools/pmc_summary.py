#!/usr/bin/env python3
"""Per-kernel average of a rocprofv3 --pmc pass stored as rocpd sqlite (values of FETCH_SIZE / WRITE_SIZE are KB).

    python tools/pmc_summary.py gpurun_out/prof_x_pmc_fetch/pmc_results.db
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute("select name, counter_name, count(*), avg(counter_value), min(counter_value), max(counter_value), avg(duration) "
                      "from pmc_events group by name, counter_name order by 4 desc").fetchall()
    print(f"{'kernel':<62} {'counter':<12} {'calls':>6} {'avg':>14} {'min':>14} {'max':>14} {'avg_dur_us':>11}")
    for name, cname, n, avg, mn, mx, dur in rows:
        short = name if len(name) <= 62 else name[:59] + "..."
        print(f"{short:<62} {cname:<12} {n:>6} {avg:>14.1f} {mn:>14.1f} {mx:>14.1f} {dur / 1e3:>11.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
