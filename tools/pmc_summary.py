#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc pass stored as rocpd sqlite: counter values SUMMED over the hardware
instances of one dispatch (XCDs / shader engines report separate rows), then averaged over the dispatches of a kernel.
FETCH_SIZE / WRITE_SIZE are in KB.

    python tools/pmc_summary.py gpurun_out/prof_x_pmc_fetch/pmc_results.db
"""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, counter_name, count(*), avg(v), min(v), max(v), avg(d) from "
        "(select name, counter_name, dispatch_id, sum(counter_value) as v, avg(duration) as d from pmc_events "
        " group by name, counter_name, dispatch_id) group by name, counter_name order by name, counter_name").fetchall()
    print(f"{'kernel':<62} {'counter':<22} {'launches':>8} {'avg_per_launch':>16} {'min':>14} {'max':>14} {'avg_dur_us':>11}")
    for name, cname, n, avg, mn, mx, dur in rows:
        short = name if len(name) <= 62 else name[:59] + "..."
        print(f"{short:<62} {cname:<22} {n:>8} {avg:>16.1f} {mn:>14.1f} {mx:>14.1f} {dur / 1e3:>11.1f}")


if __name__ == "__main__":
    main(sys.argv[1])
