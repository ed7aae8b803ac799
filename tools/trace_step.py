"""Timeline of one bench step from a rocprofv3 kernel trace (csv): kernel, duration, gap to the previous kernel.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python bench.py --steps 3 --warmup 1
    python tools/trace_step.py gpurun_out/trace
"""
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
    if not files:
        print("no kernel_trace.csv under", root)
        return 1
    rows = []
    with open(files[0]) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    rows.sort()
    # the last reg_loop_kernel of the trace ends a step; walk back to the previous one
    loops = [i for i, r in enumerate(rows) if "reg_loop_kernel" in r[2]]
    if len(loops) < 2:
        print("need two steps in the trace")
        return 1
    # pick two consecutive resident-loop launches that are one step apart (same kernels in between)
    best = None
    for a, b in zip(loops, loops[1:]):
        if 5 <= b - a <= 14:
            best = (a, b)
    if best is None:
        print("no step found")
        return 1
    a, b = best
    prev_end = rows[a][1]
    total_gap = 0
    print(f"{'kernel':60s} {'dur_us':>9s} {'gap_us':>8s}")
    for s, e, name in rows[a + 1:b + 1]:
        gap = (s - prev_end) / 1000.0
        total_gap += gap
        print(f"{name[:60]:60s} {(e - s) / 1000.0:9.2f} {gap:8.2f}")
        prev_end = e
    print(f"step = {(rows[b][1] - rows[a][1]) / 1000.0:.1f} us, of which gaps {total_gap:.1f} us")
    return 0


if __name__ == "__main__":
    sys.exit(main())
