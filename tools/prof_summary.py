"""Summarise a rocprofv3 --kernel-trace CSV over the LAST `window_ms` of the run (the timed steps of bench.py),
so MIOpen's first-call search kernels and warm-up launches do not pollute the per-kernel table.

    python tools/prof_summary.py <kernel_trace.csv> <window_ms> [n_steps] > profiles/rNN_steady.txt
"""
import csv
import sys
from collections import defaultdict


def main():
    path, window_ms = sys.argv[1], float(sys.argv[2])
    n_steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(window_ms * 1e6)
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, name in rows:
        if s >= t0:
            a = agg[name]
            a[0] += 1
            a[1] += e - s
            busy += e - s
    print("window: last %.1f ms (%d steps); kernels busy %.1f ms (%.1f%% of window)" %
          (window_ms, n_steps, busy / 1e6, 100.0 * busy / (window_ms * 1e6)))
    print("%-100s %8s %12s %12s %7s" % ("kernel", "calls", "total_ms", "avg_us", "%busy"))
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%-100s %8d %12.3f %12.1f %6.1f%%" % (name[:100], c, t / 1e6, t / c / 1e3, 100.0 * t / busy))


if __name__ == "__main__":
    main()
