"""Idle gaps on the busiest queue of a rocprofv3 --kernel-trace CSV, over the last `window_ms` (development tool).

    python tools/gaps.py <kernel_trace.csv> <window_ms>
Prints a histogram of gap lengths and the (previous kernel -> next kernel) pairs that own the most idle time.
"""
import csv
import re
import sys
from collections import defaultdict


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name[:70]


def main():
    path, window_ms = sys.argv[1], float(sys.argv[2])
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Stream_Id", r.get("Queue_Id", "0"))))
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(window_ms * 1e6)
    rows = sorted(r for r in rows if r[0] >= t0)
    count = defaultdict(int)
    for r in rows:
        count[r[3]] += 1
    main_q = max(count, key=count.get)
    seq = [r for r in rows if r[3] == main_q]
    other = [r for r in rows if r[3] != main_q]
    edges = [0, 2e3, 5e3, 10e3, 20e3, 50e3, 100e3, 1e12]
    hist = [[0, 0] for _ in edges[:-1]]
    pairs = defaultdict(lambda: [0, 0])
    covered = 0   # gap time during which some other queue had a kernel running
    oi = 0
    for a, b in zip(seq, seq[1:]):
        g = b[0] - a[1]
        if g <= 0:
            continue
        for i in range(len(edges) - 1):
            if edges[i] <= g < edges[i + 1]:
                hist[i][0] += 1
                hist[i][1] += g
        p = pairs[(short(a[2]), short(b[2]))]
        p[0] += 1
        p[1] += g
        for s, e, _, _ in other:
            lo, hi = max(s, a[1]), min(e, b[0])
            if hi > lo:
                covered += hi - lo
    total = sum(h[1] for h in hist)
    print("queue %s: %d kernels, idle between kernels %.1f ms of %.1f ms (other queues busy during %.1f ms of it, summed)"
          % (main_q, len(seq), total / 1e6, window_ms, covered / 1e6))
    for i, h in enumerate(hist):
        print("  gap %6.0f..%-8.0f us: %6d gaps %8.2f ms" % (edges[i] / 1e3, min(edges[i + 1], 1e9) / 1e3, h[0], h[1] / 1e6))
    print("%-72s %-72s %6s %9s" % ("after", "before", "n", "gap_ms"))
    for (a, b), (n, g) in sorted(pairs.items(), key=lambda kv: -kv[1][1])[:45]:
        print("%-72s %-72s %6d %9.3f" % (a, b, n, g / 1e6))


def context(path, window_ms, min_gap_us, width=6):
    """Kernels around every gap longer than `min_gap_us` on the busiest queue, and what the other queues ran meanwhile."""
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Stream_Id", r.get("Queue_Id", "0"))))
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(window_ms * 1e6)
    rows = sorted(r for r in rows if r[0] >= t0)
    count = defaultdict(int)
    for r in rows:
        count[r[3]] += 1
    main_q = max(count, key=count.get)
    seq = [r for r in rows if r[3] == main_q]
    other = [r for r in rows if r[3] != main_q]
    for i in range(len(seq) - 1):
        g = seq[i + 1][0] - seq[i][1]
        if g < min_gap_us * 1e3:
            continue
        print("---- gap %.1f us at t=%.3f ms" % (g / 1e3, (seq[i][1] - t0) / 1e6))
        for r in seq[max(0, i - width + 1):i + 1]:
            print("   before  %8.1f us  %s" % ((r[1] - r[0]) / 1e3, short(r[2])))
        for s, e, name, q in other:
            if e > seq[i][1] and s < seq[i + 1][0]:
                print("     [q%s] %8.3f..%8.3f ms  %s" % (q, (s - t0) / 1e6, (e - t0) / 1e6, short(name)))
        for r in seq[i + 1:i + 1 + width]:
            print("   after   %8.1f us  %s" % ((r[1] - r[0]) / 1e3, short(r[2])))


if __name__ == "__main__":
    if len(sys.argv) > 3:
        context(sys.argv[1], float(sys.argv[2]), float(sys.argv[3]))
        sys.exit(0)
    main()
