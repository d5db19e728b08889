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
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                         r.get("Stream_Id", r.get("Queue_Id", "0"))))
    t_end = max(r[1] for r in rows)
    t0 = t_end - int(window_ms * 1e6)
    # OGC_BENCH_MARK=1 makes bench.py launch a marker kernel at the start of every timed step: with markers the window
    # is exactly the steps between the first and the last one (under the tracer a step is slower than the un-traced
    # ms_per_step, so a window guessed from that covers fewer steps than assumed)
    marks = sorted(r[0] for r in rows if "spin" in r[2].lower() or "sleep" in r[2].lower())
    if len(marks) >= 2:
        t0, t_end = marks[0], marks[-1]
        n_steps = len(marks) - 1
        window_ms = (t_end - t0) / 1e6
        rows = [r for r in rows if r[0] < t_end and "spin" not in r[2].lower() and "sleep" not in r[2].lower()]
        print("markers: %d -> window = %d whole steps, %.1f ms (%.2f ms per step UNDER THE TRACER)" %
              (len(marks), n_steps, window_ms, window_ms / n_steps))
    agg = defaultdict(lambda: [0, 0])
    busy = 0
    for s, e, name, _q in rows:
        if s >= t0:
            a = agg[name]
            a[0] += 1
            a[1] += e - s
            busy += e - s
    print("window: last %.1f ms (%d steps); kernels busy %.1f ms (%.1f%% of window)" %
          (window_ms, n_steps, busy / 1e6, 100.0 * busy / (window_ms * 1e6)))
    # union of busy intervals over all queues, and per queue: how much of the window has NO kernel running
    def union(iv):
        iv = sorted(iv)
        tot, cs, ce = 0, None, None
        for s, e in iv:
            if cs is None or s > ce:
                if cs is not None:
                    tot += ce - cs
                cs, ce = s, e
            else:
                ce = max(ce, e)
        return tot + (ce - cs if cs is not None else 0)
    win = [r for r in rows if r[0] >= t0]
    print("union of kernel intervals (any queue): %.1f ms -> GPU idle %.1f%% of the window" %
          (union([(s, e) for s, e, _, _ in win]) / 1e6, 100.0 - 100.0 * union([(s, e) for s, e, _, _ in win]) / (window_ms * 1e6)))
    byq = defaultdict(list)
    for s, e, _, q in win:
        byq[q].append((s, e))
    for q, iv in sorted(byq.items(), key=lambda kv: -sum(e - s for s, e in kv[1])):
        print("  queue/stream %-6s kernels %6d  busy %8.1f ms" % (q, len(iv), union(iv) / 1e6))
    print("%-100s %8s %12s %12s %7s %12s" % ("kernel", "calls", "total_ms", "avg_us", "%busy", "us_per_step"))
    for name, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
        print("%-100s %8d %12.3f %12.1f %6.1f%% %12.1f" % (name[:100], c, t / 1e6, t / c / 1e3, 100.0 * t / busy,
                                                        t / 1e3 / n_steps))


if __name__ == "__main__":
    main()
