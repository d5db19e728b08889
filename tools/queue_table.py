"""Per-queue kernel table of a rocprofv3 --kernel-trace CSV over the last `window_ms` (development tool):
which kernels run on the side queues, and for how long.    python tools/queue_table.py <kernel_trace.csv> <window_ms> [steps]"""
import csv, re, sys
from collections import defaultdict

path, window_ms = sys.argv[1], float(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
t0 = max(r[1] for r in rows) - int(window_ms * 1e6)
byq = defaultdict(lambda: defaultdict(lambda: [0, 0]))
for s, e, name, q in rows:
    if s >= t0:
        a = byq[q][re.sub(r"\(anonymous namespace\)::|^void ", "", name)[:90]]
        a[0] += 1; a[1] += e - s
for q, ks in sorted(byq.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
    tot = sum(v[1] for v in ks.values())
    print("queue %s: %d kernels, %.2f ms per step" % (q, sum(v[0] for v in ks.values()) // steps, tot / 1e6 / steps))
    for name, (c, t) in sorted(ks.items(), key=lambda kv: -kv[1][1])[:14]:
        print("   %6.1f us/step  %5.1f calls/step  %s" % (t / 1e3 / steps, c / steps, name))
