"""Kernel time per phase of the training step from a rocprofv3 kernel trace of `MARK=1 python tools/step_timeline.py`
(the phases are separated by torch.cuda._sleep marker kernels on the main queue).  Development tool.
    python tools/phase_busy.py <kernel_trace.csv>"""
import csv, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", r.get("Queue_Id", "0"))))
rows.sort()
cnt = defaultdict(int)
for r in rows:
    cnt[r[3]] += 1
mq = max(cnt, key=cnt.get)
main = [r for r in rows if r[3] == mq]
marks = [i for i, r in enumerate(main) if "spin" in r[2].lower() or "sleep" in r[2].lower()]
print("markers found:", len(marks))
names = ["forward", "loss", "backward", "(side work)", "optimizer", "between steps"]
per = 6
# use the last 3 complete steps
usable = len(marks) // per
agg = defaultdict(lambda: [0.0, 0.0, 0])
import re
top = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
for st in range(max(0, usable - 4), usable - 1):
    for ph in range(per):
        a, b = marks[st * per + ph], marks[st * per + ph + 1]
        ks = main[a + 1:b]
        busy = sum(e - s for s, e, _, _ in ks)
        for s_, e_, nm_, _ in ks:
            t_ = top[names[ph]][re.sub(r"\(anonymous namespace\)::|^void ", "", nm_)[:100]]
            t_[0] += 1; t_[1] += (e_ - s_) / 1e3
        span = (main[b][0] - main[a][1])
        g = agg[names[ph]]
        g[0] += busy / 1e6; g[1] += span / 1e6; g[2] += len(ks)
n = min(3, usable - 1)
print("%-14s %10s %10s %8s   (mean of %d steps, under the tracer)" % ("phase", "busy_ms", "span_ms", "kernels", n))
for nm in names:
    g = agg[nm]
    print("%-14s %10.2f %10.2f %8d" % (nm, g[0] / n, g[1] / n, g[2] // n))
for nm in ("forward", "loss", "backward"):
    print("--", nm)
    for k, (c, t) in sorted(top[nm].items(), key=lambda kv: -kv[1][1])[:22]:
        print("   %7.1f us/step %5.1f calls/step  %s" % (t / n, c / n, k))
