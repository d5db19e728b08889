"""Per queue and per phase of the last complete step of a `MARK=1 tools/step_timeline.py` kernel trace: number of
kernels, busy time, first start and last end relative to the phase start.  Shows how long the chains on the side
queues (slot branch, geometry plans, monitors) are next to the main queue's work.  Development tool.
    python tools/queue_phases.py <kernel_trace.csv>"""
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
marks = [r for r in main if "spin" in r[2].lower() or "sleep" in r[2].lower()]
names = ["forward", "loss", "backward", "(side work)", "optimizer", "between steps"]
st = len(marks) // 6 - 2
for ph in range(5):
    t0, t1 = marks[st * 6 + ph][1], marks[st * 6 + ph + 1][0]
    print("-- %s: %.2f ms (traced)" % (names[ph], (t1 - t0) / 1e6))
    per = defaultdict(list)
    for s, e, n, q in rows:
        if s >= t0 and s < t1 and not ("spin" in n.lower() or "sleep" in n.lower()):
            per[q].append((s, e, n))
    for q, ks in sorted(per.items(), key=lambda kv: -len(kv[1])):
        busy = sum(e - s for s, e, _ in ks) / 1e6
        print("   queue %-4s %s kernels %4d  busy %6.2f ms  first +%6.2f ms  last end +%6.2f ms   e.g. %s" % (
            q, "(main)" if q == mq else "      ", len(ks), busy, (ks[0][0] - t0) / 1e6, (max(e for _, e, _ in ks) - t0) / 1e6,
            ks[len(ks) // 2][2][:50]))
        if q != mq and len(ks) > 30:
            agg = defaultdict(lambda: [0, 0.0])
            for s_, e_, n_ in ks:
                a = agg[n_[:110]]
                a[0] += 1
                a[1] += (e_ - s_) / 1e3
            for n_, (c_, t_) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
                print("        %7.1f us %3d x  %s" % (t_, c_, n_))
if len(sys.argv) > 2:   # python tools/queue_phases.py <trace> <phase> <queue rank>: the kernels of that queue in order
    ph = names.index(sys.argv[2])
    t0, t1 = marks[st * 6 + ph][1], marks[st * 6 + ph + 1][0]
    per = defaultdict(list)
    for s, e, n, q in rows:
        if t0 <= s < t1:
            per[q].append((s, e, n))
    q = sorted(per, key=lambda k: -len(per[k]))[int(sys.argv[3])]
    import re
    for s, e, n in per[q]:
        n = re.sub(r"\(anonymous namespace\)::|^void |at::native::", "", n)
        m = re.search(r"(direct_copy|CUDAFunctor_add|MulFunctor|DivFunctor|where_kernel|masked_fill|FillFunctor|MeanOps|sum_functor|threshold|neg_kernel|sqrt|clamp)", n)
        print("%8.1f +%7.3f ms  %s" % ((e - s) / 1e3, (s - t0) / 1e6, (n.split("<")[0] + ":" + m.group(0)) if m else n[:90]))
