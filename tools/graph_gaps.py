"""Per-step anatomy of a kernel trace of tools/graph_step.py (PHASE=eager | graph): steps are cut at adam_update_kernel; for the
last STEPS steps the span, per-queue busy time, kernel count, the gaps of the busiest queue by size, and per-kernel time.
   rocprofv3 --kernel-trace --output-format csv -d D -o k -- python tools/graph_step.py 8192      (PHASE set)
   python tools/graph_gaps.py D/.../k_kernel_trace.csv out.json      |      python tools/graph_gaps.py --compare a.json b.json"""
import csv, json, sys

if sys.argv[1] == "--compare":
    a, b = json.load(open(sys.argv[2])), json.load(open(sys.argv[3]))
    names = sorted(set(a["kernels"]) | set(b["kernels"]), key=lambda n: -abs(a["kernels"].get(n, [0, 0])[0] - b["kernels"].get(n, [0, 0])[0]))
    print("%-90s %10s %10s %8s" % ("kernel (us/step, calls/step)", "A", "B", "B-A"))
    for n in names[:30]:
        x, y = a["kernels"].get(n, [0, 0]), b["kernels"].get(n, [0, 0])
        print("%-90s %7.1f/%-3.0f %7.1f/%-3.0f %+8.1f" % (n[:90], x[0], x[1], y[0], y[1], y[0] - x[0]))
    print("sum of kernel time per step: A %.1f us, B %.1f us" % (sum(v[0] for v in a["kernels"].values()), sum(v[0] for v in b["kernels"].values())))
    sys.exit(0)

STEPS = 10
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0")))
rows.sort()
cuts = [e for s, e, n, q in rows if "adam_update_kernel" in n]
assert len(cuts) > STEPS + 1, "fewer than %d steps in the trace" % (STEPS + 1)
lo, hi = cuts[-STEPS - 1], cuts[-1]
sel = [r for r in rows if r[1] > lo and r[1] <= hi]
span = (hi - lo) / STEPS / 1e3


def union(ivs):
    ivs = sorted(ivs)
    busy, cs, ce = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > ce:
            busy += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    return busy + ce - cs


byq = {}
for s, e, n, q in sel:
    byq.setdefault(q, []).append((s, e, n))
print("step span %.1f us over %d steps, %d kernels/step" % (span, STEPS, len(sel) // STEPS))
print("any-queue busy %.1f us/step" % (union([(s, e) for s, e, _, _ in sel]) / STEPS / 1e3))
main = max(byq, key=lambda q: union([(s, e) for s, e, _ in byq[q]]))
for q, v in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    print("  queue %-4s %6.1f kernels/step  busy %8.1f us/step%s" % (q, len(v) / STEPS, union([(s, e) for s, e, _ in v]) / STEPS / 1e3,
                                                                   "   <- main" if q == main else ""))
mq = sorted(byq[main])
buckets = [(0, 2), (2, 5), (5, 10), (10, 20), (20, 50), (50, 1e9)]
tot = {b: [0, 0.0] for b in buckets}
big = []
for (s0, e0, n0), (s1, e1, n1) in zip(mq, mq[1:]):
    g = (s1 - e0) / 1e3
    if g <= 0:
        continue
    for b in buckets:
        if b[0] <= g < b[1]:
            tot[b][0] += 1
            tot[b][1] += g
    if g >= 20:
        big.append((g, n0[:60], n1[:60]))
print("gaps on the main queue per step:")
for b in buckets:
    print("   %5s..%-5s us: %6.1f gaps, %8.1f us" % (b[0], b[1] if b[1] < 1e9 else "", tot[b][0] / STEPS, tot[b][1] / STEPS))
big.sort(reverse=True)
seen = {}
for g, a, b in big:
    seen.setdefault((a, b), []).append(g)
print("largest gaps (>= 20 us), by kernel pair: count/step, mean us")
for (a, b), v in sorted(seen.items(), key=lambda kv: -sum(kv[1]))[:25]:
    print("   %5.1f x %6.1f us   %s  ->  %s" % (len(v) / STEPS, sum(v) / len(v), a, b))
kern = {}
for s, e, n, q in sel:
    k = kern.setdefault(n.split("(")[0][:120], [0.0, 0])
    k[0] += (e - s) / 1e3 / STEPS
    k[1] += 1.0 / STEPS
if len(sys.argv) > 2:
    json.dump({"span": span, "kernels": kern}, open(sys.argv[2], "w"))
