"""Kernel trace of graph-replayed C4 steps: per queue the busy time, and how much of the slot branch's kernels (small_linear /
attn / layer_norm) overlaps in time with kernels of other queues.   rocprofv3 --kernel-trace ... -- python tools/graph_step.py 8192
then:  python tools/graph_trace.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
t0, t1 = rows[0][0], rows[-1][1]
# last 40 % of the run = the graph-replay phase of tools/graph_step.py (eager steps come first)
cut = t0 + int(0.62 * (t1 - t0))
for label, sel in (("eager phase", [r for r in rows if r[1] < t0 + int(0.45 * (t1 - t0))]), ("graph phase", [r for r in rows if r[0] > cut])):
    if not sel:
        continue
    span = sel[-1][1] - sel[0][0]
    byq = {}
    for s, e, n, q, st in sel:
        byq.setdefault(q, []).append((s, e, n))
    print("%s: span %.1f ms, %d kernels, queues: %s" % (label, span / 1e6, len(sel), {q: round(sum(e - s for s, e, _ in v) / 1e6, 1) for q, v in byq.items()}))
    slot = [(s, e) for s, e, n, q, st in sel if "small_linear" in n or "attn_" in n or "layer_norm" in n]
    other = sorted((s, e) for s, e, n, q, st in sel if not ("small_linear" in n or "attn_" in n or "layer_norm" in n))
    ov = 0
    j = 0
    for s, e in slot:
        for os_, oe in other:
            if oe <= s:
                continue
            if os_ >= e:
                break
            ov += min(e, oe) - max(s, os_)
    tot = sum(e - s for s, e in slot)
    print("   slot-branch kernels: %.2f ms in total, %.2f ms of it overlapped by other kernels (%.0f %%)" % (tot / 1e6, ov / 1e6, 100.0 * ov / max(tot, 1)))
    # union busy
    ivs = sorted((s, e) for s, e, *_ in sel)
    busy, cs, ce = 0, ivs[0][0], ivs[0][1]
    for s, e in ivs[1:]:
        if s > ce:
            busy += ce - cs; cs, ce = s, e
        else:
            ce = max(ce, e)
    busy += ce - cs
    print("   any-queue busy %.1f ms of %.1f ms span (%.1f %% idle)" % (busy / 1e6, span / 1e6, 100 - 100.0 * busy / span))
