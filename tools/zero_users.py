"""Which operators still zero a buffer of their own: for every ogc_zero_kernel of the last step of a kernel trace (cut at
adam_update_kernel), its size (grid x workgroup) and the kernel that follows it on the same queue.
   python tools/zero_users.py <kernel_trace.csv>"""
import csv, sys
rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"),
                 int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0), int(r.get("Workgroup_Size_X", r.get("Workgroup_Size", 0)) or 0)))
rows.sort()
cuts = [e for s, e, n, *_ in rows if "adam_update_kernel" in n]
lo, hi = cuts[-2], cuts[-1]
sel = [r for r in rows if lo < r[1] <= hi]
byq = {}
for r in sel:
    byq.setdefault(r[3], []).append(r)
out = {}
for q, v in byq.items():
    for a, b in zip(v, v[1:]):
        if "ogc_zero_kernel" in a[2]:
            key = (q, a[4], b[2].split("(")[0][-80:])
            out[key] = out.get(key, 0) + 1
print("%d zero fills in the step" % sum(out.values()))
for (q, grid, nxt), c in sorted(out.items(), key=lambda kv: (kv[0][0], -kv[1])):
    print("  queue %s  %3d x  grid %8d   -> %s" % (q, c, grid, nxt))
aten = {}
for r in sel:
    if "at::native" in r[2] or "rocclr" in r[2] or "Cijk" in r[2]:
        k = (r[3], r[2].split("(")[0][:110])
        v = aten.setdefault(k, [0, 0.0])
        v[0] += 1
        v[1] += (r[1] - r[0]) / 1e3
print("framework / library kernels of the step by queue:")
for (q, n), (c, t) in sorted(aten.items(), key=lambda kv: (kv[0][0], -kv[1][1])):
    print("  queue %s %3d x %7.1f us  %s" % (q, c, t, n))
