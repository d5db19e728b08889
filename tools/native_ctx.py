"""Where do the framework's own small kernels (copies, adds, reductions) sit in the backward pass?  For the last
complete step of a rocprofv3 kernel trace of `MARK=1 python tools/step_timeline.py`, prints every at::native kernel of
the chosen phase with its duration, grid size and the nearest non-native kernels before / after it.  Development tool.
    python tools/native_ctx.py <kernel_trace.csv> [phase=backward]"""
import csv, re, sys
from collections import defaultdict
rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        grid = int(r.get("Grid_Size_X", r.get("Grid_Size", 0)) or 0)
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"],
                     r.get("Stream_Id", r.get("Queue_Id", "0")), grid))
rows.sort()
cnt = defaultdict(int)
for r in rows:
    cnt[r[3]] += 1
mq = max(cnt, key=cnt.get)
main = [r for r in rows if r[3] == mq]
marks = [i for i, r in enumerate(main) if "spin" in r[2].lower() or "sleep" in r[2].lower()]
names = ["forward", "loss", "backward", "(side work)", "optimizer", "between steps"]
phase = sys.argv[2] if len(sys.argv) > 2 else "backward"
ph = names.index(phase)
st = len(marks) // 6 - 2
a, b = marks[st * 6 + ph], marks[st * 6 + ph + 1]
ks = main[a + 1:b]


def short(n):
    n = re.sub(r"\(anonymous namespace\)::|^void |at::native::", "", n)
    m = re.search(r"(direct_copy|CUDAFunctor_add|MulFunctor|DivFunctor|where_kernel|masked_fill|FillFunctor|MeanOps|"
                  r"sum_functor|reduce_kernel|CatArray|fillBuffer|threshold|sigmoid|softmax|clamp|multi_tensor|"
                  r"layer_norm|neg|exp|log|sqrt|pow|abs)[A-Za-z_]*", n)
    if n.startswith(("elementwise", "vectorized", "unrolled", "reduce")) and m:
        return n.split("<")[0] + ":" + m.group(0)
    return n[:60]


def native(n):
    return "at::native" in n or "rocclr" in n


tot = 0.0
for i, (s, e, n, _, g) in enumerate(ks):
    if not native(n):
        continue
    j = i - 1
    while j >= 0 and native(ks[j][2]):
        j -= 1
    k = i + 1
    while k < len(ks) and native(ks[k][2]):
        k += 1
    prev = short(ks[j][2]) if j >= 0 else "-"
    nxt = short(ks[k][2]) if k < len(ks) else "-"
    tot += (e - s) / 1e3
    print("%4d %7.1f us grid %9d  %-44s | after %-40s | before %s" % (i, (e - s) / 1e3, g, short(n)[:44], prev[:40], nxt[:40]))
print("native kernels in %s: %.1f us" % (phase, tot))
