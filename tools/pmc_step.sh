#!/bin/bash
# HBM traffic of one C4 training step per kernel: rocprofv3 --pmc FETCH_SIZE and WRITE_SIZE in separate passes
# (counter passes carry --kernel-trace only) over `OGC_BENCH_MARK=1 python bench.py --timed-only --steps 3 --warmup 2`; the
# marker kernel of every timed step delimits the window.    bash tools/pmc_step.sh > profiles/rNN_step_hbm_traffic.txt
# A pass that dies (rocprofv3 has crashed with a segmentation fault under some builds of this repo) is repeated up to three
# times; when a counter still has no table the script prints NOTHING on stdout and exits 1 — tools/collect_profiles.sh then keeps
# the file it has.
export TMPDIR=/tmp
for PM in FETCH_SIZE WRITE_SIZE; do
  ok=0
  for try in 1 2 3; do
    rm -rf /tmp/pmc_$PM
    OGC_BENCH_MARK=1 timeout 500 rocprofv3 --pmc $PM --kernel-trace --output-format csv -d /tmp/pmc_$PM -o p -- \
      python bench.py --timed-only --steps 3 --warmup 2 > /tmp/pmc_$PM.log 2>&1
    rc=$?
    # (a crash of the profiler while it exits, AFTER the table was written, still leaves a usable table: markers of all steps present)
    csv=$(ls /tmp/pmc_$PM/*counter_collection.csv 2> /dev/null | head -1)
    if [ -n "$csv" ] && [ "$(grep -c -i -E "spin|sleep" "$csv")" -ge 3 ]; then ok=1; break; fi
    echo "pmc_step.sh: $PM pass $try failed (exit $rc): $(tail -2 /tmp/pmc_$PM.log | tr '\n' ' ')" >&2
  done
  if [ $ok -ne 1 ]; then echo "pmc_step.sh: no $PM table after three passes" >&2; exit 1; fi
done
python - <<'PY' || exit 1
import csv, glob, re
from collections import defaultdict
tot = {}
steps = None
for pm in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob("/tmp/pmc_%s/*counter_collection.csv" % pm)[0]
    rows = [r for r in csv.DictReader(open(f)) if r["Counter_Name"] == pm]
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    marks = [i for i, r in enumerate(rows) if "spin" in r["Kernel_Name"].lower() or "sleep" in r["Kernel_Name"].lower()]
    lo, hi = marks[0], marks[-1]          # whole steps only: from the first marker to the last
    steps = len(marks) - 1
    agg = defaultdict(lambda: [0, 0.0])
    for r in rows[lo:hi]:
        n = r["Kernel_Name"]
        if "spin" in n.lower() or "sleep" in n.lower():
            continue
        a = agg[re.sub(r"\(anonymous namespace\)::|^void ", "", n)[:78]]
        a[0] += 1
        a[1] += float(r["Counter_Value"])
    tot[pm] = agg
names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]),
               key=lambda k: -(tot["FETCH_SIZE"].get(k, [0, 0])[1] + tot["WRITE_SIZE"].get(k, [0, 0])[1]))
print("HBM-side traffic per C4 training step (16 clouds x 8192 points), mean of %d whole timed steps (marker to marker).\n"
      "MiB per step as reported\n"
      "(KiB counters / 1024); FETCH_SIZE under-reports wide coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md), so the\n"
      "true read volume of the streaming kernels is up to twice the column." % steps)
print("%-80s %7s %12s %12s" % ("kernel", "calls", "fetch MiB", "write MiB"))
sf = sw = 0.0
for k in names[:45]:
    c, fv = tot["FETCH_SIZE"].get(k, [0, 0.0])
    _, wv = tot["WRITE_SIZE"].get(k, [0, 0.0])
    print("%-80s %7.1f %12.1f %12.1f" % (k, c / steps, fv / 1024 / steps, wv / 1024 / steps))
for k in names:
    sf += tot["FETCH_SIZE"].get(k, [0, 0.0])[1]
    sw += tot["WRITE_SIZE"].get(k, [0, 0.0])[1]
print("%-80s %7s %12.1f %12.1f" % ("ALL KERNELS", "", sf / 1024 / steps, sw / 1024 / steps))
PY
