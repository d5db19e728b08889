#!/bin/bash
# Per-kernel time of a command (idle GPU):  bash tools/kprof.sh <outfile> <python args...>
# rocprofv3 with CSV output only (the default rocpd database step can hang on this image) and a hard timeout.
OUT=$1; shift
export TMPDIR=/tmp
rm -rf /tmp/kprof
( cd /tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kprof -o k -- python "$@" ) > /tmp/kprof_stdout.txt 2>&1
grep -v "^W2\|^E2\|rocprofv3" /tmp/kprof_stdout.txt | tail -40
f=$(find /tmp/kprof -name "*kernel_stats.csv" | head -1)
python - "$f" "$OUT" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
lines = ["%-100s %8s %12s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "%")]
for r in rows[:40]:
    lines.append("%-100s %8s %12.1f %10.2f %6s" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e3,
                                                  float(r["AverageNs"]) / 1e3, r["Percentage"]))
open(sys.argv[2], "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:25]))
PY
