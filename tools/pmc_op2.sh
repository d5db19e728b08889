#!/bin/bash
# Kernel durations + issue / wait / occupancy counters of one operator:  bash tools/pmc_op2.sh <op> <B> <filter> > out.txt
# (CSV output, hard timeouts, counter passes separate from each other; kernel-trace only besides --pmc)
OP=${1:-ball}; B=${2:-16}; FILTER=${3:-grid}
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/pm0; timeout 200 rocprofv3 --kernel-trace --output-format csv -d /tmp/pm0 -o p -- python $GRAFT_REPO_ROOT/tools/one_op.py $OP $B 9 > /dev/null 2>&1
# per kernel: calls, MEDIAN, mean, min, max of the launch durations (the first launch of a process can take several times the
# others — code-object load —, which a mean over a handful of launches carries and the median does not)
python - "$FILTER" <<'PY'
import csv, glob, sys
from collections import defaultdict
d = defaultdict(list)
for r in csv.DictReader(open(glob.glob("/tmp/pm0/*kernel_trace.csv")[0])):
    if sys.argv[1] in r["Kernel_Name"]:
        d[r["Kernel_Name"]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for name, v in sorted(d.items()):
    v.sort()
    print("%-60s calls %d median %.2f us avg %.2f us min %.2f max %.2f" % (name[:60], len(v), v[len(v) // 2], sum(v) / len(v), v[0], v[-1]))
PY
for PM in "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_INSTS_SMEM" "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc
  timeout 200 rocprofv3 --pmc $PM --kernel-trace --output-format csv -d /tmp/pmc -o p -- python $GRAFT_REPO_ROOT/tools/one_op.py $OP $B 3 > /dev/null 2>&1
  python - "$FILTER" <<'PY'
import csv, glob, sys
flt = sys.argv[1]
f = glob.glob("/tmp/pmc/*counter_collection.csv")
agg = {}
if f:
    for r in csv.DictReader(open(f[0])):
        if flt in r["Kernel_Name"]:
            agg.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-42s %-22s %s" % (k, c, " ".join("%.5g" % x for x in v)))
PY
done
