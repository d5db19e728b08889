#!/bin/bash
# HBM traffic + issue counters of one operator's kernels, one rocprofv3 --pmc pass per counter group
# (counter passes carry --kernel-trace only).   tools/pmc_op.sh ball 16 [name-filter] > profiles/rNN_<op>_pmc.txt
OP=${1:-ball}; B=${2:-16}; FILTER=${3:-grid}
export TMPDIR=/tmp
EXTRA=${4:-}
for PM in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" $EXTRA; do
  rm -rf /tmp/pmc
  timeout 300 rocprofv3 --pmc $PM --kernel-trace --output-format csv -d /tmp/pmc -o p -- python tools/one_op.py $OP $B 3 > /dev/null 2>&1
  python - "$FILTER" <<'PY'
import csv, glob, sys
flt = sys.argv[1]
f = glob.glob("/tmp/pmc/*counter_collection.csv")
agg = {}
for r in csv.DictReader(open(f[0])):
    if flt in r["Kernel_Name"]:
        agg.setdefault((r["Kernel_Name"][:40], r["Counter_Name"]), []).append(float(r["Counter_Value"]))
for (k, c), v in sorted(agg.items()):
    print("%-42s %-22s %s" % (k, c, " ".join("%.5g" % x for x in v)))
PY
done
