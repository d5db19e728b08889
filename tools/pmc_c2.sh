#!/bin/bash
# HBM traffic of a C2 training step (segnet_ogcdr, 32 clouds x 4096 points, matmul_precision bf16), 16-bit activations on / off:
# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (counter passes carry --kernel-trace only) over
# `python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 6` (5 warm-up + 6 timed steps: 11 equal steps), per-kernel sums
# divided by 11.      bash tools/pmc_c2.sh > profiles/rNN_c2_hbm_traffic.txt
export TMPDIR=/tmp PYTHONPATH=$PWD
for MODE in 1 0; do
  for PM in FETCH_SIZE WRITE_SIZE; do
    for try in 1 2 3; do
      rm -rf /tmp/pc2_${MODE}_$PM
      OGC_ACT16=$MODE timeout 500 rocprofv3 --pmc $PM --kernel-trace --output-format csv -d /tmp/pc2_${MODE}_$PM -o p -- \
        python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 6 > /tmp/pc2_${MODE}_$PM.log 2>&1
      csv=$(ls /tmp/pc2_${MODE}_$PM/*counter_collection.csv 2> /dev/null | head -1)
      [ -n "$csv" ] && [ "$(wc -l < "$csv")" -gt 1000 ] && break
      echo "pmc_c2.sh: ACT16=$MODE $PM pass $try failed: $(tail -2 /tmp/pc2_${MODE}_$PM.log | tr '\n' ' ')" >&2
    done
  done
done
python - <<'PY' || exit 1
import csv, glob, re
from collections import defaultdict
STEPS = 11
for mode, title in (("1", "bf16 operands + 16-bit activations (default)"), ("0", "bf16 operands, fp32 activations (OGC_ACT16=0)")):
    tot = {}
    for pm in ("FETCH_SIZE", "WRITE_SIZE"):
        f = glob.glob("/tmp/pc2_%s_%s/*counter_collection.csv" % (mode, pm))[0]
        agg = defaultdict(lambda: [0, 0.0])
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != pm:
                continue
            a = agg[re.sub(r"\(anonymous namespace\)::|^void ", "", r["Kernel_Name"])[:78]]
            a[0] += 1
            a[1] += float(r["Counter_Value"])
        tot[pm] = agg
    names = sorted(set(tot["FETCH_SIZE"]) | set(tot["WRITE_SIZE"]),
                   key=lambda k: -(tot["FETCH_SIZE"].get(k, [0, 0])[1] + tot["WRITE_SIZE"].get(k, [0, 0])[1]))
    print("== C2 step, %s: MiB per step as reported (KiB counters / 1024, whole process / %d steps); FETCH_SIZE under-reports wide\n"
          "   coalesced reads by 2x on gfx950 (MI355X_MICROARCH.md)" % (title, STEPS))
    print("%-80s %7s %12s %12s" % ("kernel", "calls", "fetch MiB", "write MiB"))
    for k in names[:22]:
        c, fv = tot["FETCH_SIZE"].get(k, [0, 0.0])
        _, wv = tot["WRITE_SIZE"].get(k, [0, 0.0])
        print("%-80s %7.1f %12.1f %12.1f" % (k, c / STEPS, fv / 1024 / STEPS, wv / 1024 / STEPS))
    sf = sum(v[1] for v in tot["FETCH_SIZE"].values())
    sw = sum(v[1] for v in tot["WRITE_SIZE"].values())
    print("%-80s %7s %12.1f %12.1f   -> 2 x fetch + write = %.1f GB per step\n" %
          ("ALL KERNELS", "", sf / 1024 / STEPS, sw / 1024 / STEPS, (2 * sf + sw) * 1024 / STEPS / 1e9))
PY
