export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/c2
python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 2>&1 | tail -2 > gpurun_out/c2/line.txt
rm -rf /tmp/cs_c2
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cs_c2 -o s -- python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 > /dev/null 2>&1
python - <<'PY'
import csv, glob
f = glob.glob("/tmp/cs_c2/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open("gpurun_out/c2/kernels.txt", "w") as o:
    o.write("total kernel ms over 25 steps: %.1f\n" % (tot / 1e6))
    for r in rows[:60]:
        o.write("%-150s %6s %10.3f %9.1f %5.1f\n" % (r["Name"][:150], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
cat gpurun_out/c2/line.txt
