#!/bin/bash
# rocprofv3 kernel statistics of the other BASELINE configurations (run on the GPU box from the repo root):
#   bash tools/config_stats.sh gpurun_out/cfg    -> c2_ogcdr_bf16 / c3_flowstep3d / c5_waymo: bench line + top kernels
out=${1:-gpurun_out/cfg}
mkdir -p "$out"
export PYTHONPATH=$PWD TMPDIR=/tmp
top() {  # $1 = rocprof output dir, $2 = destination
  python - "$1" "$2" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[2], "a") as o:
    o.write("%-100s %8s %12s %10s %6s\n" % ("kernel (rocprofv3 --kernel-trace --stats, whole process)", "calls", "total_ms", "avg_us", "%"))
    for r in rows[:30]:
        o.write("%-100s %8s %12.3f %10.1f %6.1f\n" % (r["Name"][:100], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                                    float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
}
run() {  # $1 = name, rest = command
  name=$1; shift
  "$@" 2>&1 | grep -v "amdgpu.ids\|UserWarning\|run_backward" | tail -3 > "$out/$name.txt"
  rm -rf /tmp/cs_$name
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cs_$name -o s -- "$@" > /dev/null 2>&1
  top /tmp/cs_$name "$out/$name.txt"
}
run c2_ogcdr_bf16 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20
run c5_waymo python tools/bench_config.py config/waymo_unsup_synthetic.yaml 20
run c3_flowstep3d python tools/bench_flow.py 8192 1
run c1_sapien python tools/bench_config.py config/sapien_unsup_synthetic.yaml 20
ls -la "$out"
