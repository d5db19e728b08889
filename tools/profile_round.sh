#!/bin/bash
# The round's measurement artefacts in one go (run on the GPU box from the repo root):
#   bench line, rocprofv3 kernel stats of the bench, marker-exact steady table, per-phase table.
#   bash tools/profile_round.sh <out_dir under gpurun_out/>
set -u
out=${1:-gpurun_out/round}
mkdir -p "$out"
export TMPDIR=/tmp
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1   # a fresh box pages the image in and clocks up
timeout 600 python bench.py > "$out/bench_line.json" 2> "$out/bench_err.txt"
rm -rf /tmp/pr_a /tmp/pr_b
OGC_BENCH_MARK=1 timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/pr_a -o t --output-format csv -- python bench.py --no-cpu-baseline > "$out/bench_traced.log" 2>&1
cp "$(find /tmp/pr_a -name '*kernel_stats.csv' | head -1)" "$out/bench_kernel_stats.csv"
python tools/prof_summary.py "$(find /tmp/pr_a -name "*kernel_trace.csv" | head -1)" 150 10 > "$out/steady.txt" 2>&1
MARK=1 timeout 600 rocprofv3 --kernel-trace -d /tmp/pr_b -o t --output-format csv -- python tools/step_timeline.py > "$out/timeline.log" 2>&1
f=$(find /tmp/pr_b -name '*kernel_trace.csv' | head -1)
python tools/phase_busy.py "$f" > "$out/phase_busy.txt" 2>&1
timeout 300 python tools/step_timeline.py > "$out/timeline_untraced.log" 2>&1
python tools/native_ctx.py "$f" backward > "$out/native_backward.txt" 2>&1
tail -1 "$out/bench_line.json" | cut -c1-300
head -8 "$out/phase_busy.txt"
