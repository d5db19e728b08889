#!/bin/bash
# Round-2 measurement artefacts in one go (on the GPU box, from the repo root):  bash tools/profile_r02.sh gpurun_out/r02
out=${1:-gpurun_out/r02}
mkdir -p "$out"
bash tools/profile_round.sh "$out" > /dev/null 2>&1
bash tools/pmc_op2.sh ball 16 grid > "$out/ball_query_pmc.txt" 2>&1
bash tools/pmc_op2.sh knnc 16 grid > "$out/knn_clamped_pmc.txt" 2>&1
bash tools/pmc_op2.sh knn 16 grid > "$out/knn_plain_pmc.txt" 2>&1
timeout 600 python tools/bench_ops.py --ops knn,knnc,nn3,ball,fps,group,interp,conv,gn --iters 20 > "$out/ops.txt" 2>&1
timeout 300 python tools/gn_bwd_compare.py > "$out/gn_bwd_compare.txt" 2>&1
timeout 300 python tools/corr_layer_time.py all > "$out/corr_layer.txt" 2>&1
tail -1 "$out/bench_line.json" | cut -c1-200
