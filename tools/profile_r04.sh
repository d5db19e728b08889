#!/bin/bash
# Round-3 measurement artefacts in one go (on the GPU box, from the repo root):  bash tools/profile_r04.sh gpurun_out/r04
out=${1:-gpurun_out/r04}
mkdir -p "$out"
export PYTHONPATH=$PWD
bash tools/profile_round.sh "$out" > /dev/null 2>&1
f=$(find /tmp/pr_b -name '*kernel_trace.csv' | head -1)
for ph in forward loss; do python tools/native_ctx.py "$f" $ph > "$out/native_$ph.txt" 2>&1; done
bash tools/pmc_step.sh > "$out/step_hbm_traffic.txt" 2>&1
bash tools/pmc_op2.sh ball 16 grid > "$out/ball_query_pmc.txt" 2>&1
bash tools/pmc_op2.sh knnc 16 grid > "$out/knn_clamped_pmc.txt" 2>&1
bash tools/pmc_op2.sh knn 16 grid > "$out/knn_plain_pmc.txt" 2>&1
timeout 600 python tools/bench_ops.py  --iters 20 > "$out/ops.txt" 2>&1
timeout 300 python tools/graph_step.py > "$out/graph_step.txt" 2>&1
for c in sapien ogcdr waymo kittisf; do
  timeout 300 python tools/bench_config.py config/${c}_unsup_synthetic.yaml 20 > "$out/config_$c.txt" 2>&1
done
PRECISION=fp32 timeout 300 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 > "$out/config_ogcdr_fp32.txt" 2>&1
GRAPH=1 timeout 300 python tools/bench_config.py config/sapien_unsup_synthetic.yaml 30 > "$out/config_sapien_graph.txt" 2>&1
timeout 300 python tools/bench_flow.py 8192 1 > "$out/flowstep3d.txt" 2>&1
timeout 300 python tools/corr_layer_time.py all > "$out/corr_layer.txt" 2>&1
timeout 300 python tools/bq_ab.py > "$out/ball_ab.txt" 2>&1
tail -1 "$out/bench_line.json" | cut -c1-200
