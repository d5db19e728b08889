#!/bin/bash
# The round's measurement artefacts in one go (on the GPU box, from the repo root):  bash tools/profile_r05.sh gpurun_out/r05/final
out=${1:-gpurun_out/r05}
mkdir -p "$out"
export PYTHONPATH=$PWD
bash tools/profile_round.sh "$out" > /dev/null 2>&1
f=$(find /tmp/pr_b -name '*kernel_trace.csv' | head -1)
for ph in forward loss; do python tools/native_ctx.py "$f" $ph > "$out/native_$ph.txt" 2>&1; done
bash tools/pmc_step.sh > "$out/step_hbm_traffic.txt" 2> "$out/step_hbm_traffic.err"
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
[ -x tools/_bin/bq_probe ] && { for b in 16 64; do echo "==== $b clouds per launch"; timeout 120 tools/_bin/bq_probe 2.0 $b; done; } > "$out/bq_probe.txt" 2>&1
timeout 300 python tools/library_gemms.py > "$out/library_gemms.txt" 2>&1
{ for cfg in "4 8192" "16 8192" "4 16384" "32 16384"; do set -- $cfg; timeout 300 python bench.py --timed-only --batch $1 --npoint $2 --steps 10 --warmup 3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.readline()); print('batch %s npoint %s: %.2f ms/step, %.0f clouds/s, peak HBM %.1f GiB' % (sys.argv[1], sys.argv[2], d['ms_per_step'], d['value'], d['config']['peak_hbm_gib']))" $1 $2; done; } > "$out/sizing.txt" 2>&1
# C2 (bf16 operands + 16-bit activations): kernel table of the step, the forward kernels alone, and the step with single switches off
bash tools/c2_kernels.sh > /dev/null 2>&1; cp gpurun_out/c2/kernels.txt "$out/c2_kernels.txt" 2>/dev/null
{ echo "persistent kernel (conv1x1_h.hip)"; timeout 300 python tools/gemm16_bench.py 2>&1 | grep "TB/s"; echo "tile kernel (OGC_GEMM16=0)"; OGC_GEMM16=0 timeout 300 python tools/gemm16_bench.py 2>&1 | grep "TB/s"; } > "$out/c2_forward_kernels.txt"
{ for sw in "" "OGC_GEMM16=0" "OGC_ACT16=0" "OGC_ACT16_MOMENT_WIDTH=128"; do echo "== ${sw:-default}"; env $sw timeout 300 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 2>&1 | grep "ms/step"; done; } > "$out/c2_switches.txt"
bash tools/pmc_c2.sh > "$out/c2_hbm_traffic.txt" 2> "$out/c2_hbm_traffic.err"
tail -1 "$out/bench_line.json" | cut -c1-200
