#!/bin/bash
# The round's measurement artefacts in one go (on the GPU box, from the repo root):  bash tools/profile_r06.sh gpurun_out/r06/final
out=${1:-gpurun_out/r06}
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
# round 6: object-aware ICP at its own shape, the FlowStep3D training step, the deterministic mode's cost, the graph with RCCL inside
{ timeout 300 python tools/bench_icp.py 4 20; bash tools/kprof.sh /tmp/icp_k.txt $PWD/tools/bench_icp.py 4 20 > /dev/null 2>&1; grep -v "Cijk\|at::native" /tmp/icp_k.txt | head -12; } > "$out/oa_icp.txt" 2>&1
{ timeout 600 python tools/flow_train_prof.py 8 | tail -1; timeout 600 python tools/flow_graph_step.py 4 | grep -v Warn | tail -4; bash tools/kprof.sh /tmp/flow_k.txt $PWD/tools/flow_train_prof.py 6 > /dev/null 2>&1; head -30 /tmp/flow_k.txt | cut -c1-150; } > "$out/flow_train.txt" 2>&1
{ echo "C4 step (bench.py --timed-only), atomic kernels / OGC_DETERMINISTIC=1:"; for d in 0 1; do OGC_DETERMINISTIC=$d timeout 900 python bench.py --timed-only --steps 10 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('  OGC_DETERMINISTIC=$d  ms/step', d['ms_per_step'])"; done
  for w in seg flow; do OGC_DETERMINISTIC=1 timeout 600 python tools/det_probe.py $w 3 2>&1 | grep "differ\|sha256\|switch"; done
  for w in seg flow; do OGC_DETERMINISTIC=0 timeout 600 python tools/det_probe.py $w 3 2>&1 | grep "differ between\|switch"; done; } > "$out/deterministic.txt" 2>&1
HSA_ENABLE_IPC_MODE_LEGACY=0 timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/graph_rccl.py 8192 2>&1 | grep "graph+rccl\|losses\|GRAPH_RCCL" >> "$out/graph_step.txt"
{ for v in "" "monitor" "monitor,segnet-slots" "all"; do echo "== OGC_GRAPH_INLINE='$v'"; OGC_GRAPH_INLINE="$v" timeout 300 python tools/graph_step.py 8192 2>&1 | grep "eager :\|graph :"; done; } >> "$out/graph_step.txt" 2>&1
timeout 900 python tools/gate_flip_list.py 3 > /dev/null 2> "$out/gate_flip.txt"
{ for mw in 8 4; do echo "== ball_query_cells_kernel built for $mw wavefronts per SIMD"; timeout 120 tools/_bin/bq_probe_mw$mw 2.0 16 2>&1 | grep -E "four lanes|candidates|rank sort|emit|differ" | tail -5; done; } > "$out/bq_occupancy.txt" 2>&1
