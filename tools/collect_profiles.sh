#!/bin/bash
# Copies the artefacts of `bash tools/profile_r04.sh <dir>` (made on the GPU box, merged back under gpurun_out/) into profiles/ under
# the round's names:  bash tools/collect_profiles.sh gpurun_out/r04/final2 r04
src=${1:?source directory}; r=${2:-r04}
for f in bench_line.json bench_kernel_stats.csv; do cp "$src/$f" "profiles/${r}_$f"; done
for f in steady phase_busy native_forward native_loss native_backward step_hbm_traffic ball_query_pmc knn_clamped_pmc knn_plain_pmc ops \
         graph_step flowstep3d corr_layer ball_ab; do
  [ -f "$src/$f.txt" ] && grep -v "amdgpu.ids" "$src/$f.txt" > "profiles/${r}_$f.txt"
done
cat "$src"/config_sapien.txt "$src"/config_sapien_graph.txt "$src"/config_ogcdr.txt "$src"/config_ogcdr_fp32.txt "$src"/config_waymo.txt "$src"/config_kittisf.txt 2>/dev/null \
  | grep "ms/step" > "profiles/${r}_configs.txt"
ls -la profiles/${r}_* | wc -l
