#!/bin/bash
# Copies the artefacts of `bash tools/profile_r05.sh <dir>` (made on the GPU box, merged back under gpurun_out/) into profiles/ under
# the round's names:  bash tools/collect_profiles.sh gpurun_out/r05/final r05
# A source file that is empty, or holds a Python traceback, a segmentation fault or a rocprofv3 abort instead of a table, is NOT
# copied: the tracked file of the earlier collection stays, and the name is listed on stderr (exit status 1 at the end).
src=${1:?source directory}; r=${2:-r05}
bad=0
good() {  # good <file>: a non-empty artefact without the marks of a failed run
  [ -s "$1" ] || return 1
  if grep -q -E "Traceback \(most recent call last\)|Segmentation fault|core dumped|rocprofv3: error|No such file or directory: '/tmp/p" "$1"; then return 1; fi
  return 0
}
take() {  # take <source> <destination> [filter]
  if good "$1"; then
    if [ -n "$3" ]; then grep -v "$3" "$1" > "$2"; else cp "$1" "$2"; fi
  else
    echo "collect_profiles.sh: NOT copied (missing, empty or a failed run): $1 -> $2 keeps its earlier contents" >&2
    bad=1
  fi
}
for f in bench_line.json bench_kernel_stats.csv; do take "$src/$f" "profiles/${r}_$f"; done
for f in steady phase_busy native_forward native_loss native_backward step_hbm_traffic ball_query_pmc knn_clamped_pmc knn_plain_pmc ops \
         graph_step flowstep3d corr_layer ball_ab bq_probe library_gemms sizing c2_kernels c2_forward_kernels c2_switches c2_hbm_traffic oa_icp flow_train deterministic gate_flip bq_occupancy; do
  [ -f "$src/$f.txt" ] && take "$src/$f.txt" "profiles/${r}_$f.txt" "amdgpu.ids"
done
cfg=$(cat "$src"/config_sapien.txt "$src"/config_sapien_graph.txt "$src"/config_ogcdr.txt "$src"/config_ogcdr_fp32.txt "$src"/config_waymo.txt "$src"/config_kittisf.txt 2>/dev/null \
  | grep "ms/step")
[ -n "$cfg" ] && echo "$cfg" > "profiles/${r}_configs.txt"
ls -la profiles/${r}_* | wc -l
exit $bad
