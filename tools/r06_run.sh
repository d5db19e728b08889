export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config2_ogcdr_bf16']['ms_per_step'], d['config3_flow_train']['ms_per_step'], d['oa_icp']['ms_per_call'], d['ms_per_step_with_h2d'], d.get('ms_per_step_hip_graph'))"
