export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
(time timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err); tail -3 $O/bench_full.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r06/bench_full.json').read().strip().splitlines()[-1])
for k in ('value','ms_per_step','ms_per_step_hip_graph','launch_thread_ms_per_step','config3_flow_train','oa_icp'):
    print(k, json.dumps(d.get(k))[:900])
print('roofline frac', d['roofline']['frac'], 'c2', d.get('config2_ogcdr_bf16',{}).get('ms_per_step'))
PY
