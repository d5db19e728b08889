export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 1500 python -m pytest tests/test_deterministic_gpu.py -q -m gpu > $O/t_det.log 2>&1; tail -15 $O/t_det.log
for i in 1 2 3; do timeout 900 python -m pytest tests/test_driver_golden.py -q -m gpu -k "flow_trainer_replays" 2>&1 | tail -1; done
# cost of the switch at C4
timeout 600 python bench.py --timed-only --steps 10 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 atomic kernels  ms/step', d['ms_per_step'])"
OGC_DETERMINISTIC=1 timeout 900 python bench.py --timed-only --steps 10 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C4 deterministic    ms/step', d['ms_per_step'])"
