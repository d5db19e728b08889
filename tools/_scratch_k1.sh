export PYTHONPATH=$PWD
mkdir -p gpurun_out/k2
python tools/bench_flow.py 8192 1 2>&1 | grep -v Warn > gpurun_out/k2/flow_time.txt
python -m pytest tests -x -q -m gpu -k "flow or golden or config_sizes or fps" 2>&1 | tail -5 > gpurun_out/k2/pytest.txt
bash tools/kprof.sh gpurun_out/k2/flow.txt $PWD/tools/bench_flow.py 8192 1 > gpurun_out/k2/flow.log 2>&1
cat gpurun_out/k2/flow_time.txt gpurun_out/k2/pytest.txt
