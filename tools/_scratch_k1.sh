export PYTHONPATH=$PWD
mkdir -p gpurun_out/k4
timeout 300 python tools/bench_flow.py 8192 1 2>&1 | grep -v "Warn\|return Var" > gpurun_out/k4/flow_time.txt
cat gpurun_out/k4/flow_time.txt
