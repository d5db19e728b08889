export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06/final
timeout 300 python bench.py --no-cpu-baseline --steps 10 --warmup 3 > /dev/null 2>&1
timeout 900 python bench.py > gpurun_out/r06/final/bench_line.json 2> gpurun_out/r06/final/bench_err.txt
tail -c 300 gpurun_out/r06/final/bench_line.json
timeout 300 python tools/flow_train_prof.py 8 2>&1 | tail -1 > gpurun_out/r06/final/flow_train_head.txt
