export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06
O=$PWD/gpurun_out/r06
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_flow_store.py tests/test_fused_loss_gpu.py tests/test_truth_f64_gpu.py -q -m gpu > $O/t_g.log 2>&1; tail -5 $O/t_g.log
timeout 300 python tools/bench_icp.py 4 20 > $O/icp.txt 2>&1; tail -3 $O/icp.txt
bash tools/kprof.sh $O/icp_kernels.txt $PWD/tools/bench_icp.py 4 20 > $O/icp_kprof.log 2>&1; grep -v "Cijk\|at::native" $O/icp_kernels.txt | head -12
