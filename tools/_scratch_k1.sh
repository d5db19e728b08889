export PYTHONPATH=$PWD
mkdir -p gpurun_out/k6
for v in 1024 256 1024 256; do
echo "OGC_KNN_GRID_MIN=$v" >> gpurun_out/k6/ab.txt
OGC_KNN_GRID_MIN=$v timeout 300 python tools/bench_flow.py 8192 1 2>&1 | grep "forward eval iters=5:" | head -1 >> gpurun_out/k6/ab.txt
OGC_KNN_GRID_MIN=$v timeout 300 python tools/bench_config.py config/sapien_unsup_synthetic.yaml 20 2>&1 | grep "ms/step" | cut -c1-80 >> gpurun_out/k6/ab.txt
done
OGC_KNN_GRID_MIN=128 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py -x -q -k "knn or golden or three" 2>&1 | tail -3 >> gpurun_out/k6/ab.txt
cat gpurun_out/k6/ab.txt
