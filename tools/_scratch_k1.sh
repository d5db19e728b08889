export PYTHONPATH=$PWD
OGC_KNN_LANES=32 python -m pytest tests/test_ops_gpu.py tests/test_golden_gpu.py tests/test_config_sizes_gpu.py tests/test_fullsize_gpu.py -x -q -k "knn or golden or config or three or fullsize" 2>&1 | tail -2
for v in 16 0 16 0; do echo -n "lanes=$v "; OGC_KNN_LANES=$v python tools/bench_flow.py 8192 1 2>&1 | grep "forward eval iters=5:" | head -1; done
for v in 16 32; do echo "== lanes=$v"; OGC_KNN_LANES=$v python tools/bench_ops.py --ops knn --iters 20 2>&1 | grep "^knn   B=1 "; done
