export PYTHONPATH=$PWD
for v in 8 16; do echo "== lanes=$v"; OGC_KNN_LANES=$v python tools/bench_ops.py --ops knn,knnc --iters 20 2>&1 | grep "^knn"; done
