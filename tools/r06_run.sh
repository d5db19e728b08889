export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 600 python -m pytest tests/test_bn_fold_gpu.py -q -m gpu 2>&1 | tail -5
