export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_grouped_first_layer_gpu.py -q -m gpu -s 2>&1 | grep -v Warn | tail -12
timeout 900 python -m pytest tests/test_fallbacks_gpu.py tests/test_config_sizes_gpu.py tests/test_bf16_gpu.py tests/test_act16_gpu.py -q -m gpu 2>&1 | tail -3
