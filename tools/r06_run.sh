export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_act16_gpu.py tests/test_bf16_gpu.py tests/test_config_sizes_gpu.py tests/test_fallbacks_gpu.py tests/test_grouped_first_layer_gpu.py -q -m gpu 2>&1 | tail -3
for v in 1 0 1 0; do OGC_GROUP_LINEAR_DIRECT=$v timeout 300 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 2>&1 | grep "ms/step" | sed "s/^/direct=$v  /"; done
