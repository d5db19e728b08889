export PYTHONPATH=$PWD TMPDIR=/tmp
timeout 900 python -m pytest tests/test_golden_gpu.py tests/test_driver_golden.py tests/test_deterministic_gpu.py tests/test_truth_f64_gpu.py tests/test_flow_glue_gpu.py tests/test_config_sizes_gpu.py -q -m gpu 2>&1 | tail -3
for v in 1 0 1 0; do OGC_BN_FOLD=$v timeout 600 python tools/flow_train_prof.py 8 2>&1 | tail -1 | cut -c100-200 | sed "s/^/fold=$v /"; done
