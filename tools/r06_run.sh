export PYTHONPATH=$PWD TMPDIR=/tmp
mkdir -p gpurun_out/r06
timeout 600 python tools/flow_train_prof.py 8 2>&1 | tail -1
timeout 900 python -m pytest tests/test_driver_golden.py tests/test_golden_gpu.py tests/test_drivers_gpu.py tests/test_config_sizes_gpu.py tests/test_flow_glue_gpu.py -q -m gpu 2>&1 | tail -3
