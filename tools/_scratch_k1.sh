export PYTHONPATH=$PWD
mkdir -p gpurun_out/k5
for i in 1 2; do
OGC_BF16_WIDE_POOL=0 timeout 300 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 2>&1 | grep "ms/step" | sed 's/^/off: /' >> gpurun_out/k5/ab.txt
OGC_BF16_WIDE_POOL=1 timeout 300 python tools/bench_config.py config/ogcdr_unsup_synthetic.yaml 20 2>&1 | grep "ms/step" | sed 's/^/on:  /' >> gpurun_out/k5/ab.txt
done
python -m pytest tests/test_bf16_gpu.py tests/test_fallbacks_gpu.py -x -q 2>&1 | tail -15 >> gpurun_out/k5/ab.txt
cat gpurun_out/k5/ab.txt
