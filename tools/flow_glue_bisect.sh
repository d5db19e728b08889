for w in linear_cn soft_corr gru advance gather_pair three_nn_w; do
  echo "== off: $w"; OGC_FLOW_GLUE_OFF=$w python -m pytest tests/test_driver_golden.py -q -k "train_flow_trainer_replays_the_reference_trainer_gpu" 2>&1 | grep -E "AssertionError: epoch|passed|failed" | cut -c1-200
done
