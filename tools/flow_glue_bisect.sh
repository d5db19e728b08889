# which groups of the FlowStep3D inference glue move the flow trainer replay (tests/test_driver_golden.py) — three runs each (the
# training part of the replay is not deterministic to the last bit: atomics)
for w in soft_corr,linear_cn soft_corr,gru soft_corr,linear_cn,gru soft_corr,three_nn_w,linear_cn; do
  echo "== off: $w"
  for i in 1 2 3; do OGC_FLOW_GLUE_OFF=$w python -m pytest tests/test_driver_golden.py -q -k "train_flow_trainer_replays_the_reference_trainer_gpu" 2>&1 | grep -E "AssertionError: epoch|passed" | cut -c1-100; done
done
