# where the flow trainer replay (tests/test_driver_golden.py) lands inside its budget, glue on / off, several runs each (the
# training part of the replay is not deterministic to the last bit: atomics)
for w in "" "OGC_FLOW_GLUE=0"; do
  echo "== ${w:-glue on}"
  for i in 1 2 3 4 5; do env $w OGC_TEST_VERBOSE=1 python -m pytest tests/test_driver_golden.py -q -s -k "train_flow_trainer_replays_the_reference_trainer_gpu" 2>&1 | grep -E "BUDGET epoch 2 validation epe3d_#1|BUDGET epoch 1 validation loss sum" | cut -c1-140; done
done
