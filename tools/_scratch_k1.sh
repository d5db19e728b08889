export PYTHONPATH=$PWD
for c in sapien ogcdr kittisf waymo; do
for g in 0 1; do echo -n "GRAPH=$g: "; GRAPH=$g timeout 300 python tools/bench_config.py config/${c}_unsup_synthetic.yaml 30 2>&1 | grep "ms/step" | cut -c1-95; done; done
