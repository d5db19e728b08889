export PYTHONPATH=$PWD TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 tools/graph_rccl.py 1024 2>&1 | grep -v Warn | tail -12
