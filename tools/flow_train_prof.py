"""One FlowStep3D training step at C3's training shape (4 pairs of 8192 points, iters = 4) in a loop, for rocprofv3 --kernel-trace --stats."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
dev = torch.device("cuda", 0)
print(bench.config3_flow_train_reading(dev, steps=int(sys.argv[1]) if len(sys.argv) > 1 else 6, warm=2))
