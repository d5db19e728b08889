"""Where do two runs of the flow trainer replay part ways?  (development tool)  Replays tests/test_driver_golden.py::run_flow's
setup and prints, before every training iteration, one checksum per parameter / optimizer-state / buffer tensor; then the flow
predictions' checksums of that iteration.  Diff the outputs of a passing and a failing run."""
import os, sys, hashlib
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import detgen, driver_cases as dc
from ogc_amd.models.flownet_sapien import FlowStep3D
from ogc_amd.train_flow import Trainer, build_flow_criterion
from ogc_amd.train_seg import norm_momentum, schedule_factor
from ogc_amd.train_step import make_optimizer
from ogc_amd.utils.pytorch_util import BNMomentumScheduler, LambdaLR
import tempfile
cfg = dc.FLOW_CFG
dev = "cuda"
net = detgen.fill_module(FlowStep3D(**cfg["flownet"]), 32).to(dev)
opt = make_optimizer(net.parameters(), lr=cfg["lr"], weight_decay=cfg["weight_decay"])
lr_s = LambdaLR(opt, lr_lambda=lambda it: schedule_factor(cfg, it * cfg["batch_size"]))
bn_s = BNMomentumScheduler(net, bn_lambda=lambda it: norm_momentum(cfg, it * cfg["batch_size"]))
tmp = tempfile.mkdtemp()
trainer = Trainer(net, cfg["model_iters"], build_flow_criterion(cfg["loss"]), opt, exp_base=os.path.join(tmp, "flow"), lr_scheduler=lr_s,
                  bnm_scheduler=bn_s, device=torch.device(dev), log=lambda l: None,
                  on_iteration=lambda it, ld, stepped: print("LOSS it %d chamfer#1 %.6f sum %.6f" % (it, ld["chamfer_loss_#1"], ld["sum"]), flush=True))


def h(t):
    return hashlib.md5(t.detach().cpu().numpy().tobytes()).hexdigest()[:8]


inner = trainer._train_it


def recording(it, batch, **kw):
    torch.cuda.synchronize()
    names = dict(net.named_parameters())
    lines = []
    for k, p in names.items():
        st = opt.state.get(p, {})
        lines.append("%s %s %s %s %s" % (k, h(p), h(st["exp_avg"]) if "exp_avg" in st else "-", h(st["exp_avg_sq"]) if "exp_avg_sq" in st else "-",
                                         float(st["step"]) if "step" in st else -1))
    for k, b in net.named_buffers():
        lines.append("%s %s" % (k, h(b)))
    print("STATE it %d all %s" % (it, hashlib.md5("\n".join(lines).encode()).hexdigest()[:8]), flush=True)
    if os.environ.get("STATE_FULL"):
        for l in lines:
            print("  S%d %s" % (it, l))
    out = inner(it, batch, **kw)
    grads = hashlib.md5("".join(h(p.grad) if p.grad is not None else "-" for p in net.parameters()).encode()).hexdigest()[:8]
    print("GRADS it %d %s" % (it, grads), flush=True)
    return out


trainer._train_it = recording
seed = 0
import numpy as _np
gold = _np.load(os.path.join(ROOT, "tests", "golden", "train_flow_trace.npz"), allow_pickle=True)
seed = int(gold["data_seed"][0])
tl = torch.utils.data.DataLoader(dc.FlowPairs(True, seed), batch_size=cfg["batch_size"], shuffle=False)
vl = torch.utils.data.DataLoader(dc.FlowPairs(False, seed), batch_size=cfg["batch_size"], shuffle=False)
trainer.train(cfg["epochs"], tl, vl)
