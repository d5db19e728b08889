"""Which parameter gradients of a training step differ from run to run (same weights, same batch, fresh process state not needed:
float atomics already differ between two launches), and which entry points of the library the step calls.
   python tools/det_probe.py flow|seg [runs]"""
import collections, hashlib, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: F401
from ogc_amd import _lib
from ogc_amd.utils.synthetic import make_scene_batch

which = sys.argv[1] if len(sys.argv) > 1 else "flow"
runs = int(sys.argv[2]) if len(sys.argv) > 2 else 4
dev = "cuda"


def build():
    torch.manual_seed(10)
    if which == "flow":
        from ogc_amd.models.flownet_sapien import FlowStep3D
        from ogc_amd.losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
        net = FlowStep3D(npoint=512, use_instance_norm=False, loc_flow_nn=8, loc_flow_rad=0.2, k_decay_fact=0.5).to(dev)
        crit = UnsupervisedFlowStep3DLoss(ChamferLoss(2), SmoothLoss(3., 1., {'k': 4, 'radius': 0.05, 'loss_norm': 1},
                                                                      {'k': 8, 'radius': 0.1, 'loss_norm': 1}),
                                          weights=[0.75, 0.25], iters_w=[0.8, 0.2, 0.4, 0.6])
        batch = make_scene_batch(2, 512, 8, seed=3, outdoor=False, aug=False, device=dev)
        return net, crit, batch
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion
    net = MaskFormer3D(n_slot=10, n_point=2048, use_xyz=True, n_transformer_layer=2, transformer_embed_dim=128,
                       transformer_input_pos_enc=False).to(dev)
    return net, build_criterion(KITTI_LOSS), make_scene_batch(2, 2048, 10, seed=3, outdoor=True, aug=True, device=dev)


def grads_of_one_step():
    net, crit, batch = build()
    net.train()
    if which == "flow":
        pcs = batch[0]
        pc1, pc2 = pcs[:, 0].contiguous(), pcs[:, 1].contiguous()
        preds = net(pc1, pc2, pc1, pc2, iters=4)
        loss, _ = crit(pc1, pc2, preds, sync=False)
    else:
        from ogc_amd.train_step import _views, _SplitViews
        flat, pcs_l, flows_l, pcs_s, flows_s = _views(batch)
        masks = net(flat, flat).view(2, 4, 2048, -1)
        *masks_l, masks_s = _SplitViews.apply(masks)
        kw = {"stacked": (pcs_s, masks_s, flows_s)} if getattr(crit, "takes_stacked_views", False) else {}
        loss, _ = crit(pcs_l, masks_l, flows_l, step_w=True, it=2000, aug_transform=True, sync=False, **kw)
    loss.backward()
    torch.cuda.synchronize()
    return {n: p.grad.detach().clone() for n, p in net.named_parameters() if p.grad is not None}, float(loss)


_lib.CALL_COUNTS = collections.Counter()
ref, l0 = grads_of_one_step()
calls = dict(_lib.CALL_COUNTS)
_lib.CALL_COUNTS = None
differ = collections.Counter()
worst = {}
losses = [l0]
for r in range(runs - 1):
    g, l = grads_of_one_step()
    losses.append(l)
    for n in ref:
        if not torch.equal(ref[n], g[n]):
            differ[n] += 1
            d = ((ref[n] - g[n]).norm() / ref[n].norm().clamp_min(1e-30)).item()
            worst[n] = max(worst.get(n, 0.0), d)
print("deterministic switch:", os.environ.get("OGC_DETERMINISTIC", "0"), " losses:", ["%.9g" % v for v in losses])
print("%d of %d parameter gradients differ between runs (%d runs)" % (len(differ), len(ref), runs))
for n in ref:
    if n in differ:
        print("   %-60s %d/%d  rel %.1e" % (n, differ[n], runs - 1, worst[n]))
h = hashlib.sha256()
for n in sorted(ref):
    h.update(ref[n].cpu().numpy().tobytes())
print("sha256 of the first run's gradients:", h.hexdigest())
print("entry points called:")
for k in sorted(calls):
    print("   %4d  %s" % (calls[k], k))
