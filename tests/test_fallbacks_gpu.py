"""Which fused paths does a training / inference step of each BASELINE configuration NOT take?

Every `*_available` gate of ogc_amd/fused.py that answers False sends its caller down the plain operator sequence — correct,
slower and silent.  One step of C2 / C4 / C5 (training) and one forward of C3 run here with the gates counted
(fused.GATE_MISSES); the set of gates that answered False at least once is pinned per configuration, so a layer shape that
quietly drops off a fused kernel (a changed channel count, a new LDS bound, a renamed wrapper) fails a test instead of only
costing time.  Reduced batch sizes: the gates look at channel counts, neighbourhood sizes and points per cloud, not at the batch.
"""
import os

import pytest
import torch
import yaml

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

# configuration -> {gate: times it answered False during one step} (observed on MI355X, and why)
EXPECTED = {
    # C4, segnet_kitti fp32: every gate answers yes since round 4 (the pooled tails of SA2, 128 channels out, and SA3, 256 out,
    # run as one node through the pooled forms of the plain weight- / input-gradient kernels: fused.WIDE_POOL_BACKWARD)
    "kittisf": {},
    # C5, the same network at 16384 points, one-frame loss
    "waymo": {},
    # C2 (bf16 operands): the two-level encoder of segnet_ogcdr ends in 128- and 256-channel tails as well; since round 5 both
    # run as one node under bf16 too (statistics + neighbourhood extremes in the epilogue of the bf16 tile kernel, the pooled
    # backward forms with fp32 operands).  norm_act_conv: the 256 -> 128 layer of the first feature-propagation module has more
    # input channels than the GEMM with the folded normalisation takes (K <= 160)
    "ogcdr": {"norm_act_conv_available": 1},
    # C1 (512 points, 2 samples): the deeper of the two wide tails has fewer than 1024 tiles of 64 positions
    "sapien": {"norm_act_conv_pool_available": 1, "norm_act_conv_available": 1},
}
# C3 forward: one set-abstraction block of FlowStep3D has an MLP the three-layer chain kernel does not cover
EXPECTED_FLOW = {"mlp_chain_pool_available": 1}


def _one_step(cfg_name, batch_size):
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    from ogc_amd.pointnet2 import pointnet2 as api
    from ogc_amd.train_seg import build_segnet
    from ogc_amd.train_step import build_criterion, make_optimizer, train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    cfg = yaml.safe_load(open(os.path.join(ROOT, "config", cfg_name + "_unsup_synthetic.yaml")))
    prev = api._native.get_matmul_precision()
    api._native.set_matmul_precision(cfg.get("matmul_precision", "fp32"))
    try:
        torch.manual_seed(cfg["random_seed"])
        net = build_segnet(cfg).cuda()
        single = cfg["dataset"] == "waymo"
        crit = build_criterion(cfg["loss"], single_frame=single)
        opt = make_optimizer(net.parameters(), lr=cfg["lr"])
        outdoor = cfg["dataset"] in ("kittisf", "waymo")
        batch = make_scene_batch(batch_size, cfg["segnet"]["n_point"], cfg["segnet"]["n_slot"], seed=1, outdoor=outdoor, aug=True,
                                 device="cuda")
        if single:
            batch = tuple(x[:, ::2].contiguous() for x in batch)
        fused.GATE_MISSES.clear()
        loss_dict, stepped = train_step(net, crit, opt, batch, 10 ** 6, True)
        torch.cuda.synchronize()
        assert stepped and all(v == v for v in loss_dict.values()), loss_dict
        return dict(fused.GATE_MISSES)
    finally:
        api._native.set_matmul_precision(prev)


@pytest.mark.parametrize("cfg_name,batch_size", [("kittisf", 1), ("waymo", 1), ("ogcdr", 2), ("sapien", 2)])
def test_training_step_takes_the_fused_paths(cfg_name, batch_size):
    misses = _one_step(cfg_name, batch_size)
    print("gate misses, %s: %r" % (cfg_name, misses))
    assert misses == EXPECTED[cfg_name], misses


def test_flowstep3d_forward_takes_the_fused_paths():
    import ogc_amd  # noqa: F401
    from ogc_amd import fused
    from ogc_amd.models.flownet_kitti import FlowStep3D
    torch.manual_seed(0)
    net = FlowStep3D(npoint=8192, loc_flow_nn=16, loc_flow_rad=1.5).cuda().eval()
    g = torch.Generator().manual_seed(3)
    pc1 = ((torch.rand(1, 8192, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda()
    pc2 = pc1 + 0.05 * torch.randn(1, 8192, 3, generator=g).cuda()
    fused.GATE_MISSES.clear()
    with torch.no_grad():
        net(pc1, pc2, pc1, pc2, iters=2)
    torch.cuda.synchronize()
    misses = dict(fused.GATE_MISSES)
    print("gate misses, flownet_kitti forward: %r" % (misses,))
    assert misses == EXPECTED_FLOW, misses
