"""Re-runs the scenarios of tests/golden/make_golden.py with THIS repo's host-side layers (ogc_amd.*) on a given
device and compares with the committed outputs of the reference's Python.  Shared by the CPU tests (operators =
the oracle, injected as ogc_amd.pointnet2.pointnet2._native) and the GPU tests (operators = HIP kernels)."""
import importlib
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import detgen  # noqa: E402

GOLD = os.path.join(HERE, "golden")


def load(name):
    return dict(np.load(os.path.join(GOLD, name + ".npz"), allow_pickle=False))


def close(got, want, rtol, atol, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = np.asarray(want)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, got.shape, want.shape)
    if not np.allclose(got, want, rtol=rtol, atol=atol, equal_nan=True):
        err = np.abs(got.astype(np.float64) - want.astype(np.float64))
        rel = err / (np.abs(want).astype(np.float64) + atol)
        raise AssertionError("%s: max abs err %.3e, max rel err %.3e (rtol %.1e atol %.1e)" %
                             (what, err.max(), rel.max(), rtol, atol))


def close_scaled(got, want, rel, what):
    """|got - want| <= rel * max|want| — for gradients of ill-conditioned terms (unit-direction vectors of tiny
    residuals amplify 1e-7 coordinate differences), where an element-wise relative test is meaningless."""
    want = np.asarray(want)
    close(got, want, 0.0, rel * float(np.abs(want).max()), what)


def exact(got, want, what):
    got = got.detach().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    assert np.array_equal(got, np.asarray(want)), "%s: integer/bit mismatch" % what


OUTPUT_REL_L2 = 1e-5   # north star: float outputs within 1e-5 relative of the reference's


def rel_l2_close(got, want, tol, what):
    """||got - want|| <= tol * ||want||: the bar for every float OUTPUT (masks, flows, features) against the fp32 fixtures."""
    got = got.detach().cpu().double().numpy() if torch.is_tensor(got) else np.asarray(got, np.float64)
    want = np.asarray(want, np.float64)
    assert got.shape == want.shape, "%s: shape %s vs %s" % (what, got.shape, want.shape)
    err = float(np.linalg.norm(got - want) / max(np.linalg.norm(want), 1e-300))
    assert err <= tol, "%s: relative L2 error %.2e > %.0e" % (what, err, tol)
    return err


def check_grads(module, loss, gold, prefix, rtol, atol):
    """Parameter gradients: 1e-5 where the tensor is that well conditioned; tensors beyond it must stay inside the conditioning
    budget `rtol` (tests/golden/make_truth_f64.py measured what 2-ulp noise does to single gradient tensors of these piecewise
    linear nets: 1e-4 .. 1e-2) and are listed, so that the budget's users are visible in the test log."""
    module.zero_grad()
    loss.backward()
    budget_users = []
    for name, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        want_norm = gold[prefix + "gnorm/" + name]
        close(g.norm().reshape(1), want_norm, rtol, atol * 10, prefix + "gnorm/" + name)
        scale = float(want_norm[0]) / max(np.sqrt(p.numel()), 1.0)
        close(g.flatten()[:32], gold[prefix + "ghead/" + name], rtol, atol + 10 * rtol * scale,
              prefix + "ghead/" + name)
        head = gold[prefix + "ghead/" + name].astype(np.float64)
        got = g.flatten()[:32].detach().cpu().double().numpy()
        err = float(np.linalg.norm(got - head) / max(np.linalg.norm(head), 1e-300)) if np.linalg.norm(head) > 1e-12 else 0.0
        nerr = abs(float(g.norm()) - float(want_norm[0])) / max(float(want_norm[0]), 1e-300) if float(want_norm[0]) > 1e-12 else 0.0
        if max(err, nerr) > OUTPUT_REL_L2:
            budget_users.append("%s%s (%.1e)" % (prefix, name, max(err, nerr)))
    if budget_users:
        print("gradient tensors beyond 1e-5, inside the conditioning budget %.0e: %d of %d — %s" %
              (rtol, len(budget_users), sum(1 for _ in module.parameters()), ", ".join(budget_users[:8]) + (" ..." if len(budget_users) > 8 else "")))


def run_operator_layer(dev, rtol=1e-6, atol=1e-7):
    from ogc_amd.pointnet2.pointnet2 import (QueryAndGroup, ball_query, furthest_point_sample, gather_operation,
                                             grouping_operation, knn, three_interpolate, three_nn)
    g = load("operator_layer")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    pc = detgen.cloud(2, 700, 11)
    pc[:, 300:350] = pc[:, :50]
    xyz = T(pc)
    fps_idx = furthest_point_sample(xyz, 128)
    exact(fps_idx, g["fps_idx"], "fps_idx")
    new_xyz = torch.gather(xyz, 1, fps_idx.long().unsqueeze(-1).expand(-1, -1, 3)).contiguous()
    dist, idx = knn(16, new_xyz, xyz)
    exact(idx, g["knn_idx"], "knn_idx")
    close(dist, g["knn_dist"], 1e-6, 0, "knn_dist")
    d3, i3 = three_nn(xyz, new_xyz)
    exact(i3, g["nn3_idx"], "nn3_idx")
    close(d3, g["nn3_dist"], 1e-6, 0, "nn3_dist")
    bq = ball_query(6.0, 24, xyz, new_xyz)
    exact(bq, g["ball_idx"], "ball_idx")
    feats = T(detgen.uniform((2, 7, 700), 12)).requires_grad_(True)
    nf, gx = QueryAndGroup(radius=5.0, nsample=16)(xyz, new_xyz, feats)
    exact(nf, g["qg_features"], "qg_features")
    exact(gx, g["qg_xyz"], "qg_xyz")
    gg = torch.autograd.grad(nf, feats, T(detgen.uniform(tuple(nf.shape), 13)))[0]
    close(gg, g["qg_grad"], 1e-5, 1e-5, "qg_grad")
    w = T(g["interp_w"])
    f128 = T(detgen.uniform((2, 7, 128), 14)).requires_grad_(True)
    interp = three_interpolate(f128, i3, w)
    exact(interp, g["interp"], "interp")
    gi = torch.autograd.grad(interp, f128, T(detgen.uniform(tuple(interp.shape), 15)))[0]
    close(gi, g["interp_grad"], 1e-5, 1e-5, "interp_grad")
    gathered = gather_operation(feats, fps_idx)
    exact(gathered, g["gathered"], "gathered")
    g2 = torch.autograd.grad(gathered, feats, T(detgen.uniform(tuple(gathered.shape), 16)))[0]
    close(g2, g["gather_grad"], 1e-5, 1e-5, "gather_grad")
    exact(grouping_operation(feats, bq), g["grouped_ball"], "grouped_ball")


def run_modules(dev, rtol=1e-5, atol=1e-6):
    from ogc_amd.utils.flowstep3d_util import FlowEmbedding, PointNetFeaturePropogation, PointNetSetAbstraction
    from ogc_amd.utils.pointnet2_util import PointnetFPModule, PointnetSAModuleMSG
    g = load("modules")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    bn = {"class": "GroupNorm", "num_groups": 4}
    pc = T(detgen.cloud(2, 512, 21, scale=(1, 1, 1)))
    feats = T(detgen.uniform((2, 3, 512), 22))
    sa = detgen.fill_module(PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4], nsamples=[16, 32],
                                                mlps=[[3, 16, 16], [3, 16, 32]], bn=bn), 1).to(dev)
    new_xyz, new_feats, inds = sa(pc, feats, return_inds=True)
    exact(inds, g["sa_inds"], "sa_inds")
    exact(new_xyz, g["sa_xyz"], "sa_xyz")
    close(new_feats, g["sa_feats"], rtol, atol, "sa_feats")
    fp = detgen.fill_module(PointnetFPModule(mlp=[48 + 3, 32, 16], bn=bn), 2).to(dev)
    close(fp(pc, new_xyz, feats, new_feats), g["fp_out"], rtol, atol, "fp_out")
    check_grads(sa, (new_feats ** 2).mean(), g, "sa_", 1e-4, 1e-7)

    xyz_t = pc.transpose(1, 2).contiguous()
    f3 = detgen.fill_module(PointNetSetAbstraction(npoint=128, radius=None, nsample=8, in_channel=3, mlp=[16, 32],
                                                   group_all=False, return_fps=True), 3).to(dev)
    nx, nf, fidx = f3(xyz_t, feats)
    exact(fidx, g["f3_fps"], "f3_fps")
    exact(nx, g["f3_xyz"], "f3_xyz")
    close(nf, g["f3_feats"], rtol, atol, "f3_feats")
    f3b = detgen.fill_module(PointNetSetAbstraction(npoint=128, radius=0.3, nsample=8, in_channel=32, mlp=[16],
                                                    group_all=False, use_act=False, mean_aggr=True), 4).to(dev)
    close(f3b(nx, nf)[1], g["f3b_feats"], rtol, atol, "f3b_feats")
    fpf = detgen.fill_module(PointNetFeaturePropogation(in_channel=32 + 3, mlp=[16]), 5).to(dev)
    close(fpf(xyz_t, nx, feats, nf), g["fpf_out"], rtol, atol, "fpf_out")
    pc_b = T(detgen.cloud(2, 128, 23, scale=(1, 1, 1))).transpose(1, 2).contiguous()
    fb = T(detgen.uniform((2, 32, 128), 24))
    fe = detgen.fill_module(FlowEmbedding(radius=0.5, nsample=8, in_channel=32, mlp=[32, 32]), 6).to(dev)
    _, corr = fe(nx, pc_b, nf.detach(), fb)
    close(corr, g["fe_out"], rtol, atol, "fe_out")
    check_grads(fe, (corr ** 2).mean(), g, "fe_", 1e-4, 1e-7)


def run_waymo_loss(dev, rtol=1e-5, atol=1e-6, grad_rel=5e-3):
    """UnsupervisedOGCLossSingleFrame against the class in the reference's train_seg_waymo.py:244-334 (fixture
    losses_waymo.npz, inputs shared with `losses`)."""
    from ogc_amd.losses.seg_loss_unsup import (DynamicLoss, EntropyLoss, InvarianceLoss, RankLoss, SmoothLoss,
                                               UnsupervisedOGCLossSingleFrame)
    g, gw = load("losses"), load("losses_waymo")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    smooth_params = {'w_knn': 3., 'w_ball_q': 1.,
                     'knn_loss_params': {'k': 8, 'radius': 0.1, 'cross_entropy': False, 'loss_norm': 1},
                     'ball_q_loss_params': {'k': 16, 'radius': 0.2, 'cross_entropy': False, 'loss_norm': 1}}
    crit = UnsupervisedOGCLossSingleFrame(DynamicLoss(loss_norm=2), SmoothLoss(**smooth_params), InvarianceLoss(loss_norm=2),
                                          EntropyLoss(), RankLoss(), weights=[10.0, 0.1, 0.1], start_steps=[0, 100, 0])
    keys = ('dynamic', 'smooth', 'invariance', 'entropy', 'rank', 'sum')
    for tag, aug, step_w, it in [("1v", False, False, 0), ("2v", True, True, 50), ("2v_gated", True, True, 500)]:
        sel = [0, 2][:2 if aug else 1]
        pcs = [T(g["pc%d" % v]) for v in sel]
        flows = [T(g["flow%d" % v]) for v in sel]
        masks = [T(g["mask%d" % v]).requires_grad_(True) for v in sel]
        loss, ld = crit(pcs, masks, flows, step_w=step_w, it=it, aug_transform=aug)
        assert tuple(ld.keys()) == keys
        gs = torch.autograd.grad(loss, masks)
        close(loss, gw["%s_loss" % tag], rtol, atol, "waymo_%s_loss" % tag)
        close(np.array([ld[k] for k in keys]), gw["%s_dict" % tag], rtol, atol, "waymo_%s_dict" % tag)
        for i, gr in enumerate(gs):
            close_scaled(gr, gw["%s_gmask%d" % (tag, i)], grad_rel, "waymo_%s_gmask%d" % (tag, i))


def run_vote(dev, rtol=1e-4, atol=1e-6, corr_rtol=1e-4):
    """Multi-frame voting and the clustering metrics against the reference's vote.py / metrics.seg_metric (fixture
    vote.npz: a 4-frame sequence, every frame predicting the objects in its own slot order).
    corr_rtol: the correspondence weights are softmax(-cdist / 0.01); torch.cdist's |a|^2 + |b|^2 - 2ab form leaves
    ~1e-7 absolute error on d^2, i.e. ~1e-5 on a centimetre distance and ~1e-3 on its logit, so two BLAS back ends
    (the fixture was made on the CPU) agree on individual weights to ~1e-3 relative only; sums over them (the carried
    and voted masks) agree far better."""
    from ogc_amd import vote
    from ogc_amd.metrics.seg_metric import ClusteringMetrics
    g = load("vote")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    pc, flows, mask = T(g["pc"]), T(g["flows"]), T(g["mask"])
    corrs = vote.collect_correspondences(pc, flows)
    assert set(corrs) == {"%d_%d" % (a, b) for a in range(4) for b in range(4)}
    for key in ("0_1", "2_0", "0_3", "3_1"):
        assert corrs[key].shape == (1, 200, 200)
        close(corrs[key][0], g["corr_" + key], corr_rtol, atol, "corr_" + key)
    # the chain-free path: (chained correspondence) @ X from the adjacent pairs only
    adj = vote._adjacent(pc, flows)
    for (t, v) in ((0, 3), (3, 1), (2, 0), (1, 2)):
        close(vote._carry(adj, t, v, mask[v]), (corrs["%d_%d" % (t, v)][0] @ mask[v]).cpu().numpy(), rtol, atol,
              "carry_%d_%d" % (t, v))
    close(vote.match_mask_by_cost(mask[0], mask[1], measure='ce'), g["matched_ce"], 0, 0, "matched_ce")
    close(vote.match_mask_by_cost(mask[0], mask[2], measure='iou'), g["matched_iou"], 0, 0, "matched_iou")
    for w in (1, 3):
        close(vote.mask_voting(pc, mask, flows, time_window_size=w), g["voted_w%d" % w], rtol, atol, "voted_w%d" % w)
    close(vote.vote_batch(torch.cat([pc, pc]), torch.cat([mask, mask]),
                          torch.cat([flows, flows[:1], flows, flows[:1]]), 4, 3),
          np.concatenate([g["voted_w3"], g["voted_w3"]]), rtol, atol, "vote_batch")
    segm = T(g["segm_small"])
    for thresh in (0, 30):
        res = ClusteringMetrics()(mask, segm, ignore_npoint_thresh=thresh)
        close(np.array(res["iou"], np.float64), g["cluster_iou_%d" % thresh], 1e-6, 1e-7, "cluster_iou_%d" % thresh)
        close(np.array(res["ri"], np.float64), g["cluster_ri_%d" % thresh], 1e-6, 1e-7, "cluster_ri_%d" % thresh)
    only_ri = ClusteringMetrics(spec=[ClusteringMetrics.RI])(mask, segm)
    assert list(only_ri) == ["ri"]


def run_losses(dev, rtol=1e-5, atol=1e-6, grad_rel=5e-3):
    from ogc_amd.losses.flow_loss_unsup import ChamferLoss, UnsupervisedFlowStep3DLoss
    from ogc_amd.losses.flow_loss_unsup import SmoothLoss as FlowSmoothLoss
    from ogc_amd.losses.seg_loss_unsup import (DynamicLoss, EntropyLoss, InvarianceLoss, RankLoss, SmoothLoss,
                                               UnsupervisedOGCLoss, fit_motion_svd_batch, interpolate_mask_by_flow,
                                               match_mask_by_iou)
    from ogc_amd.oa_icp import object_aware_icp, weighted_kabsch
    g = load("losses")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    B, N, K = 2, 512, 6
    pcs = [T(g["pc%d" % v]) for v in range(4)]
    flows = [T(g["flow%d" % v]) for v in range(4)]
    masks = [T(g["mask%d" % v]).requires_grad_(True) for v in range(4)]

    R, t = fit_motion_svd_batch(pcs[0], pcs[0] + flows[0], masks[0][..., 0].detach())
    close(R, g["svd_R"], rtol, 1e-5, "svd_R")
    close(t, g["svd_t"], rtol, 1e-5, "svd_t")
    R, t = fit_motion_svd_batch(pcs[0], pcs[0] + flows[0], None)
    close(R, g["svd_R_nomask"], rtol, 1e-5, "svd_R_nomask")
    close(t, g["svd_t_nomask"], rtol, 1e-5, "svd_t_nomask")
    zero_mask = masks[0][..., 0].detach().clone()
    zero_mask[1] = 0.0
    R, t = fit_motion_svd_batch(pcs[0], pcs[0] + flows[0], zero_mask)
    close(R, g["svd_R_zero"], rtol, 1e-5, "svd_R_zero")
    close(t, g["svd_t_zero"], rtol, 1e-5, "svd_t_zero")

    smooth_params = {'w_knn': 3., 'w_ball_q': 1.,
                     'knn_loss_params': {'k': 8, 'radius': 0.1, 'cross_entropy': False, 'loss_norm': 1},
                     'ball_q_loss_params': {'k': 16, 'radius': 0.2, 'cross_entropy': False, 'loss_norm': 1}}
    dyn, smooth, inv = DynamicLoss(loss_norm=2), SmoothLoss(**smooth_params), InvarianceLoss(loss_norm=2)
    ent, rank = EntropyLoss(), RankLoss()
    crit = UnsupervisedOGCLoss(dyn, smooth, inv, ent, rank, weights=[10.0, 0.1, 0.1], start_steps=[0, 100, 0])
    keys = ('dynamic', 'smooth', 'invariance', 'entropy', 'rank', 'sum')
    for tag, aug, step_w, it in [("2v", False, False, 0), ("4v", True, True, 50)]:
        nv = 4 if aug else 2
        loss, ld = crit(pcs[:nv], masks[:nv], flows[:nv], step_w=step_w, it=it, aug_transform=aug)
        assert tuple(ld.keys()) == keys
        gs = torch.autograd.grad(loss, masks[:nv])
        close(loss, g["ogc_%s_loss" % tag], rtol, atol, "ogc_%s_loss" % tag)
        close(np.array([ld[k] for k in keys]), g["ogc_%s_dict" % tag], rtol, atol, "ogc_%s_dict" % tag)
        for v in range(nv):
            close_scaled(gs[v], g["ogc_%s_gmask%d" % (tag, v)], grad_rel, "ogc_%s_gmask%d" % (tag, v))
    close(dyn(pcs[0], masks[0], flows[0]), g["dyn"], rtol, atol, "dyn")
    close(smooth.knn_loss(pcs[0], masks[0]), g["smooth_knn"], rtol, atol, "smooth_knn")
    close(smooth.ball_q_loss(pcs[0], masks[0]), g["smooth_ball"], rtol, atol, "smooth_ball")
    ce = SmoothLoss(3., 1., {'k': 8, 'radius': 0.1, 'cross_entropy': True}, {'k': 16, 'radius': 0.2, 'cross_entropy': True})
    close(ce(pcs[0], masks[0]), g["smooth_ce"], rtol, atol, "smooth_ce")
    close(interpolate_mask_by_flow(pcs[0], pcs[1], masks[0].detach(), flows[0], k=1), g["interp_mask_k1"], rtol, atol, "interp_mask_k1")
    close(interpolate_mask_by_flow(pcs[0], pcs[1], masks[0].detach(), flows[0], k=3), g["interp_mask_k3"], rtol, atol, "interp_mask_k3")
    exact(match_mask_by_iou(masks[0].detach(), masks[2].detach()), g["perm"], "perm")
    close(inv(masks[0], masks[2]), g["inv"], rtol, atol, "inv")
    close(ent(masks[0]), g["entropy"], rtol, atol, "entropy")
    close(rank(masks[0]), g["rank"], rtol, atol, "rank")

    fcrit = UnsupervisedFlowStep3DLoss(ChamferLoss(loss_norm=2),
                                       FlowSmoothLoss(3., 1., {'k': 4, 'radius': 0.05, 'loss_norm': 1},
                                                      {'k': 8, 'radius': 0.1, 'loss_norm': 1}),
                                       weights=[0.75, 0.25], iters_w=[0.5, 1.0])
    fp = [flows[0].clone().requires_grad_(True), (flows[0] * 0.9).clone().requires_grad_(True)]
    pc2 = T(g["flow_pc2"])
    floss, fd = fcrit(pcs[0], pc2, fp)
    gr = torch.autograd.grad(floss, fp)
    close(floss, g["flow_loss"], rtol, atol, "flow_loss")
    close(np.array([fd[k] for k in sorted(fd)]), g["flow_dict"], rtol, atol, "flow_dict")
    close_scaled(gr[0], g["flow_g0"], grad_rel, "flow_g0")
    close_scaled(gr[1], g["flow_g1"], grad_rel, "flow_g1")

    close(weighted_kabsch(pcs[0], flows[0], masks[0].detach()), g["kabsch_flow"], rtol, 1e-5, "kabsch_flow")
    icp = object_aware_icp(pcs[0], pc2, T(g["icp_noisy_flow"]), masks[0].detach(),
                           masks[0].detach()[:, :, [1, 0, 2, 3, 4, 5]], icp_iter=3, temperature=0.01)
    close(icp, g["icp_flow"], 1e-4, 1e-5, "icp_flow")


SEG_CASES = [("segnet_sapien", dict(n_slot=8, n_point=512, transformer_embed_dim=128), 512, 2),
             ("segnet_ogcdr", dict(n_slot=8, n_point=512, transformer_embed_dim=128), 512, 2),
             ("segnet_kitti", dict(n_slot=10, n_point=1024, transformer_embed_dim=128), 1024, 2)]
FLOW_CASES = [("flownet_sapien", dict(npoint=512, loc_flow_nn=8, loc_flow_rad=0.3), 512, 3),
              ("flownet_ogcdr", dict(npoint=512, loc_flow_nn=8, loc_flow_rad=0.3), 512, 2),
              ("flownet_kitti", dict(npoint=1024, loc_flow_nn=16, loc_flow_rad=1.5), 1024, 2)]


def run_segnet(dev, name, kw, N, B, rtol=1e-4, atol=1e-6, grad_rtol=1e-3):
    g = load("model_" + name)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mod = importlib.import_module("ogc_amd.models." + name)
    net = detgen.fill_module(mod.MaskFormer3D(**kw), 7).to(dev)
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["state_keys"]], "state_dict keys differ from the reference"
    scale = (60, 4, 80) if name == "segnet_kitti" else (1, 1, 1)
    pc = T(detgen.cloud(B, N, 41, scale=scale))
    mask = net(pc, pc)
    close(mask, g["mask"], rtol, atol, name + ".mask")
    rel_l2_close(mask, g["mask"], OUTPUT_REL_L2, name + ".mask")
    target = T(detgen.uniform(tuple(mask.shape), 42, 0.0, 1.0))
    check_grads(net, ((mask - target) ** 2).mean(), g, "", grad_rtol, 1e-8)


def run_flownet(dev, name, kw, N, iters, rtol=1e-4, atol=1e-5, grad_rtol=2e-3):
    g = load("model_" + name)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mod = importlib.import_module("ogc_amd.models." + name)
    net = detgen.fill_module(mod.FlowStep3D(**kw), 8).to(dev)
    net.eval()
    assert sorted(net.state_dict().keys()) == [str(k) for k in g["state_keys"]], "state_dict keys differ from the reference"
    scale = (60, 4, 80) if name == "flownet_kitti" else (1, 1, 1)
    pc1 = T(detgen.cloud(2, N, 51, scale=scale))
    pc2 = T(g["pc2"])
    preds = net(pc1, pc2, pc1, pc2, iters=iters)
    assert len(preds) == iters
    for i, p in enumerate(preds):
        close(p, g["flow%d" % i], rtol, atol, "%s.flow%d" % (name, i))
        rel_l2_close(p, g["flow%d" % i], OUTPUT_REL_L2, "%s.flow%d" % (name, i))
    check_grads(net, sum((p ** 2).mean() for p in preds), g, "", grad_rtol, 1e-8)


# ---------------------------------------------------------------------------------------------------------------------
# fp64 truths (tests/golden/make_truth_f64.py): "as close to the exact result as the reference's own fp32 arithmetic"

def rel_l2(a, truth, denom_floor=1e-300):
    a = (a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)).astype(np.float64)
    truth = np.asarray(truth, np.float64)
    assert a.shape == truth.shape, (a.shape, truth.shape)
    return float(np.linalg.norm(a - truth) / max(np.linalg.norm(truth), denom_floor))


class ErrorBudget:
    """Collects, per tensor, err(this repo vs f64 truth) next to err(reference fp32 vs f64 truth) and asserts

            err_ours <= max(factor * err_ref, factor * cond, floor)        (and err_ours <= cap when a cap is given)

    `cond` is the conditioning of that tensor at fp32 resolution: the largest distance to the truth of eight float64
    re-evaluations of the reference with 2-ulp relative noise on every parameter and layer output (make_truth_f64.py
    ::Fp32Noise).  The networks are piecewise linear (ReLU gates, max-pool winners, neighbour selections on warped
    coordinates); a gate that flips under that noise moves parameter gradients by 1e-4 .. 1e-2 relative while outputs
    move by < 1e-6 — measured, and it is what the old 1e-2 gradient tolerances were silently absorbing.  `floor` keeps
    a tensor on which the reference's fp32 run happened to land unusually close to the truth from failing the test."""

    def __init__(self, factor=2.0, gate_flips=(), gate_flip_cap=2e-3):
        self.factor, self.rows, self.bad = factor, [], []
        # tensors (by name) known to sit behind a ReLU / max-pool gate that flips inside the last bit of one fp32 product on
        # this fixture: they may miss the bound above, up to `gate_flip_cap`.  The list is explicit and its length pinned.
        self.gate_flips, self.gate_flip_cap, self.used_flips = frozenset(gate_flips), gate_flip_cap, set()

    def add(self, what, ours, ref32, truth, floor, cap=None, denom_floor=1e-300, cond=0.0):
        e_o, e_r = rel_l2(ours, truth, denom_floor), rel_l2(ref32, truth, denom_floor)
        ok = e_o <= max(self.factor * e_r, self.factor * cond, floor) and (cap is None or e_o <= cap)
        if not ok and what in self.gate_flips and e_o <= self.gate_flip_cap:
            self.used_flips.add(what)
            ok = True
        self.rows.append((what, e_o, e_r, ok))
        if not ok:
            self.bad.append("%s: ours %.3e vs reference-fp32 %.3e, conditioning %.3e (floor %.1e, cap %s)" %
                            (what, e_o, e_r, cond, floor, cap))

    def table(self, top=12):
        rows = sorted(self.rows, key=lambda r: -r[1] / max(r[2], 1e-12))[:top]
        return "\n".join("%-70s ours %.2e  ref32 %.2e  ratio %.2f" % (w, o, r, o / max(r, 1e-300)) for w, o, r, _ in rows)

    def check(self):
        assert not self.bad, "%d of %d tensors exceed the reference's own fp32 error:\n%s" % (
            len(self.bad), len(self.rows), "\n".join(self.bad[:20]))


def _cond(truth, key):
    v = truth.get("cond/" + key)
    return 0.0 if v is None else float(v[0])


# Gradient tensors allowed to miss the (reference-fp32 error, stored float64 conditioning, floor) bound, by fixture and NAME, up to
# ErrorBudget.gate_flip_cap = 2e-3 relative (worst measured: 1.24e-3).  Why they exist: on the segnet_ogcdr fixture a ReLU /
# max-pool gate of the first set-abstraction level sits between two fp32 roundings of one activation; the vendor GEMM's values moved
# by ONE ulp flip it and change the gradients of everything upstream and beside it — 68 of the model's 181 gradient rows — by
# ~1e-3 (own_grad_conditioning below measures that and the test prints it: a report, no part of the bound).  The forward output
# of the same fixture is held to 1e-5 like every other.  The list is data (tests/golden/gate_flip_tensors.json) and its length is
# pinned here: it may only shrink.
def _gate_flip_tensors():
    import json
    with open(os.path.join(GOLD, "gate_flip_tensors.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


GATE_FLIP_TENSORS = _gate_flip_tensors()   # tests/golden/gate_flip_tensors.json, written from tools/gate_flip_list.py's output
# Round 6: EMPTY.  The 68 rows of segnet_ogcdr came from ONE product: the grouped first layer of the first level applied its
# three feature columns per point (P = W_f f, gathered) and added W_xyz rel on top — two separately rounded halves where the
# reference's convolution (and this library's matrix kernels) run one chain over the six input channels.  That level now takes
# ogc_group_linear_fwd_direct (one k-ascending chain per output, csrc/gather_group.hip) and every gradient row of every fixture is
# inside the ordinary bound (tools/gate_flip_list.py: 0 / 0 / 0; with OGC_GROUP_LINEAR_DIRECT=0 the 68 rows come back).
MAX_GATE_FLIP_TENSORS = {"segnet_ogcdr": 0}
assert {k: len(v) for k, v in GATE_FLIP_TENSORS.items()} == MAX_GATE_FLIP_TENSORS, "the exception list may only shrink"


def own_grad_conditioning(run, samples=6):
    """(A report, not a tolerance.)  Conditioning of the parameter gradients of THIS implementation at fp32 resolution, measured: run() (a fresh forward
    and backward pass, returning {name: gradient}) is repeated with every output of fused._product — the plain matrix
    products of the feature-propagation layers and of the point-wise part of a set-abstraction layer — moved by ONE ulp up
    or down at random, and the largest relative change of each gradient's norm / 32-entry head is returned.  Why: those
    products moved from the vendor library to csrc/gemm_chunk.hip in round 4; both are equally close to the float64 product
    (2.9e-7 relative), they round differently, and on the segnet_ogcdr fixture a ReLU / max-pool gate sits inside that
    last bit (the library's values moved by one ulp at random give the same 1e-3 gradient change as the new kernel:
    tools/ measured in round 4, HISTORY.md).  The eight noise samples of make_truth_f64.py did not visit that gate, so the
    stored conditioning is blind to it; this one sees it.  An implementation error does NOT hide behind it: a wrong product
    changes the gradients whether or not the noise is on, and the noise runs then agree with each other."""
    from ogc_amd import fused
    base = run()
    orig = fused._product
    cond = {}
    try:
        for sd in range(samples):
            gen = torch.Generator(device="cuda").manual_seed(1000 + sd)

            def noisy(w2d, x3, transpose, out=None, _orig=orig, _gen=gen):
                v = _orig(w2d, x3, transpose, out=out)
                if not v.is_cuda:
                    return v
                up = torch.rand(v.shape, device=v.device, generator=_gen) < 0.5
                inf = torch.full_like(v, float("inf"))
                moved = torch.where(up, torch.nextafter(v, inf), torch.nextafter(v, -inf))
                if out is not None:
                    out.copy_(moved)
                    return out
                return moved
            fused._product = noisy
            try:
                g = run()
            finally:
                fused._product = orig
            for k, gb in base.items():
                nb = float(gb.norm())
                if nb == 0.0:
                    continue
                typical = nb * np.sqrt(min(32, gb.numel()) / gb.numel())
                hb, hn = gb.flatten()[:32], g[k].flatten()[:32]
                cond["gnorm/" + k] = max(cond.get("gnorm/" + k, 0.0), abs(float(g[k].norm()) - nb) / nb)
                cond["ghead/" + k] = max(cond.get("ghead/" + k, 0.0), float((hn - hb).norm()) / max(float(hb.norm()), typical))
    finally:
        fused._product = orig
    return cond


def _grad_rows(budget, module, loss, gold32, truth, prefix32, prefix64, floor):
    """Parameter gradients: the norm (relative) and the stored head of 32 entries (relative in L2 over the head).
    A flipped gate perturbs the gradient of EVERY parameter upstream of it, and which gates flip depends on the last
    bits of the forward pass (eight noise samples do not visit all of them): a tensor is therefore also allowed an
    eighth of the largest STORED conditioning (float64 re-evaluations of the reference, make_truth_f64.py) seen on any
    gradient tensor of the model.  Nothing measured on this implementation enters the bound."""
    module.zero_grad()
    loss.backward()
    model_cond = {kind: max([float(v[0]) for k, v in truth.items() if k.startswith("cond/" + prefix64 + kind)] + [0.0])
                  for kind in ("gnorm/", "ghead/")}
    for name, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        tn, th = truth[prefix64 + "gnorm/" + name], truth[prefix64 + "ghead/" + name]
        if float(tn[0]) == 0.0:
            assert float(g.norm()) == 0.0, name
            continue
        budget.add("gnorm/" + name, g.norm().reshape(1), gold32[prefix32 + "gnorm/" + name], tn, floor,
                   cond=max(_cond(truth, prefix64 + "gnorm/" + name), 0.125 * model_cond["gnorm/"]))
        # the stored head (first 32 entries): error relative to the head's norm, or to the share of the whole
        # gradient's norm 32 typical entries carry when the head happens to be a vanishing part of it
        typical = float(tn[0]) * np.sqrt(min(32, p.numel()) / p.numel())
        # (a head is 32 numbers: its own floor is 6x the norm's — still 300x below the 1e-2 these tests used to allow)
        budget.add("ghead/" + name, g.flatten()[:32], gold32[prefix32 + "ghead/" + name], th, 6 * floor, denom_floor=typical,
                   cond=max(_cond(truth, prefix64 + "ghead/" + name), 0.125 * model_cond["ghead/"]))


def truth_segnet(dev, name, kw, N, B, out_cap=1e-5, floor=1e-6, grad_floor=5e-6):
    g, t = load("model_" + name), load("truth_f64")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mod = importlib.import_module("ogc_amd.models." + name)
    net = detgen.fill_module(mod.MaskFormer3D(**kw), 7).to(dev)
    scale = (60, 4, 80) if name == "segnet_kitti" else (1, 1, 1)
    pc = T(detgen.cloud(B, N, 41, scale=scale))
    mask = net(pc, pc)
    budget = ErrorBudget(gate_flips=GATE_FLIP_TENSORS.get(name, ()))
    budget.add(name + ".mask", mask, g["mask"], t["model_%s/mask" % name], floor, cap=out_cap)
    target = T(detgen.uniform(tuple(mask.shape), 42, 0.0, 1.0))
    own_cond = None
    if str(dev).startswith("cuda"):
        def again():
            net.zero_grad()
            m = net(pc, pc)
            ((m - target) ** 2).mean().backward()
            return {k: (p.grad if p.grad is not None else torch.zeros_like(p)).detach().clone() for k, p in net.named_parameters()}
        own_cond = own_grad_conditioning(again)       # a REPORT (test_truth_f64_gpu prints it): no part of any bound
        budget.own_cond_max = max(own_cond.values()) if own_cond else 0.0
    _grad_rows(budget, net, ((mask - target) ** 2).mean(), g, t, "", "model_%s/" % name, grad_floor)
    return budget


def truth_flownet(dev, name, kw, N, iters, out_cap=1e-5, floor=1e-6, grad_floor=5e-6):
    g, t = load("model_" + name), load("truth_f64")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    mod = importlib.import_module("ogc_amd.models." + name)
    net = detgen.fill_module(mod.FlowStep3D(**kw), 8).to(dev)
    net.eval()
    scale = (60, 4, 80) if name == "flownet_kitti" else (1, 1, 1)
    pc1, pc2 = T(detgen.cloud(2, N, 51, scale=scale)), T(g["pc2"])
    preds = net(pc1, pc2, pc1, pc2, iters=iters)
    budget = ErrorBudget()
    for i, p in enumerate(preds):
        budget.add("%s.flow%d" % (name, i), p, g["flow%d" % i], t["model_%s/flow%d" % (name, i)], floor, cap=out_cap)
    _grad_rows(budget, net, sum((p ** 2).mean() for p in preds), g, t, "", "model_%s/" % name, grad_floor)
    return budget


def truth_modules(dev, floor=5e-7, grad_floor=5e-6):
    """The module scenarios of run_modules against their float64 evaluation."""
    from ogc_amd.utils.flowstep3d_util import FlowEmbedding, PointNetFeaturePropogation, PointNetSetAbstraction
    from ogc_amd.utils.pointnet2_util import PointnetFPModule, PointnetSAModuleMSG
    g, t = load("modules"), load("truth_f64")
    tt = lambda k: t["modules/" + k]  # noqa: E731
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    bn = {"class": "GroupNorm", "num_groups": 4}
    pc = T(detgen.cloud(2, 512, 21, scale=(1, 1, 1)))
    feats = T(detgen.uniform((2, 3, 512), 22))
    budget = ErrorBudget()
    sa = detgen.fill_module(PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4], nsamples=[16, 32],
                                                mlps=[[3, 16, 16], [3, 16, 32]], bn=bn), 1).to(dev)
    new_xyz, new_feats, inds = sa(pc, feats, return_inds=True)
    exact(inds, tt("sa_inds"), "sa_inds (f64 run selects the same points)")
    budget.add("sa_feats", new_feats, g["sa_feats"], tt("sa_feats"), floor, cap=1e-5)
    fp = detgen.fill_module(PointnetFPModule(mlp=[48 + 3, 32, 16], bn=bn), 2).to(dev)
    budget.add("fp_out", fp(pc, new_xyz, feats, new_feats), g["fp_out"], tt("fp_out"), floor, cap=1e-5)
    _grad_rows(budget, sa, (new_feats ** 2).mean(), g, {k[len("modules/"):]: v for k, v in t.items() if k.startswith("modules/")},
               "sa_", "sa_", grad_floor)
    xyz_t = pc.transpose(1, 2).contiguous()
    f3 = detgen.fill_module(PointNetSetAbstraction(npoint=128, radius=None, nsample=8, in_channel=3, mlp=[16, 32],
                                                   group_all=False, return_fps=True), 3).to(dev)
    nx, nf, fidx = f3(xyz_t, feats)
    budget.add("f3_feats", nf, g["f3_feats"], tt("f3_feats"), floor, cap=1e-5)
    f3b = detgen.fill_module(PointNetSetAbstraction(npoint=128, radius=0.3, nsample=8, in_channel=32, mlp=[16],
                                                    group_all=False, use_act=False, mean_aggr=True), 4).to(dev)
    budget.add("f3b_feats", f3b(nx, nf)[1], g["f3b_feats"], tt("f3b_feats"), floor, cap=1e-5)
    fpf = detgen.fill_module(PointNetFeaturePropogation(in_channel=32 + 3, mlp=[16]), 5).to(dev)
    budget.add("fpf_out", fpf(xyz_t, nx, feats, nf), g["fpf_out"], tt("fpf_out"), floor, cap=1e-5)
    pc_b = T(detgen.cloud(2, 128, 23, scale=(1, 1, 1))).transpose(1, 2).contiguous()
    fb = T(detgen.uniform((2, 32, 128), 24))
    fe = detgen.fill_module(FlowEmbedding(radius=0.5, nsample=8, in_channel=32, mlp=[32, 32]), 6).to(dev)
    _, corr = fe(nx, pc_b, nf.detach(), fb)
    budget.add("fe_out", corr, g["fe_out"], tt("fe_out"), floor, cap=1e-5)
    _grad_rows(budget, fe, (corr ** 2).mean(), g, {k[len("modules/"):]: v for k, v in t.items() if k.startswith("modules/")},
               "fe_", "fe_", grad_floor)
    return budget


GCORR_CASES = [("flownet_kitti", 1024, 4, 256, (60, 4, 80)), ("flownet_sapien", 512, 3, 256, (1, 1, 1)),
               ("flownet_ogcdr", 512, 3, 128, (1, 1, 1))]


def gcorr_inputs(npoint, n_level, feat_c, scale):
    """Same construction as tests/golden/make_truth_f64.py::gcorr_inputs (integer hashes only)."""
    B = 2
    pc1 = detgen.cloud(B, npoint // 4, 71, scale=scale)
    pc2 = pc1 + detgen.uniform((B, npoint // 4, 3), 72, -0.3, 0.3) * np.asarray(scale, np.float32) / 20
    pc2 = np.ascontiguousarray(pc2[:, ::-1])
    lv1, lv2 = [pc1], [pc2]
    for _ in range(n_level - 1):
        lv1.append(np.ascontiguousarray(lv1[-1][:, ::2]))
        lv2.append(np.ascontiguousarray(lv2[-1][:, ::2]))
    n_top = lv1[-1].shape[1]
    return lv1, lv2, detgen.uniform((B, feat_c, n_top), 73), detgen.uniform((B, feat_c, n_top), 74)


def run_global_corr(dev, name, npoint, n_level, feat_c, scale, rtol=1e-5, atol=1e-6):
    """GlobalCorrLayer alone (reference models/flownet_kitti.py:41-81 and twins; fixture global_corr.npz made by the
    reference's class, fp64 twin in truth_f64.npz): correlation matrix, upsampled correlation features and the
    gradients w.r.t. both feature maps and epsilon."""
    g, t = load("global_corr"), load("truth_f64")
    mod = importlib.import_module("ogc_amd.models." + name)
    lv1, lv2, f1n, f2n = gcorr_inputs(npoint, n_level, feat_c, scale)
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    layer = detgen.fill_module(mod.GlobalCorrLayer(npoint, False, mod.CONFIG), 9).to(dev).eval()
    P1 = [T(a).transpose(1, 2).contiguous() for a in lv1]
    P2 = [T(a).transpose(1, 2).contiguous() for a in lv2]
    f1, f2 = T(f1n).requires_grad_(True), T(f2n).requires_grad_(True)
    feats = layer(P1, P2, f1, f2)
    corr = layer.calc_corr_mat(P1[-1].permute(0, 2, 1), P2[-1].permute(0, 2, 1), f1.permute(0, 2, 1), f2.permute(0, 2, 1))
    tgt = T(detgen.uniform(tuple(feats.shape), 75))
    g1, g2, ge = torch.autograd.grad((feats * tgt).sum(), [f1, f2, layer.epsilon])
    key = "gcorr/%s/" % name
    budget = ErrorBudget()
    for k, v in (("feats", feats), ("corr", corr), ("g_f1", g1), ("g_f2", g2), ("g_eps", ge.reshape(1))):
        # epsilon's gradient is ONE number summed over B * n * n terms of mixed sign: two fp32 evaluations agree to 1e-4
        close(v, g[key + k], 1e-4 if k == "g_eps" else rtol, atol * max(1.0, float(np.abs(g[key + k]).max())), key + k)
        if k in ("feats", "corr"):
            budget.add(key + k, v, g[key + k], t[key + k], 1e-6, cap=1e-5)
        else:   # gradients; epsilon's is a single cancelling sum (the reference's own fp32 run is 8e-6 off on it)
            budget.add(key + k, v, g[key + k], t[key + k], 1e-4 if k == "g_eps" else 5e-6)
    return budget


def run_data_ops(dev):
    """fps_downsample / upsample_feat (reference utils/data_util.py:8-38; fixture data_ops.npz made by its functions)."""
    from ogc_amd.utils.data_util import fps_downsample, upsample_feat
    g = load("data_ops")
    pc = detgen.cloud(1, 3000, 91)[0]
    pc[100:140] = pc[:40]
    idx = fps_downsample(pc, n_sample_point=512, device=dev)
    assert isinstance(idx, np.ndarray) and idx.shape == (512,)
    exact(idx, g["fps_idx"], "fps_downsample")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    pcs = T(detgen.cloud(2, 1500, 92))
    sub = pcs[:, ::5].contiguous()
    sub[:, :7] = pcs[:, :7]
    feat = T(detgen.uniform((2, 300, 6), 93))
    up = upsample_feat(pcs, sub, feat)
    assert up.shape == (2, 1500, 6)
    close(up, g["up_feat"], 1e-5, 1e-6, "upsample_feat")


def run_group_all(dev):
    """GroupAll in its three modes and a FlowStep3D set-abstraction layer with group_all=True against the reference's classes
    (tests/golden/group_all.npz).  Pure re-arrangements are exact; the layer's features within 1e-5."""
    from ogc_amd.pointnet2.pointnet2 import GroupAll
    from ogc_amd.utils.flowstep3d_util import PointNetSetAbstraction
    g = load("group_all")
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)  # noqa: E731
    xyz, feats = T(detgen.cloud(2, 40, 95, scale=(1, 1, 1))), T(detgen.uniform((2, 5, 40), 96))
    for tag, use_xyz, f in (("xyz_feats", True, feats), ("feats", False, feats), ("xyz", True, None)):
        nf, gx = GroupAll(use_xyz)(xyz, None, f)
        exact(nf, g[tag], "GroupAll " + tag)
        exact(gx, g[tag + "_grouped_xyz"], "GroupAll grouped_xyz " + tag)
    sa = detgen.fill_module(PointNetSetAbstraction(npoint=None, radius=None, nsample=None, in_channel=5, mlp=[8, 8],
                                                   group_all=True), 9).to(dev)
    new_xyz, new_points = sa(xyz.transpose(1, 2).contiguous(), feats)
    exact(new_xyz, g["sa_new_xyz"], "group_all new_xyz")
    close(new_points, g["sa_new_points"], 1e-5, 1e-6, "group_all features")
