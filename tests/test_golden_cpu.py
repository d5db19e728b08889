"""CPU tests: this repo's host-side layers (operator API, SA/FP/FlowEmbedding modules, losses, OA-ICP, the six
models) against outputs of the REFERENCE's Python (tests/golden/*.npz).  The native operators are replaced by the
CPU oracle for these tests only (the product has no CPU path)."""
import pytest
import torch

import golden_cases as gc


@pytest.fixture()
def cpu_ops(monkeypatch, oracle):
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    torch.manual_seed(0)
    return api


def test_operator_layer(cpu_ops):
    gc.run_operator_layer("cpu")


def test_modules(cpu_ops):
    gc.run_modules("cpu")


def test_losses_and_oa_icp(cpu_ops):
    gc.run_losses("cpu")


def test_data_ops(cpu_ops):
    gc.run_data_ops("cpu")


def test_group_all(cpu_ops):
    gc.run_group_all("cpu")


@pytest.mark.parametrize("name,npoint,n_level,feat_c,scale", gc.GCORR_CASES, ids=[c[0] for c in gc.GCORR_CASES])
def test_global_corr_layer(cpu_ops, name, npoint, n_level, feat_c, scale):
    gc.run_global_corr("cpu", name, npoint, n_level, feat_c, scale).check()


def test_data_util_helpers():
    """augment_transform (same numpy seed -> same augmentations) and the label helpers vs the reference's own."""
    import numpy as np
    from ogc_amd.utils.data_util import augment_transform, compress_label_id, segm_to_mask
    g = gc.load("data_util")
    args_seg = {'scale_low': 0.95, 'scale_high': 1.05, 'degree_range': [0, 180, 0], 'shift_range': [0, 0.1, 0.2]}
    args_flow = dict(args_seg, degree_range=[5, 10, 15], aug_pc2={'degree_range': [1, 2, 3], 'shift_range': [0.01, 0.02, 0.03]})
    for tag, args, nv in (("seg", args_seg, 2), ("flow", args_flow, 3)):
        np.random.seed(1234)
        a, b = augment_transform(g["pcs"], g["flows"], args, n_view=nv)
        np.testing.assert_allclose(a, g["aug_%s_pcs" % tag], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(b, g["aug_%s_flows" % tag], rtol=1e-12, atol=1e-12)
        a2, _ = augment_transform(g["pcs"], g["flows"], args, n_view=nv, rng=np.random.RandomState(1234))
        np.testing.assert_allclose(a2, a, rtol=0, atol=0)
    np.testing.assert_array_equal(compress_label_id(g["segm"]), g["segm_cpr"])
    np.testing.assert_array_equal(segm_to_mask(g["segm"]), g["segm_mask"])
    np.testing.assert_array_equal(segm_to_mask(g["segm"], 8), g["segm_mask8"])
    from ogc_amd.metrics.flow_metric import epe_metric, eval_flow
    import torch
    gt, pred = torch.from_numpy(g["metric_gt"]), torch.from_numpy(g["metric_pred"])
    np.testing.assert_allclose(eval_flow(gt, pred, 0.05), g["metric_005"], rtol=1e-6)
    np.testing.assert_allclose(eval_flow(gt, pred, 0.01), g["metric_001"], rtol=1e-6)
    assert abs(epe_metric(gt, [pred, gt])["epe3d_#0"] - g["metric_005"][0]) < 1e-7 and epe_metric(gt, [pred, gt])["epe3d_#1"] == 0


def test_seg_metrics():
    """Device-style accumulate_eval_results + AP / PQ / F1 against the reference's numpy implementation."""
    import numpy as np
    import torch
    from ogc_amd.metrics.seg_metric import accumulate_eval_results, calculate_AP, calculate_PQ_F1
    g = gc.load("data_util")
    segm, mask = torch.from_numpy(g["seg_segm"]), torch.from_numpy(g["seg_mask"])
    for thresh in (0, 40):
        iou, matched, conf, n_gt = accumulate_eval_results(segm, mask, ignore_npoint_thresh=thresh)
        assert n_gt == int(g["seg_ngt_%d" % thresh][0])
        np.testing.assert_allclose(iou, g["seg_iou_%d" % thresh], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(matched, g["seg_matched_%d" % thresh])
        np.testing.assert_allclose(conf, g["seg_conf_%d" % thresh], rtol=1e-6)
        np.testing.assert_allclose(calculate_AP(matched, conf, n_gt), g["seg_ap_%d" % thresh][0], rtol=1e-12)
        np.testing.assert_allclose(calculate_PQ_F1(iou, matched, n_gt), g["seg_pqf1_%d" % thresh], rtol=1e-12)


def test_waymo_single_frame_loss(cpu_ops):
    gc.run_waymo_loss("cpu")


def test_vote_and_clustering_metrics(cpu_ops):
    gc.run_vote("cpu")


@pytest.mark.parametrize("name,kw,N,B", gc.SEG_CASES, ids=[c[0] for c in gc.SEG_CASES])
def test_segnet_forward_backward(cpu_ops, name, kw, N, B):
    gc.run_segnet("cpu", name, kw, N, B)


@pytest.mark.parametrize("name,kw,N,iters", gc.FLOW_CASES, ids=[c[0] for c in gc.FLOW_CASES])
def test_flownet_forward_backward(cpu_ops, name, kw, N, iters):
    gc.run_flownet("cpu", name, kw, N, iters)
