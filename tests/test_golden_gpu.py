"""GPU tests: the same scenarios on the MI355X with the HIP operators, against the reference-Python goldens."""
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip():
    assert torch.cuda.is_available()
    import ogc_amd  # noqa: F401
    torch.manual_seed(0)
    torch.backends.cuda.matmul.allow_tf32 = False


def test_operator_layer():
    gc.run_operator_layer("cuda")


def test_modules():
    gc.run_modules("cuda", rtol=1e-4, atol=1e-5)  # element-wise vs ONE fp32 run; vs the exact result: test_truth_f64_gpu.py


def test_losses_and_oa_icp():
    gc.run_losses("cuda", rtol=1e-5, atol=1e-6)


def test_waymo_single_frame_loss():
    gc.run_waymo_loss("cuda", rtol=1e-5, atol=1e-6)


def test_data_ops():
    gc.run_data_ops("cuda")


def test_group_all():
    gc.run_group_all("cuda")


def test_seg_and_flow_metrics_on_device():
    """accumulate_eval_results / AP / PQ / F1 and eval_flow with the tensors on the GPU, against the reference's numpy
    implementations (fixture data_util.npz; same assertions as tests/test_golden_cpu.py)."""
    import numpy as np
    from ogc_amd.metrics.flow_metric import epe_metric, eval_flow
    from ogc_amd.metrics.seg_metric import accumulate_eval_results, calculate_AP, calculate_PQ_F1
    g = gc.load("data_util")
    segm, mask = torch.from_numpy(g["seg_segm"]).cuda(), torch.from_numpy(g["seg_mask"]).cuda()
    for thresh in (0, 40):
        iou, matched, conf, n_gt = accumulate_eval_results(segm, mask, ignore_npoint_thresh=thresh)
        assert n_gt == int(g["seg_ngt_%d" % thresh][0])
        np.testing.assert_allclose(iou, g["seg_iou_%d" % thresh], rtol=1e-12, atol=1e-12)
        np.testing.assert_array_equal(matched, g["seg_matched_%d" % thresh])
        np.testing.assert_allclose(conf, g["seg_conf_%d" % thresh], rtol=1e-6)
        np.testing.assert_allclose(calculate_AP(matched, conf, n_gt), g["seg_ap_%d" % thresh][0], rtol=1e-12)
        np.testing.assert_allclose(calculate_PQ_F1(iou, matched, n_gt), g["seg_pqf1_%d" % thresh], rtol=1e-12)
    gt, pred = torch.from_numpy(g["metric_gt"]).cuda(), torch.from_numpy(g["metric_pred"]).cuda()
    np.testing.assert_allclose(eval_flow(gt, pred, 0.05), g["metric_005"], rtol=1e-6)
    np.testing.assert_allclose(eval_flow(gt, pred, 0.01), g["metric_001"], rtol=1e-6)
    assert abs(epe_metric(gt, [pred, gt])["epe3d_#0"] - g["metric_005"][0]) < 1e-7


def test_vote_and_clustering_metrics():
    gc.run_vote("cuda", rtol=2e-3, atol=1e-5, corr_rtol=1e-2)


@pytest.mark.parametrize("name,kw,N,B", gc.SEG_CASES, ids=[c[0] for c in gc.SEG_CASES])
def test_segnet_forward_backward(name, kw, N, B):
    # closeness to ONE fp32 evaluation of the reference; how close both are to the exact result, and why single gradient
    # tensors can differ by 1e-3 (gate flips), is tests/test_truth_f64_gpu.py
    gc.run_segnet("cuda", name, kw, N, B, rtol=2e-4, atol=2e-6, grad_rtol=2e-3)


@pytest.mark.parametrize("name,kw,N,iters", gc.FLOW_CASES, ids=[c[0] for c in gc.FLOW_CASES])
def test_flownet_forward_backward(name, kw, N, iters):
    gc.run_flownet("cuda", name, kw, N, iters, rtol=1e-4, atol=1e-5, grad_rtol=2e-3)
