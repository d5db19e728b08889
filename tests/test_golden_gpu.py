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
    gc.run_modules("cuda", rtol=1e-4, atol=1e-5)


def test_losses_and_oa_icp():
    gc.run_losses("cuda", rtol=1e-5, atol=1e-6)


def test_waymo_single_frame_loss():
    gc.run_waymo_loss("cuda", rtol=1e-5, atol=1e-6)


def test_vote_and_clustering_metrics():
    gc.run_vote("cuda", rtol=2e-3, atol=1e-5, corr_rtol=1e-2)


@pytest.mark.parametrize("name,kw,N,B", gc.SEG_CASES, ids=[c[0] for c in gc.SEG_CASES])
def test_segnet_forward_backward(name, kw, N, B):
    gc.run_segnet("cuda", name, kw, N, B, rtol=1e-3, atol=1e-5, grad_rtol=1e-2)


@pytest.mark.parametrize("name,kw,N,iters", gc.FLOW_CASES, ids=[c[0] for c in gc.FLOW_CASES])
def test_flownet_forward_backward(name, kw, N, iters):
    gc.run_flownet("cuda", name, kw, N, iters, rtol=1e-3, atol=1e-4, grad_rtol=2e-2)
