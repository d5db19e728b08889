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


def test_waymo_single_frame_loss(cpu_ops):
    gc.run_waymo_loss("cpu")


@pytest.mark.parametrize("name,kw,N,B", gc.SEG_CASES, ids=[c[0] for c in gc.SEG_CASES])
def test_segnet_forward_backward(cpu_ops, name, kw, N, B):
    gc.run_segnet("cpu", name, kw, N, B)


@pytest.mark.parametrize("name,kw,N,iters", gc.FLOW_CASES, ids=[c[0] for c in gc.FLOW_CASES])
def test_flownet_forward_backward(cpu_ops, name, kw, N, iters):
    gc.run_flownet("cpu", name, kw, N, iters)
