"""ogc_conv1x1_gemm_any (csrc/gemm_chunk.hip): OUT[b] = A . IN[b] for any reduction length and row count, both orientations,
against a float64 product — every launch shape (few positions: wavefronts split the rows, 16 or 32 rows each; many positions:
1 .. 8 row blocks per wavefront), ragged rows and reduction lengths, and the module-level routes of fused.py that use it.
Reference: the Conv1d / Conv2d of SharedMLP, utils/nn_util.py:45-85."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [  # (B, M, K, hw)
    (2, 256, 448, 256), (2, 128, 256, 256), (2, 448, 256, 256), (2, 131, 128, 512), (2, 64, 128, 512),
    (16, 128, 384, 1024), (16, 384, 128, 1024), (16, 64, 224, 2048), (16, 224, 64, 2048), (16, 67, 64, 8192),
    (16, 32, 3, 8192), (3, 1, 1, 64), (1, 17, 33, 64), (2, 200, 161, 128), (5, 96, 64, 2048), (1, 130, 37, 4096),
    (16, 128, 256, 16384), (4, 67, 64, 65536), (4, 33, 100, 65536), (4, 48, 40, 65600 - 64), (8, 224, 64, 32768),
    (4, 100, 200, 65536), (4, 16, 16, 65536), (4, 81, 9, 65536), (2, 300, 50, 131072),
]


@pytest.mark.parametrize("transpose", [0, 1])
@pytest.mark.parametrize("B,M,K,hw", SHAPES)
def test_gemm_any_against_float64(B, M, K, hw, transpose):
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator(device="cuda").manual_seed(B * 7 + M * 3 + K + hw + transpose)
    w = torch.randn((K, M) if transpose else (M, K), device="cuda", generator=g)
    x = torch.randn(B, K, hw, device="cuda", generator=g)
    out = torch.full((B, M, hw), float("nan"), device="cuda")
    nat.conv1x1_gemm_any_wrapper(B, M, K, hw, transpose, w, x, out)
    A = (w.t() if transpose else w).double()
    for b in sorted({0, B - 1}):
        ref = A @ x[b].double()
        err = float((out[b].double() - ref).abs().max())
        # fp32 products summed in fp32: a few ulp of the largest partial sum
        assert err <= 4e-6 * float(ref.abs().max()) * max(1.0, (K / 64) ** 0.5), (b, err, float(ref.abs().max()))
    assert bool(torch.isfinite(out).all())


def test_pointwise_conv_routes_match_the_library():
    """The module-level routes (fused._PointwiseConv on few positions / wide reductions, the point-wise part of the grouped first
    layer) with this kernel and with the vendor library: forward values and gradients to fp32 rounding."""
    from ogc_amd import fused
    torch.manual_seed(3)
    for (B, cin, cout, n) in [(2, 448, 256, 256), (16, 384, 128, 1024), (2, 64, 64, 512), (4, 224, 64, 2048)]:
        conv = torch.nn.Conv1d(cin, cout, 1, bias=False).cuda()
        x = torch.randn(B, cin, n, device="cuda", requires_grad=True)
        gy = torch.randn(B, cout, n, device="cuda")
        res = []
        for lib in (False, True):
            fused.LIBRARY_GEMMS = lib
            try:
                conv.weight.grad = None
                x.grad = None
                y = fused.pointwise_conv(x, conv)
                y.backward(gy)
                res.append((y.detach().clone(), x.grad.clone(), conv.weight.grad.clone()))
            finally:
                fused.LIBRARY_GEMMS = False
        for a, b in zip(*res):
            scale = float(b.abs().max())
            assert float((a - b).abs().max()) <= 2e-5 * scale, (cin, cout, n, float((a - b).abs().max()), scale)


@pytest.mark.parametrize("B,M,K,hw", [(2, 40, 67, 256), (4, 100, 131, 65536), (2, 130, 259, 4096), (4, 16, 9, 65536)])
def test_ragged_reduction_does_not_spread_non_finite_values(B, M, K, hw):
    """K not a multiple of the chunk: the kernel reads the rows beyond K from row K - 1.  They must count as ZEROS, not as
    "row K - 1 times a zero weight": an Inf in the last real channel gives Inf (not Inf + 0 * Inf = NaN) in the output rows
    that use the channel, as in any IEEE product of the real terms."""
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator(device="cuda").manual_seed(K)
    w = torch.randn(M, K, device="cuda", generator=g)
    w[::2, K - 1] = 0.0                                  # every other output row ignores the last channel
    x = torch.randn(B, K, hw, device="cuda", generator=g)
    x[:, K - 1, 5] = float("inf")
    x[:, K - 1, 70] = float("nan")
    for transpose in (0, 1):
        a = w.t().contiguous() if transpose else w
        out = torch.zeros(B, M, hw, device="cuda")
        nat.conv1x1_gemm_any_wrapper(B, M, K, hw, transpose, a, x, out)
        # (rows that ignore the channel hold 0 * Inf = NaN at that position in any product: not checked)
        ref = torch.matmul(w, x)
        clean = torch.ones(hw, dtype=torch.bool, device="cuda")
        clean[5] = clean[70] = False
        assert bool(torch.isfinite(out[:, :, clean]).all())
        torch.testing.assert_close(out[:, :, clean], ref[:, :, clean], rtol=1e-4, atol=1e-4)
        assert bool(torch.isinf(out[:, 1::2, 5]).all()) and bool(torch.isnan(out[:, 1::2, 70]).all())
