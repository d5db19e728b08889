"""HIP path vs the reference's Python evaluated in float64 (tests/golden/truth_f64.npz): every output and parameter
gradient of the six models and of the module scenarios must be as close to the exact result as the reference's own
fp32 run is (factor 2 + a floor of a few ulps), and every output within the north star's 1e-5 relative of the truth."""
import pytest
import torch

import golden_cases as gc

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _hip():
    assert torch.cuda.is_available()
    import ogc_amd  # noqa: F401
    torch.backends.cuda.matmul.allow_tf32 = False


def _report(budget, capsys):
    with capsys.disabled():
        print("\n" + budget.table())
        if getattr(budget, "own_cond_max", None) is not None:
            print("largest gradient change under 1-ulp noise on the plain products (own conditioning): %.2e" % budget.own_cond_max)
    budget.check()


def test_modules_vs_f64(capsys):
    _report(gc.truth_modules("cuda"), capsys)


@pytest.mark.parametrize("name,kw,N,B", gc.SEG_CASES, ids=[c[0] for c in gc.SEG_CASES])
def test_segnet_vs_f64(name, kw, N, B, capsys):
    _report(gc.truth_segnet("cuda", name, kw, N, B), capsys)


@pytest.mark.parametrize("name,kw,N,iters", gc.FLOW_CASES, ids=[c[0] for c in gc.FLOW_CASES])
def test_flownet_vs_f64(name, kw, N, iters, capsys):
    _report(gc.truth_flownet("cuda", name, kw, N, iters), capsys)


@pytest.mark.parametrize("name,npoint,n_level,feat_c,scale", gc.GCORR_CASES, ids=[c[0] for c in gc.GCORR_CASES])
def test_global_corr_layer(name, npoint, n_level, feat_c, scale, capsys):
    _report(gc.run_global_corr("cuda", name, npoint, n_level, feat_c, scale), capsys)
