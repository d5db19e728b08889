"""The deterministic-gradient mode (include/ogc_ops.h: ogc_set_deterministic; OGC_DETERMINISTIC=1) — SURVEY.md §5 "race detection":
the reference's gradient atomics (pointnet2/src/group_points_gpu.cu:24, interpolate_gpu.cu:211-213, sampling_gpu.cu:62) and this
library's fast kernels sum in an order that changes from launch to launch.  With the mode on:

  * every converted entry point gives the SAME BITS on every call, and the same result as the atomic kernels / the CPU oracle up
    to summation order (1e-5 relative);
  * the parameter gradients of a whole training step — segnet_kitti (OGC loss) and flownet_sapien (FlowStep3D loss) — have the
    same SHA-256 in two separate processes (tools/det_probe.py run as a child).
"""
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture
def det():
    from ogc_amd import _lib
    before = _lib.DETERMINISTIC
    _lib.set_deterministic(True)
    yield _lib
    _lib.set_deterministic(before)


def _both_modes(fn, repeats=3):
    """fn() -> tensor(s).  (results of `repeats` deterministic calls, result of one call of the atomic kernels)."""
    from ogc_amd import _lib
    before = _lib.DETERMINISTIC
    try:
        _lib.set_deterministic(False)
        fast = fn()
        _lib.set_deterministic(True)
        assert _lib.load().ogc_get_deterministic() == 1
        outs = [fn() for _ in range(repeats)]
    finally:
        _lib.set_deterministic(before)
    torch.cuda.synchronize()
    return outs, fast


def _as_list(x):
    return list(x) if isinstance(x, (tuple, list)) else [x]


def _check(outs, fast, rel=1e-5, what=""):
    first = _as_list(outs[0])
    for other in outs[1:]:
        for a, b in zip(first, _as_list(other)):
            assert torch.equal(a, b), "%s: two deterministic calls differ" % what
    for a, f in zip(first, _as_list(fast)):
        err = (a.double() - f.double()).norm() / f.double().norm().clamp_min(1e-30)
        assert err <= rel, "%s: deterministic vs atomic kernels rel L2 %.2e" % (what, err.item())


def _neighbours(b, npoint, nsample, n, g, padded=True):
    """Index rows as the searches leave them: some real neighbours, then a run of one padding index."""
    idx = torch.randint(0, n, (b, npoint, nsample), generator=g, dtype=torch.int32)
    if padded:
        keep = torch.randint(1, nsample + 1, (b, npoint, 1), generator=g)
        col = torch.arange(nsample).view(1, 1, -1)
        idx = torch.where(col < keep, idx, idx[:, :, :1].expand(-1, -1, nsample))
    return idx.contiguous()


@pytest.mark.parametrize("b,c,n,npoint,nsample", [(2, 7, 300, 96, 5), (2, 32, 2048, 512, 32), (1, 3, 17, 5, 3), (2, 16, 4096, 1024, 16)])
def test_group_points_grad(oracle, b, c, n, npoint, nsample):
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(b * 1000 + n)
    idx = _neighbours(b, npoint, nsample, n, g)
    go = torch.randn(b, c, npoint, nsample, generator=g)
    idx_d, go_d = idx.cuda(), go.cuda()

    def run():
        gp = torch.zeros(b, c, n, device="cuda")
        nat.group_points_grad_wrapper(b, c, n, npoint, nsample, go_d, idx_d, gp)
        return gp

    outs, fast = _both_modes(run)
    _check(outs, fast, what="group_points_grad")
    want = oracle.group_grad(go.numpy(), idx.numpy(), n)
    assert np.abs(outs[0].cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())


@pytest.mark.parametrize("b,c,n,m", [(2, 5, 100, 40), (2, 64, 2048, 512), (1, 128, 4096, 1024)])
def test_three_interpolate_and_gather_grads(oracle, b, c, n, m):
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(n + m)
    idx = torch.randint(0, m, (b, n, 3), generator=g, dtype=torch.int32)
    w = torch.rand(b, n, 3, generator=g)
    w = (w / w.sum(-1, keepdim=True)).contiguous()
    go = torch.randn(b, c, n, generator=g)
    idx_d, w_d, go_d = idx.cuda(), w.cuda(), go.cuda()

    def interp():
        gp = torch.zeros(b, c, m, device="cuda")
        nat.three_interpolate_grad_wrapper(b, c, n, m, go_d, idx_d, w_d, gp)
        return gp

    outs, fast = _both_modes(interp)
    _check(outs, fast, what="three_interpolate_grad")
    want = oracle.three_interpolate_grad(go.numpy(), idx.numpy(), w.numpy(), m)
    assert np.abs(outs[0].cpu().numpy() - want).max() <= 1e-5 * max(1.0, np.abs(want).max())

    gidx = torch.randint(0, m, (b, n), generator=g, dtype=torch.int32).cuda()

    def gather():
        gp = torch.zeros(b, c, m, device="cuda")
        nat.gather_points_grad_wrapper(b, c, m, n, go_d, gidx, gp)
        return gp

    outs, fast = _both_modes(gather)
    _check(outs, fast, what="gather_points_grad")


def test_transposed_lists_are_sorted_and_their_consumers_repeat(det):
    """ogc_group_reverse / ogc_reverse_neighbours hand out their lists sorted in the mode, so the gather-form gradients that walk
    them (ogc_group_points_grad_rev, ogc_three_interpolate_grad_rev, ogc_neighbour_consistency_bwd) repeat bit for bit."""
    from ogc_amd import fused
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(5)
    b, n, npoint, nsample, c = 2, 2048, 512, 32, 24
    idx = _neighbours(b, npoint, nsample, n, g).cuda()
    go = torch.randn(b, c, npoint, nsample, generator=g).cuda()
    res = []
    for _ in range(3):
        rev = fused.group_reverse(idx, n)
        assert rev is not None
        gp = torch.empty(b, c, n, device="cuda")
        nat.group_points_grad_rev_wrapper(b, c, n, npoint, nsample, go, rev[0], rev[1], rev[2], gp)
        res.append((rev[0].clone(), gp))   # (rev_pos itself has unused slots behind the lists — only run heads are listed)
    for r in res[1:]:
        assert torch.equal(r[0], res[0][0]) and torch.equal(r[1], res[0][1])
    scat = torch.zeros(b, c, n, device="cuda")
    nat.group_points_grad_wrapper(b, c, n, npoint, nsample, go, idx, scat)
    assert ((scat - res[0][1]).norm() / scat.norm()).item() <= 1e-5
    # the loss's neighbour lists
    k, N, C = 16, 1024, 10
    nidx = _neighbours(b, N, k, N, g).cuda()
    mask = torch.rand(b, N, C, generator=g).cuda()
    gout = torch.randn(b, N, generator=g).cuda()
    grads = []
    for _ in range(3):
        rs, src, mult = fused.reverse_neighbours(nidx)
        gm = torch.empty_like(mask)
        nat.neighbour_consistency_bwd_wrapper(b, N, C, k, 1, mask, nidx, rs, src, mult, gout, gm)
        used = int(rs[:, -1].max())   # (behind the last list rev_src is unused memory)
        grads.append((src[:, :used].clone() if int(rs[:, -1].min()) == used else rs.clone(), gm))
    for r in grads[1:]:
        assert torch.equal(r[0], grads[0][0]) and torch.equal(r[1], grads[0][1])


@pytest.mark.parametrize("b,cin,cout,hw", [(2, 6, 32, 4096), (4, 64, 64, 8192), (2, 131, 128, 4096), (2, 128, 256, 8192), (3, 35, 64, 48)])
def test_weight_gradient(b, cin, cout, hw):
    """All tile shapes of ogc_conv1x1_wgrad (register tiles 16..64, the shared 128 x 128 tile): partial tiles + one ordered pass."""
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(cin * cout)
    x = torch.randn(b, cin, hw, generator=g).cuda()
    dy = torch.randn(b, cout, hw, generator=g).cuda()

    def run():
        dw = torch.zeros(cout, cin, device="cuda")
        nat.conv1x1_wgrad_wrapper(b, cin, cout, hw, x, dy, dw)
        return dw

    outs, fast = _both_modes(run)
    _check(outs, fast, what="conv1x1_wgrad %d->%d" % (cin, cout))
    want = torch.einsum("bop,bip->oi", dy.double(), x.double())
    assert ((outs[0].double() - want).norm() / want.norm()).item() <= 1e-5


def test_batch_norm_statistics_and_gradients():
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(9)
    b, c, p, s = 4, 48, 512, 16
    hw = p * s
    x = torch.randn(b, c, p, s, generator=g).cuda()
    gamma, beta = torch.rand(c, generator=g).cuda() + 0.5, torch.randn(c, generator=g).cuda()
    gy = torch.randn(b, c, p, s, generator=g).cuda()
    gout = torch.randn(b, c, p, generator=g).cuda()

    def run():
        rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
        y, mean, rstd = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        ws = torch.empty(4 * c, dtype=torch.float64, device="cuda")
        nat.batch_norm_fwd_wrapper(b, c, hw, 1e-5, 1, 1, 0.1, x, gamma, beta, rm, rv, y, mean, rstd, ws, None, 0)
        gx, gg, gb = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        nat.batch_norm_bwd_wrapper(b, c, hw, 1, 1, x, gamma, beta, mean, rstd, gy, gx, gg, gb, ws)
        out, arg = torch.empty(b, c, p, device="cuda"), torch.empty(b, c, p, dtype=torch.int32, device="cuda")
        mean2, rstd2 = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        nat.batch_norm_maxpool_fwd_wrapper(b, c, p, s, 1e-5, 1, 1, 0.1, x, gamma, beta, rm, rv, out, arg, mean2, rstd2, ws, None, 0)
        px, pg, pb = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
        nat.batch_norm_maxpool_bwd_wrapper(b, c, p, s, 1, 1, x, gamma, mean2, rstd2, out, arg, gout, px, pg, pb, ws)
        return y, mean, rstd, gx, gg, gb, out, px, pg, pb, rm, rv

    outs, fast = _both_modes(run)
    _check(outs, fast, what="batch norm")
    ref = torch.nn.functional.batch_norm(x.double(), None, None, gamma.double(), beta.double(), True, 0.1, 1e-5).relu()
    assert ((outs[0][0].double() - ref).norm() / ref.norm()).item() <= 1e-5


def test_chamfer_gradient_and_rigid_moments():
    from ogc_amd import pointnet2_cuda as nat
    g = torch.Generator().manual_seed(21)
    b, n1, n2 = 3, 700, 900
    p1, pc2 = torch.randn(b, n1, 3, generator=g).cuda(), torch.randn(b, n2, 3, generator=g).cuda()
    idx12 = torch.randint(0, n2, (b, n1), generator=g, dtype=torch.int32).cuda()
    idx21 = torch.randint(0, 40, (b, n2), generator=g, dtype=torch.int32).cuda()   # (a few popular points: long lists)
    g1, g2 = torch.randn(b, n1, generator=g).cuda(), torch.randn(b, n2, generator=g).cuda()
    for p in (1, 2):
        def run():
            out = torch.empty(b, n1, 3, device="cuda")
            nat.chamfer_terms_grad_wrapper(b, n1, n2, p, p1, pc2, idx12, idx21, g1, g2, out)
            return out
        outs, fast = _both_modes(run)
        _check(outs, fast, what="chamfer_terms_grad p=%d" % p)
    vb, n, k = 4, 8192, 10
    pc, q = torch.randn(vb, n, 3, generator=g).cuda(), torch.randn(vb, n, 3, generator=g).cuda()
    mask = torch.rand(vb, n, k, generator=g).softmax(-1).cuda()

    def moments():
        mom = torch.empty(vb * k * 16, dtype=torch.float64, device="cuda")
        S, means = torch.empty(vb * k, 9, device="cuda"), torch.empty(vb * k, 6, device="cuda")
        nat.rigid_moments_wrapper(vb, n, k, pc, q, mask, mom, S, means)
        return mom, S, means

    outs, fast = _both_modes(moments)
    _check(outs, fast, rel=1e-12, what="rigid_moments")


def _child(which):
    env = dict(os.environ, OGC_DETERMINISTIC="1", PYTHONPATH=ROOT)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "det_probe.py"), which, "2"], env=env, cwd=ROOT,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    differ = re.search(r"(\d+) of (\d+) parameter gradients differ", out.stdout)
    sha = re.search(r"sha256 of the first run's gradients: ([0-9a-f]{64})", out.stdout)
    assert differ and sha, out.stdout[-2000:]
    return int(differ.group(1)), int(differ.group(2)), sha.group(1)


@pytest.mark.parametrize("which", ["seg", "flow"])
def test_two_processes_produce_identical_parameter_gradients(which):
    """One training step's backward pass of segnet_kitti (2 x 4 views x 2048 points, all terms of the OGC loss) / flownet_sapien
    (2 pairs of 512 points, 4 iterations, Chamfer + smoothness) in two child processes under OGC_DETERMINISTIC=1: inside each
    process two runs agree bit for bit on every parameter gradient, and the SHA-256 over all of them is the same in both."""
    d1, n1, sha1 = _child(which)
    d2, n2, sha2 = _child(which)
    assert d1 == 0 and d2 == 0, "%d / %d of %d parameter gradients differ between two runs of one process" % (d1, d2, n1)
    assert n1 == n2 and n1 > 100
    assert sha1 == sha2, "the two processes disagree: %s vs %s" % (sha1, sha2)
