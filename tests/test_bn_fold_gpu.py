"""BatchNorm's finalize / params steps folded into the apply / dx kernels (csrc/batch_norm.hip, round 6: BnFin / BnBwdFin) against
the separate launches they replace (OGC_BN_FOLD=0): every output — activations, mean, rstd, running statistics, input gradient,
dgamma, dbeta, for the plain and the max-pooled form — must have the SAME BITS (each workgroup evaluates the same double-precision
expressions the one-thread-per-channel kernels did).  The switch is read once per process: two child processes."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, torch
import ogc_amd
from ogc_amd import pointnet2_cuda as nat
g = torch.Generator().manual_seed(31)
b, c, p, s = 3, 40, 300, 16
hw = p * s
x = (torch.randn(b, c, p, s, generator=g) * 2 + 0.3).cuda()
gamma, beta = (torch.rand(c, generator=g) + 0.5).cuda(), torch.randn(c, generator=g).cuda()
gy, gout = torch.randn(b, c, p, s, generator=g).cuda(), torch.randn(b, c, p, generator=g).cuda()
out = {}
for relu in (0, 1):
    rm, rv = torch.zeros(c, device="cuda"), torch.ones(c, device="cuda")
    y, mean, rstd = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    ws = torch.zeros(4 * c, dtype=torch.float64, device="cuda")
    nat.batch_norm_fwd_wrapper(b, c, hw, 1e-5, relu, 1, 0.1, x, gamma, beta, rm, rv, y, mean, rstd, ws, None, 0)
    gx, gg, gb = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    nat.batch_norm_bwd_wrapper(b, c, hw, relu, 1, x, gamma, beta, mean, rstd, gy, gx, gg, gb, ws)
    o, arg = torch.empty(b, c, p, device="cuda"), torch.empty(b, c, p, dtype=torch.int32, device="cuda")
    m2, r2 = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    nat.batch_norm_maxpool_fwd_wrapper(b, c, p, s, 1e-5, relu, 1, 0.1, x, gamma, beta, rm, rv, o, arg, m2, r2, ws, None, 0)
    px, pg, pb = torch.empty_like(x), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    nat.batch_norm_maxpool_bwd_wrapper(b, c, p, s, relu, 1, x, gamma, m2, r2, o, arg, gout, px, pg, pb, ws)
    for k, v in dict(y=y, mean=mean, rstd=rstd, gx=gx, gg=gg, gb=gb, o=o, arg=arg, m2=m2, r2=r2, px=px, pg=pg, pb=pb, rm=rm, rv=rv).items():
        out["%s_relu%d" % (k, relu)] = v.cpu()
torch.save(out, sys.argv[1])
"""


def test_folded_batch_norm_has_the_bits_of_the_separate_launches(tmp_path):
    res = []
    for flag in ("1", "0"):
        path = str(tmp_path / ("bn_%s.pt" % flag))
        env = dict(os.environ, OGC_BN_FOLD=flag, PYTHONPATH=ROOT)
        r = subprocess.run([sys.executable, "-c", CHILD, path], env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        res.append(torch.load(path))
    assert set(res[0]) == set(res[1]) and len(res[0]) == 30
    for k in res[0]:
        assert torch.equal(res[0][k], res[1][k]), k
