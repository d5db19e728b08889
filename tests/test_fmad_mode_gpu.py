"""OGC_FMAD=1: libogc_ops_fmad.so — the search kernels with the squared distance contracted as `nvcc --fmad=true` (the
reference's build) contracts it, fma(dz, dz, fma(dy, dy, dx*dx)) — against the oracle under oracle_set_fmad(1), bit for bit,
on data built so that the two forms of the expression DISAGREE (a 0.1-spaced lattice: 84 of 128 FPS picks differ) and on a
scene-shaped cloud through every search path (all-pairs scans, cell lists with three and nine runs, the bucketed FPS rounds).
The variant is selected when the library is loaded, so each case runs in a process of its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_CHILD = r"""
import sys, numpy as np, torch
sys.path.insert(0, %r)
import ogc_amd
from ogc_amd import _lib, pointnet2_cuda as nat
from oracle import oracle as orc
assert _lib.FMAD and _lib.load().ogc_distance_contracted() == 1
case = sys.argv[1]
dev = "cuda"
def T(a): return torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sys.path.insert(0, %r)
import detgen
if case == "lattice":
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(8), indexing="ij"), -1).reshape(1, -1, 3).astype(np.float32)
    clouds = [(g * np.float32(0.1)).astype(np.float32)]
    shapes = [(128, 16, 0.25, 16)]
else:
    clouds = [detgen.cloud(2, 8192, 91), detgen.cloud(2, 2048, 92, scale=(1, 1, 1)), detgen.cloud(1, 20000, 93)]
    shapes = [(2048, 32, 2.0, 64), (512, 16, 0.1, 32), (700, 8, 2.0, 16)]
changed = 0
for pc, (m, k, r, ns) in zip(clouds, shapes):
    B, N, _ = pc.shape
    t = T(pc)
    with orc.fmad():
        want_fps = orc.fps(pc, m)
        sub = np.take_along_axis(pc, want_fps[..., None].astype(np.int64).repeat(3, -1), 1)
        want_d, want_i = orc.knn(k, sub, pc)
        want_3d, want_3i = orc.three_nn(pc, sub) if N <= 8192 else (None, None)
        want_b = orc.ball_query(r, ns, pc, pc) if N <= 8192 else None
    plain_fps = orc.fps(pc, m)
    changed += int((plain_fps != want_fps).sum())
    idx = torch.empty(B, m, dtype=torch.int32, device=dev); temp = torch.full((B, N), 1e10, device=dev)
    nat.furthest_point_sampling_wrapper(B, N, m, t, temp, idx)
    assert np.array_equal(idx.cpu().numpy(), want_fps), "fps"
    d2 = torch.empty(B, m, k, device=dev); ki = torch.empty(B, m, k, dtype=torch.int32, device=dev)
    nat.knn_wrapper(B, m, N, k, T(sub), t, d2, ki)
    assert np.array_equal(ki.cpu().numpy(), want_i) and np.array_equal(d2.cpu().numpy(), want_d), "knn"
    if want_b is not None:
        d3 = torch.empty(B, N, 3, device=dev); i3 = torch.empty(B, N, 3, dtype=torch.int32, device=dev)
        nat.three_nn_wrapper(B, N, m, t, T(sub), d3, i3)
        assert np.array_equal(i3.cpu().numpy(), want_3i) and np.array_equal(d3.cpu().numpy(), want_3d), "three_nn"
        bq = torch.zeros(B, N, ns, dtype=torch.int32, device=dev)
        nat.ball_query_wrapper(B, N, N, r, ns, t, t, bq)
        assert np.array_equal(bq.cpu().numpy(), want_b), "ball_query"
print("ok", case, "fps picks that differ between the two forms:", changed)
if case == "lattice":
    assert changed > 0, "the lattice was meant to separate the two forms"
"""


@pytest.mark.parametrize("case", ["lattice", "clouds"])
def test_contracted_variant_matches_the_oracle_in_fmad_mode(case, tmp_path, oracle):
    script = tmp_path / "child.py"
    script.write_text(_CHILD % (ROOT, os.path.join(ROOT, "tests", "golden")))
    env = dict(os.environ, OGC_FMAD="1")
    res = subprocess.run([sys.executable, str(script), case], env=env, capture_output=True, text=True, timeout=900)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-4000:]
    assert "ok " + case in res.stdout


def test_default_library_is_the_uncontracted_one():
    from ogc_amd import _lib
    assert not _lib.FMAD and _lib.load().ogc_distance_contracted() == 0
