"""BASELINE.json's configurations at their real sizes against fixtures made by the reference's Python in the build
container (tests/golden/fullsize.npz, make_golden.py::gen_fullsize):

  * SHA-256 of every index tensor of the C4 / C5 / C2 / C3 geometry (FPS chains, kNN + distances, 3-NN, the losses'
    kNN and ball query) — bit-exact by construction, SURVEY §8c's "config-sized index hashes";
  * C2 segnet_ogcdr @ 4096 points and C5 segnet_kitti @ 16384 points: masks and parameter gradients;
  * C3 flownet_kitti on an 8192-point pair: flows;
  * C2 as BASELINE names it — one train step with `matmul_precision: bf16` — against the same step in fp32."""
import hashlib
import importlib
import os

import numpy as np
import pytest
import torch
import yaml

import golden_cases as gc
from golden_cases import detgen

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DEV = "cuda"


@pytest.fixture(autouse=True)
def _hip():
    assert torch.cuda.is_available()
    import ogc_amd  # noqa: F401
    torch.backends.cuda.matmul.allow_tf32 = False


def sha(t):
    return hashlib.sha256(np.ascontiguousarray(t.detach().cpu().numpy()).tobytes()).hexdigest()


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(DEV)


GEOMETRY = {
    "C4": (lambda: detgen.cloud(2, 8192, 81), [2048, 1024, 512], [64, 64, 64], (32, 1.0), (64, 2.0)),
    "C5": (lambda: detgen.cloud(1, 16384, 82), [4096, 2048, 1024], [64, 64, 64], (32, 1.0), (64, 2.0)),
    "C2": (lambda: detgen.cloud(2, 4096, 83, scale=(1, 1, 1)), [2048, 1024], [64, 64], (8, 0.02), (16, 0.04)),
    "C3": (lambda: detgen.cloud(1, 8192, 84), [4096, 2048, 1024, 512, 256], [32, 32, 32, 24, 16], None, None),
}


@pytest.mark.parametrize("tag", sorted(GEOMETRY))
def test_config_sized_index_hashes(tag):
    from ogc_amd.pointnet2.pointnet2 import ball_query, furthest_point_sample, gather_nd, knn, three_nn
    g = gc.load("fullsize")
    want = dict(zip([str(k) for k in g["hash_keys"]], [str(v) for v in g["hash_vals"]]))
    make, levels, ks, loss_knn, loss_ball = GEOMETRY[tag]
    xyz = T(make())
    got, cur = {}, xyz
    for li, (npoint, k) in enumerate(zip(levels, ks)):
        idx = furthest_point_sample(cur, npoint)
        got["%s/fps%d" % (tag, li)] = sha(idx)
        nxt = gather_nd(cur, idx.long()).contiguous()
        d, ki = knn(k, nxt, cur.contiguous())
        got["%s/knn%d" % (tag, li)] = sha(ki)
        got["%s/knn%d_dist" % (tag, li)] = sha(d)
        got["%s/nn3_%d" % (tag, li)] = sha(three_nn(cur.contiguous(), nxt)[1])
        cur = nxt
    if loss_knn:
        got["%s/loss_knn" % tag] = sha(knn(loss_knn[0], xyz, xyz)[1])
    if loss_ball:
        got["%s/loss_ball" % tag] = sha(ball_query(loss_ball[1], loss_ball[0], xyz, xyz))
    assert set(got) == {k for k in want if k.startswith(tag + "/")}
    wrong = [k for k in got if got[k] != want[k]]
    assert not wrong, "index tensors differ from the reference at config size: %s" % wrong


SEG = [("C2", "segnet_ogcdr", dict(n_slot=8, n_point=4096, transformer_embed_dim=128), 4096, 2, (1, 1, 1)),
       ("C5", "segnet_kitti", dict(n_slot=10, n_point=16384, transformer_embed_dim=128), 16384, 1, (60, 4, 80))]


@pytest.mark.parametrize("tag,name,kw,N,B,scale", SEG, ids=[c[0] for c in SEG])
def test_segnet_at_config_size(tag, name, kw, N, B, scale):
    g = gc.load("fullsize")
    mod = importlib.import_module("ogc_amd.models." + name)
    net = detgen.fill_module(mod.MaskFormer3D(**kw), 7).to(DEV)
    pc = T(detgen.cloud(B, N, 85, scale=scale))
    mask = net(pc, pc)
    # two fp32 evaluations of the same function: each is ~1e-6 (relative L2) from the exact result (test_truth_f64_gpu)
    assert gc.rel_l2(mask[:, ::16], g[tag + "/mask"]) < 5e-6
    assert abs(float(mask.detach().double().norm()) / float(g[tag + "/mask_norm"][0]) - 1.0) < 1e-6
    gc.close(mask[:, ::16], g[tag + "/mask"], 2e-4, 2e-6, tag + "/mask")
    target = T(detgen.uniform(tuple(mask.shape), 86, 0.0, 1.0))
    net.zero_grad()
    ((mask - target) ** 2).mean().backward()
    ours, ref = [], []
    for pname, p in net.named_parameters():
        ours.append(float(p.grad.norm()) if p.grad is not None else 0.0)
        ref.append(float(g["%s/gnorm/%s" % (tag, pname)][0]))
    ours, ref = np.array(ours), np.array(ref)
    # all gradient norms as one vector; single tensors may sit on a flipped ReLU / max-pool gate (see ErrorBudget)
    assert np.linalg.norm(ours - ref) / np.linalg.norm(ref) < 2e-4
    assert np.median(np.abs(ours - ref) / np.maximum(ref, 1e-12)) < 5e-5


def test_flownet_kitti_at_config_size():
    g = gc.load("fullsize")
    mod = importlib.import_module("ogc_amd.models.flownet_kitti")
    net = detgen.fill_module(mod.FlowStep3D(npoint=8192, loc_flow_nn=16, loc_flow_rad=1.5), 8).to(DEV)
    net.eval()
    pc1, pc2 = T(detgen.cloud(1, 8192, 87)), T(g["C3/pc2"])
    with torch.no_grad():
        preds = net(pc1, pc2, pc1, pc2, iters=2)
    assert len(preds) == 2
    for i, p in enumerate(preds):
        assert gc.rel_l2(p[:, ::8], g["C3/flow%d" % i]) < 5e-6, i
        assert abs(float(p.double().norm()) / float(g["C3/flow%d_norm" % i][0]) - 1.0) < 1e-6
        gc.close(p[:, ::8], g["C3/flow%d" % i], 1e-4, 1e-5, "C3/flow%d" % i)


def test_flownet_two_clouds_per_encoder_call_changes_nothing():
    """In evaluation mode the encoders see both clouds of a pair as one batch (models/_flownet.py
    `_two_clouds_per_call`); the reference makes one call per cloud (models/flownet_kitti.py:213-214, :199-200).  Every
    operator works per cloud: the predictions must be the same numbers, and with BatchNorm in training mode the calls
    must stay apart."""
    mod = importlib.import_module("ogc_amd.models.flownet_kitti")
    net = detgen.fill_module(mod.FlowStep3D(npoint=2048, loc_flow_nn=16, loc_flow_rad=1.5), 8).to(DEV)
    pc1, pc2 = T(detgen.cloud(2, 2048, 5)), T(detgen.cloud(2, 2048, 6))
    net.train()
    assert not net._two_clouds_per_call()
    net.eval()
    assert net._two_clouds_per_call()
    with torch.no_grad():
        both = net(pc1, pc2, pc1, pc2, iters=3)
        net._two_clouds_per_call = lambda: False
        apart = net(pc1, pc2, pc1, pc2, iters=3)
    for a, b in zip(both, apart):
        assert torch.equal(a, b)


def test_c2_bf16_train_step_tracks_fp32():
    """BASELINE config 2: OGC-DR segnet_ogcdr, 4096-point clouds, bf16 — config/ogcdr_unsup_synthetic.yaml through one
    `train_step`, with bf16 operands in the 1x1 convolutions against the same step with fp32 operands (same weights,
    same batch).  bf16 rounds operands to 8 significand bits (2^-9 relative): losses must agree to ~1e-2, the gradient
    as a whole vector to a few percent, and the step must be taken (finite gradients)."""
    from ogc_amd.pointnet2 import pointnet2 as api
    from ogc_amd.train_seg import build_segnet
    from ogc_amd.train_step import build_criterion, make_optimizer, train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    with open(os.path.join(ROOT, "config", "ogcdr_unsup_synthetic.yaml")) as f:
        cfg = yaml.safe_load(f)
    assert cfg["matmul_precision"] == "bf16" and cfg["segnet"]["n_point"] == 4096
    batch = make_scene_batch(2, 4096, cfg["segnet"]["n_slot"], seed=3, outdoor=False, aug=True, device=DEV)
    res = {}
    try:
        for prec in ("fp32", "bf16"):
            api._native.set_matmul_precision(prec)
            torch.manual_seed(cfg["random_seed"])
            net = build_segnet(cfg).to(DEV)
            crit = build_criterion(cfg["loss"])
            opt = make_optimizer(net.parameters(), lr=cfg["lr"])
            before = [p.detach().clone() for p in net.parameters()]
            losses, stepped = train_step(net, crit, opt, batch, 10, True)
            grads = torch.cat([p.grad.flatten() for p in net.parameters()])
            moved = sum(float((p.detach() - b).abs().sum()) for p, b in zip(net.parameters(), before))
            res[prec] = (losses, stepped, grads.double(), moved)
    finally:
        api._native.set_matmul_precision("fp32")
    (l32, s32, g32, m32), (l16, s16, g16, m16) = res["fp32"], res["bf16"]
    assert s32 and s16 and m32 > 0 and m16 > 0
    assert torch.isfinite(g16).all()
    for k in ("dynamic", "smooth", "invariance", "sum"):
        assert abs(l16[k] - l32[k]) <= 2e-2 * abs(l32[k]) + 1e-4, (k, l16[k], l32[k])
    cos = float((g16 * g32).sum() / (g16.norm() * g32.norm()))
    assert cos > 0.98, cos
    assert abs(float(g16.norm() / g32.norm()) - 1.0) < 0.1
    assert float((g16 - g32).norm() / g32.norm()) > 1e-5   # bf16 really was in effect
