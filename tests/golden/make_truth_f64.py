"""Generates tests/golden/truth_f64.npz: the module- and model-level scenarios of make_golden.py executed by THE
REFERENCE'S OWN PYTHON in float64 (imported from /root/reference, build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_truth_f64.py

Why: the fp32 goldens (model_*.npz, modules.npz) are one fp32 evaluation of the reference; this repo's HIP path is
another, with a different summation order in every GEMM / normalisation.  Neither is "the" answer, so a tolerance
between the two says little.  The float64 run is the answer both approximate: the tests assert

        err(HIP fp32 vs f64)  <=  2 x err(reference fp32 vs f64)  + floor

per tensor (tests/golden_cases.py::within_reference_error), i.e. this implementation is as close to the exact
result as the reference's own fp32 arithmetic is.

How the reference runs in float64: the same four shims as make_golden.py, except that `torch.cuda.FloatTensor`
allocates float64 and the native module is `F64Backend` below — the index-producing operators (FPS, kNN, 3-NN, ball
query) decide on the float32-rounded coordinates with the oracle (the reference's index arithmetic IS fp32; the
truth must select the same neighbours), every float result (distances, gathers, interpolation, scatter-adds) is
formed in float64.  Only inputs/outputs are written; nothing of the reference is copied.
"""
import os
import sys
import types

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import detgen  # noqa: E402
from oracle import oracle as orc  # noqa: E402


class F64Backend:
    """The ten pybind names (pointnet2/src/pointnet2_api.cpp:10-25) for float64 tensors."""

    def __init__(self):
        self.f32 = orc.Pointnet2CudaCPU()

    @staticmethod
    def _r(t):
        return t.detach().to(torch.float32).contiguous()

    def furthest_point_sampling_wrapper(self, b, n, m, points, temp, idx):
        t32 = torch.full((b, n), 1e10, dtype=torch.float32)
        return self.f32.furthest_point_sampling_wrapper(b, n, m, self._r(points), t32, idx)

    def ball_query_wrapper(self, b, n, m, radius, nsample, new_xyz, xyz, idx):
        return self.f32.ball_query_wrapper(b, n, m, radius, nsample, self._r(new_xyz), self._r(xyz), idx)

    def _dist2(self, unknown, known, idx):
        nb = torch.gather(known.unsqueeze(1).expand(-1, unknown.size(1), -1, -1), 2,
                          idx.long().unsqueeze(-1).expand(-1, -1, -1, 3))
        return ((unknown.unsqueeze(2) - nb) ** 2).sum(-1)

    def knn_wrapper(self, b, n, m, k, unknown, known, dist2, idx):
        d32 = torch.empty(b, n, k, dtype=torch.float32)
        self.f32.knn_wrapper(b, n, m, k, self._r(unknown), self._r(known), d32, idx)
        dist2.copy_(self._dist2(unknown, known, idx))

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        d32 = torch.empty(b, n, 3, dtype=torch.float32)
        self.f32.three_nn_wrapper(b, n, m, self._r(unknown), self._r(known), d32, idx)
        dist2.copy_(self._dist2(unknown, known, idx))

    def group_points_wrapper(self, b, c, n, npoints, nsample, points, idx, out):
        flat = idx.long().reshape(b, 1, npoints * nsample).expand(-1, c, -1)
        out.copy_(torch.gather(points, 2, flat).reshape(b, c, npoints, nsample))
        return 1

    def group_points_grad_wrapper(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        flat = idx.long().reshape(b, 1, npoints * nsample).expand(-1, c, -1)
        grad_points.scatter_add_(2, flat, grad_out.reshape(b, c, npoints * nsample))
        return 1

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        out.copy_(torch.gather(points, 2, idx.long().unsqueeze(1).expand(-1, c, -1)))
        return 1

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        grad_points.scatter_add_(2, idx.long().unsqueeze(1).expand(-1, c, -1), grad_out)
        return 1

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        flat = idx.long().reshape(b, 1, n * 3).expand(-1, c, -1)
        out.copy_((torch.gather(points, 2, flat).reshape(b, c, n, 3) * weight.unsqueeze(1)).sum(-1))

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight, grad_points):
        flat = idx.long().reshape(b, 1, n * 3).expand(-1, c, -1)
        grad_points.scatter_add_(2, flat, (grad_out.unsqueeze(-1) * weight.unsqueeze(1)).reshape(b, c, n * 3))


def install_shims_f64():
    backend = F64Backend()
    mod = types.ModuleType("pointnet2_cuda")
    for name in dir(backend):
        if name.endswith("_wrapper"):
            setattr(mod, name, getattr(backend, name))
    sys.modules["pointnet2_cuda"] = mod
    torch.cuda.FloatTensor = lambda *shape: torch.empty(*shape, dtype=torch.float64)
    torch.cuda.IntTensor = lambda *shape: torch.empty(*shape, dtype=torch.int32)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    tbx = types.ModuleType("tensorboardX")
    tbx.SummaryWriter = object
    sys.modules["tensorboardX"] = tbx
    sys.path.insert(0, REF)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).double()


def fill(module, seed):
    """The fp32 weights of the fixtures (detgen), then promoted: both precisions evaluate the SAME function."""
    return detgen.fill_module(module, seed).double()


def grads_full(module, loss, prefix=""):
    module.zero_grad()
    loss.backward()
    out = {}
    for name, p in module.named_parameters():
        g = p.grad if p.grad is not None else torch.zeros_like(p)
        out[prefix + "gnorm/" + name] = g.norm().reshape(1).numpy()
        out[prefix + "ghead/" + name] = g.flatten()[:32].clone().numpy()
    return out


ULP = 2.0 ** -22   # two fp32 ulps, relative: the rounding noise of a result accumulated over a 64- to 256-term fp32 sum


class Fp32Noise:
    """Context in which `module` is evaluated the way an fp32 machine perturbs it — every parameter and the output of every
    leaf layer multiplied by (1 + ULP * u), u in [-1, 1) from integer hashes — but in float64, so that the ONLY
    difference to the truth is that noise.  The distance of such a run to the truth is the conditioning of the function
    at fp32 resolution (ReLU gates, max-pool winners and neighbour selections can flip under it): no fp32
    implementation, the reference's included, can be expected to be closer to the truth than that."""

    def __init__(self, module, sample):
        self.module, self.sample, self.handles, self.saved, self.count = module, sample, [], [], 0

    def _noise(self, shape):
        self.count += 1
        return torch.from_numpy(detgen.uniform(tuple(shape), 100003 * self.sample + self.count).astype(np.float64)) * ULP

    def __enter__(self):
        with torch.no_grad():
            for p in self.module.parameters():
                self.saved.append(p.detach().clone())
                p.mul_(1.0 + self._noise(p.shape))

        def hook(mod, inp, out):
            if torch.is_tensor(out) and out.is_floating_point():
                return out * (1.0 + self._noise(out.shape))
            return out
        for m in self.module.modules():
            if not list(m.children()):
                self.handles.append(m.register_forward_hook(hook))
        return self

    def __exit__(self, *exc):
        for h in self.handles:
            h.remove()
        with torch.no_grad():
            for p, v in zip(self.module.parameters(), self.saved):
                p.copy_(v)


def rel_l2(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300))


def conditioning(out, prefix, truth, rerun, module, samples=8):
    """out['cond/<key>'] = max over noisy float64 re-evaluations of the relative L2 distance to the truth."""
    worst = {}
    for s in range(samples):
        with Fp32Noise(module, s + 1):
            noisy = rerun()
        for k, v in noisy.items():
            t = truth[k]
            if k.split("/")[-2:-1] == ["ghead"] or "ghead/" in k:
                name = k.split("ghead/")[1]
                tn = float(truth[k.replace("ghead/", "gnorm/")][0])
                numel = dict(module.named_parameters())[name].numel()
                denom = max(np.linalg.norm(t), tn * np.sqrt(min(32, numel) / numel), 1e-300)
                e = float(np.linalg.norm(np.asarray(v, np.float64) - t) / denom)
            else:
                e = rel_l2(v, t)
            worst[k] = max(worst.get(k, 0.0), e)
    for k, e in worst.items():
        out["cond/" + prefix + k] = np.array([e])


def gen_modules(out):
    from utils.flowstep3d_util import FlowEmbedding, PointNetFeaturePropogation, PointNetSetAbstraction
    from utils.pointnet2_util import PointnetFPModule, PointnetSAModuleMSG
    bn = {"class": "GroupNorm", "num_groups": 4}
    pc = T(detgen.cloud(2, 512, 21, scale=(1, 1, 1)))
    feats = T(detgen.uniform((2, 3, 512), 22))
    sa = fill(PointnetSAModuleMSG(npoint=128, radii=[0.2, 0.4], nsamples=[16, 32], mlps=[[3, 16, 16], [3, 16, 32]], bn=bn), 1)
    new_xyz, new_feats, inds = sa(pc, feats, return_inds=True)
    fp = fill(PointnetFPModule(mlp=[48 + 3, 32, 16], bn=bn), 2)
    up = fp(pc, new_xyz, feats, new_feats)
    o = dict(sa_feats=new_feats.detach(), sa_inds=inds, fp_out=up.detach())
    o.update(grads_full(sa, (new_feats ** 2).mean(), "sa_"))
    xyz_t = pc.transpose(1, 2).contiguous()
    f3 = fill(PointNetSetAbstraction(npoint=128, radius=None, nsample=8, in_channel=3, mlp=[16, 32], group_all=False,
                                     return_fps=True), 3)
    nx, nf, fidx = f3(xyz_t, feats)
    f3b = fill(PointNetSetAbstraction(npoint=128, radius=0.3, nsample=8, in_channel=32, mlp=[16], group_all=False,
                                      use_act=False, mean_aggr=True), 4)
    nx2, nf2 = f3b(nx, nf)
    fpf = fill(PointNetFeaturePropogation(in_channel=32 + 3, mlp=[16]), 5)
    upf = fpf(xyz_t, nx, feats, nf)
    pc_b = T(detgen.cloud(2, 128, 23, scale=(1, 1, 1))).transpose(1, 2).contiguous()
    fb = T(detgen.uniform((2, 32, 128), 24))
    fe = fill(FlowEmbedding(radius=0.5, nsample=8, in_channel=32, mlp=[32, 32]), 6)
    _, corr = fe(nx, pc_b, nf, fb)
    o.update(f3_feats=nf.detach(), f3_fps=fidx, f3b_feats=nf2.detach(), fpf_out=upf.detach(), fe_out=corr.detach())
    o.update(grads_full(fe, (corr ** 2).mean(), "fe_"))
    for k, v in o.items():
        out["modules/" + k] = v.numpy() if torch.is_tensor(v) else v


GCORR_CASES = [("flownet_kitti", 1024, 4, 256, (60, 4, 80)), ("flownet_sapien", 512, 3, 256, (1, 1, 1)),
               ("flownet_ogcdr", 512, 3, 128, (1, 1, 1))]


def gcorr_inputs(npoint, n_level, feat_c, scale):
    """Coordinate pyramids (level l+1 = every second point of level l, sizes npoint/4, /8, ...) and the coarsest
    features of both clouds: numpy float32, built from integer hashes only (shared with tests/golden_cases.py)."""
    B = 2
    pc1 = detgen.cloud(B, npoint // 4, 71, scale=scale)
    pc2 = pc1 + detgen.uniform((B, npoint // 4, 3), 72, -0.3, 0.3) * np.asarray(scale, np.float32) / 20
    pc2 = np.ascontiguousarray(pc2[:, ::-1])
    lv1, lv2 = [pc1], [pc2]
    for _ in range(n_level - 1):
        lv1.append(np.ascontiguousarray(lv1[-1][:, ::2]))
        lv2.append(np.ascontiguousarray(lv2[-1][:, ::2]))
    n_top = lv1[-1].shape[1]
    f1, f2 = detgen.uniform((B, feat_c, n_top), 73), detgen.uniform((B, feat_c, n_top), 74)
    return lv1, lv2, f1, f2


def gen_global_corr(out):
    """GlobalCorrLayer (models/flownet_kitti.py:41-81, flownet_sapien.py:38-76) on its own, in fp32 -> global_corr.npz
    (the fixture SURVEY §8c lists) and in fp64 -> truth_f64.npz.  BatchNorm in eval mode (running statistics)."""
    import importlib
    gold = {}
    for name, npoint, n_level, feat_c, scale in GCORR_CASES:
        mod = importlib.import_module("models." + name)
        lv1, lv2, f1n, f2n = gcorr_inputs(npoint, n_level, feat_c, scale)
        for dt, tag in ((torch.float32, "f32"), (torch.float64, "f64")):
            torch.cuda.FloatTensor = lambda *shape, _dt=dt: torch.empty(*shape, dtype=_dt)
            sys.modules["pointnet2_cuda"].__dict__.update(
                {k: getattr(F64Backend() if tag == "f64" else orc.Pointnet2CudaCPU(), k)
                 for k in dir(orc.Pointnet2CudaCPU()) if k.endswith("_wrapper")})
            layer = detgen.fill_module(mod.GlobalCorrLayer(npoint, False), 9).to(dt).eval()
            P1 = [torch.from_numpy(a).to(dt).transpose(1, 2).contiguous() for a in lv1]
            P2 = [torch.from_numpy(a).to(dt).transpose(1, 2).contiguous() for a in lv2]
            f1 = torch.from_numpy(f1n).to(dt).requires_grad_(True)
            f2 = torch.from_numpy(f2n).to(dt).requires_grad_(True)
            feats = layer(P1, P2, f1, f2)
            corr = layer.calc_corr_mat(P1[-1].permute(0, 2, 1), P2[-1].permute(0, 2, 1), f1.permute(0, 2, 1), f2.permute(0, 2, 1))
            tgt = torch.from_numpy(detgen.uniform(tuple(feats.shape), 75)).to(dt)
            g1, g2, ge = torch.autograd.grad((feats * tgt).sum(), [f1, f2, layer.epsilon])
            res = dict(feats=feats.detach(), corr=corr.detach(), g_f1=g1, g_f2=g2, g_eps=ge.reshape(1))
            for k, v in res.items():
                (gold if tag == "f32" else out)["gcorr/%s/%s" % (name, k)] = v.numpy()
    torch.cuda.FloatTensor = lambda *shape: torch.empty(*shape, dtype=torch.float64)
    install_shims_f64()
    path = os.path.join(HERE, "global_corr.npz")
    np.savez_compressed(path, **gold)
    print("wrote global_corr.npz %.1f KB (%d arrays)" % (os.path.getsize(path) / 1024, len(gold)))


def gen_models(out):
    import importlib
    for name, kw, N, B in [("segnet_sapien", dict(n_slot=8, n_point=512, transformer_embed_dim=128), 512, 2),
                           ("segnet_ogcdr", dict(n_slot=8, n_point=512, transformer_embed_dim=128), 512, 2),
                           ("segnet_kitti", dict(n_slot=10, n_point=1024, transformer_embed_dim=128), 1024, 2)]:
        mod = importlib.import_module("models." + name)
        net = fill(mod.MaskFormer3D(**kw), 7)
        scale = (60, 4, 80) if name == "segnet_kitti" else (1, 1, 1)
        pc = T(detgen.cloud(B, N, 41, scale=scale))

        def run(net=net, pc=pc):
            mask = net(pc, pc)
            target = T(detgen.uniform(tuple(mask.shape), 42, 0.0, 1.0))
            res = {"mask": mask.detach().numpy()}
            res.update(grads_full(net, ((mask - target) ** 2).mean()))
            return res
        truth = run()
        for k, v in truth.items():
            out["model_%s/%s" % (name, k)] = v
        conditioning(out, "model_%s/" % name, truth, run, net)
        print(name, "done", flush=True)
    for name, kw, N, iters in [("flownet_sapien", dict(npoint=512, loc_flow_nn=8, loc_flow_rad=0.3), 512, 3),
                               ("flownet_ogcdr", dict(npoint=512, loc_flow_nn=8, loc_flow_rad=0.3), 512, 2),
                               ("flownet_kitti", dict(npoint=1024, loc_flow_nn=16, loc_flow_rad=1.5), 1024, 2)]:
        mod = importlib.import_module("models." + name)
        net = fill(mod.FlowStep3D(**kw), 8)
        net.eval()
        scale = (60, 4, 80) if name == "flownet_kitti" else (1, 1, 1)
        pc1 = T(detgen.cloud(2, N, 51, scale=scale))
        pc2 = T(dict(np.load(os.path.join(HERE, "model_%s.npz" % name)))["pc2"])   # the fp32 fixture's second cloud, exactly

        def run(net=net, pc1=pc1, pc2=pc2, iters=iters):
            preds = net(pc1, pc2, pc1, pc2, iters=iters)
            res = {"flow%d" % i: p.detach().numpy() for i, p in enumerate(preds)}
            res.update(grads_full(net, sum((p ** 2).mean() for p in preds)))
            return res
        truth = run()
        for k, v in truth.items():
            out["model_%s/%s" % (name, k)] = v
        conditioning(out, "model_%s/" % name, truth, run, net)
        print(name, "done", flush=True)


if __name__ == "__main__":
    assert os.path.isdir(REF), "the reference tree is only present in the build container"
    install_shims_f64()
    orc.build()
    out = {}
    gen_modules(out)
    gen_global_corr(out)
    gen_models(out)
    path = os.path.join(HERE, "truth_f64.npz")
    np.savez_compressed(path, **out)
    print("wrote truth_f64.npz %.1f KB (%d arrays)" % (os.path.getsize(path) / 1024, len(out)))
