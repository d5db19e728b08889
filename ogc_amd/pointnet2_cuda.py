"""Drop-in for the reference's native extension module ``pointnet2_cuda``.

The reference builds that module from pointnet2/src/*.cpp,*.cu (PYBIND11_MODULE at
pointnet2/src/pointnet2_api.cpp:10-25) and imports it at pointnet2/pointnet2.py:7.  This module
exposes the same ten functions with the same positional signatures, backed by the hand-written HIP
kernels in libogc_ops.so through its C ABI (include/ogc_ops.h).  Registering it as
``sys.modules['pointnet2_cuda']`` (``ogc_amd.install_drop_in()``) lets reference-style callers run
unchanged on an MI355X.

Like the reference's wrappers, every function fills caller-allocated tensors in place and launches
asynchronously on the *current* stream of the tensor's device, looked up at call time (backward
runs on autograd's thread: SURVEY.md §7 "Autograd threading").
"""
import torch

from . import _lib


def _check(t, dtype, name):
    # ball_query.cpp:12-19 (CHECK_INPUT) is the only reference wrapper that validates; here all do.
    # (a step makes ~900 of these checks on the launch thread, which is level with the GPU at C4: the passing case first)
    try:
        if t.is_cuda and t.dtype is dtype and t.is_contiguous():
            return t.data_ptr()
    except AttributeError:
        pass
    if not isinstance(t, torch.Tensor):
        raise TypeError("%s must be a torch.Tensor" % name)
    if t.device.type != "cuda":
        raise RuntimeError("%s must be a CUDA tensor (HIP device); ogc_amd has no CPU path" % name)
    if not t.is_contiguous():
        raise RuntimeError("%s must be contiguous" % name)
    if t.dtype != dtype:
        raise RuntimeError("%s must be %s, got %s" % (name, dtype, t.dtype))
    return t.data_ptr()


def _f(t, name):
    return _check(t, torch.float32, name)


def _acts(tensors, names):
    """The activation tensors of one call — raw convolution outputs / their gradients, all fp32 or all bf16 — as
    (entry-point suffix, pointers): "_h" selects the 16-bit form (include/ogc_ops.h "16-bit activations")."""
    if tensors[0].dtype is torch.bfloat16:
        return "_h", [_check(t, torch.bfloat16, n) for t, n in zip(tensors, names)]
    return "", [_f(t, n) for t, n in zip(tensors, names)]


def _i(t, name):
    return _check(t, torch.int32, name)


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def _stream(t):
    """The current stream of t's device as a raw handle (torch.cuda.current_stream builds a Stream object: ~5 us per launch)."""
    if _raw_stream is not None:
        return _raw_stream(t.device.index)
    return torch.cuda.current_stream(t.device).cuda_stream


class LaunchTimer:
    """Optional per-launch timing with HIP events recorded on the launch stream (bench.py's roofline leg).
    ``with LaunchTimer({"ogc_ball_query"}) as t: ...; t.durations_ms()`` -> {name: [ms, ...]}."""

    def __init__(self, names):
        self.names = set(names)
        self.records = []

    def __enter__(self):
        global _timer
        _timer = self
        return self

    def __exit__(self, *exc):
        global _timer
        _timer = None

    def durations_ms(self):
        torch.cuda.synchronize()
        out = {}
        for name, e0, e1, dims in self.records:
            out.setdefault(name, []).append((e0.elapsed_time(e1), dims))
        return out


_timer = None


def _run(name, ref_tensor, *args):
    # kernels must be launched with the tensor's device current (multi-GPU processes)
    if torch.cuda.current_device() != ref_tensor.device.index:
        with torch.cuda.device(ref_tensor.device):
            return _run(name, ref_tensor, *args)
    if _timer is not None and name in _timer.names:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        rc = _lib.call(name, *args, _stream(ref_tensor))
        e1.record()
        _timer.records.append((name, e0, e1, tuple(a for a in args if isinstance(a, float) or (isinstance(a, int) and abs(a) < 2 ** 31))))
        return rc
    return _lib.call(name, *args, _stream(ref_tensor))


def ball_query_wrapper(b, n, m, radius, nsample, new_xyz, xyz, idx):
    _run("ogc_ball_query", xyz, b, n, m, float(radius), nsample, _f(new_xyz, "new_xyz"), _f(xyz, "xyz"),
         _i(idx, "idx"))
    return 1


def group_points_wrapper(b, c, n, npoints, nsample, points, idx, out):
    _run("ogc_group_points", points, b, c, n, npoints, nsample, _f(points, "points"), _i(idx, "idx"),
         _f(out, "out"))
    return 1


def group_points_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    _run("ogc_group_points_grad", grad_out, b, c, n, npoints, nsample, _f(grad_out, "grad_out"),
         _i(idx, "idx"), _f(grad_points, "grad_points"))
    return 1


def group_reverse_chunk(n, npoints, nsample):
    """Positions per chunk of the transposed grouping lists (ogc_group_reverse_chunk)."""
    return int(_lib.load().ogc_group_reverse_chunk(int(n), int(npoints), int(nsample)))


def group_reverse_wrapper(b, n, npoints, nsample, idx, rev_start, rev_pos, heads):
    """Transposed lists of a neighbour tensor idx (b, npoints, nsample) into n points (ogc_group_reverse): rev_start
    (b, chunks, n + 1) int32, rev_pos (b, npoints * nsample) int16 storage (16-bit positions)."""
    _run("ogc_group_reverse", idx, b, n, npoints, nsample, _i(idx, "idx"), _i(rev_start, "rev_start"),
         _check(rev_pos, torch.int16, "rev_pos"), _check(heads, torch.int16, "heads"))
    return 1


def group_points_grad_rev_wrapper(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, grad_points):
    """grad_points (b, c, n) = gather-sum of grad_out (b, c, npoints, nsample) over the transposed lists; overwrites."""
    h, (go,) = _acts((grad_out,), ("grad_out",))
    _run("ogc_group_points_grad_rev" + h, grad_out, b, c, n, npoints, nsample, go,
         _i(rev_start, "rev_start"), _check(rev_pos, torch.int16, "rev_pos"), _check(heads, torch.int16, "heads"),
         _f(grad_points, "grad_points"))
    return 1


def group_points_grad_rev_dwx_wrapper(b, c, n, npoints, nsample, grad_out, rev_start, rev_pos, heads, rel, grad_points, dwx):
    """group_points_grad_rev_wrapper + dwx (c, 3) = sum grad_out * rel from the same pass (ogc_group_points_grad_rev_dwx)."""
    h, (go,) = _acts((grad_out,), ("grad_out",))
    _run("ogc_group_points_grad_rev_dwx" + h, grad_out, b, c, n, npoints, nsample, go, _i(rev_start, "rev_start"),
         _check(rev_pos, torch.int16, "rev_pos"), _check(heads, torch.int16, "heads"), _f(rel, "rel"),
         _f(grad_points, "grad_points"), _f(dwx, "dwx"))
    return 1


def three_interpolate_grad_rev_wrapper(b, c, n, m, grad_out, weight, rev_start, rev_pos, heads, grad_points):
    """grad_points (b, c, m) of three_interpolate as a gather over group_reverse(idx (b, n, 3), m); overwrites."""
    _run("ogc_three_interpolate_grad_rev", grad_out, b, c, n, m, _f(grad_out, "grad_out"), _f(weight, "weight"),
         _i(rev_start, "rev_start"), _check(rev_pos, torch.int16, "rev_pos"), _check(heads, torch.int16, "heads"),
         _f(grad_points, "grad_points"))
    return 1


def three_interpolate_grad_rev_sliced_wrapper(b, c, n, m, grad_out, weight, rev_start, rev_pos, heads, grad_points):
    """three_interpolate_grad_rev on a grad_out that is a CHANNEL SLICE of a wider (b, c', n) gradient (rows dense, samples
    c' n apart: what autograd's concatenation node hands down) — read in place (ogc_three_interpolate_grad_rev_bs)."""
    if not (isinstance(grad_out, torch.Tensor) and grad_out.is_cuda and grad_out.dtype is torch.float32 and grad_out.dim() == 3
            and tuple(grad_out.shape) == (b, c, n) and grad_out.stride(2) == 1 and grad_out.stride(1) == n
            and grad_out.stride(0) >= c * n):
        raise RuntimeError("grad_out must be a float32 HIP tensor (b, c, n) with dense rows and planes")
    _run("ogc_three_interpolate_grad_rev_bs", grad_out, b, c, n, m, grad_out.data_ptr(), grad_out.stride(0), _f(weight, "weight"),
         _i(rev_start, "rev_start"), _check(rev_pos, torch.int16, "rev_pos"), _check(heads, torch.int16, "heads"),
         _f(grad_points, "grad_points"))
    return 1


def _host_array(ctype, values):
    return (ctype * len(values))(*values)


def view_means_wrapper(tensors, rows, out):
    """out[k] = mean of row k of the dense fp32 tensors viewed as (rows[i], numel / rows[i]) (ogc_view_means)."""
    import ctypes
    lens = [t.numel() // r for t, r in zip(tensors, rows)]
    if any(l * r != t.numel() or r < 1 for t, r, l in zip(tensors, rows, lens)) or out.numel() != sum(rows):
        raise RuntimeError("view_means: rows do not divide the tensors, or out has the wrong length")
    _run("ogc_view_means", out, len(tensors), _host_array(ctypes.c_void_p, [_f(t, "tensor") for t in tensors]),
         _host_array(ctypes.c_int, rows), _host_array(ctypes.c_longlong, lens), _f(out, "out"))
    return 1


def view_means_grad_wrapper(grads, rows, weight, g_loss):
    """grads[i] viewed as (rows[i], len_i) = (weight[k] * g_loss) / len_i, row by row (ogc_view_means_grad)."""
    import ctypes
    lens = [t.numel() // r for t, r in zip(grads, rows)]
    if any(l * r != t.numel() or r < 1 for t, r, l in zip(grads, rows, lens)) or weight.numel() != sum(rows) or g_loss.numel() != 1:
        raise RuntimeError("view_means_grad: rows do not divide the tensors, or weight / g_loss have the wrong length")
    _run("ogc_view_means_grad", weight, len(grads), _host_array(ctypes.c_void_p, [_f(t, "grad") for t in grads]),
         _host_array(ctypes.c_int, rows), _host_array(ctypes.c_longlong, lens), _f(weight, "weight"), _f(g_loss, "g_loss"))
    return 1


def sum_ranges_wrapper(parts, firsts, out):
    """out (flat) = sum of the dense fp32 parts, part i laid over the elements [firsts[i], firsts[i] + numel) (ogc_sum_ranges)."""
    import ctypes
    _run("ogc_sum_ranges", out, len(parts), _host_array(ctypes.c_void_p, [_f(t, "part") for t in parts]),
         _host_array(ctypes.c_longlong, firsts), _host_array(ctypes.c_longlong, [t.numel() for t in parts]), out.numel(),
         _f(out, "out"))
    return 1


def gather_points_wrapper(b, c, n, npoints, points, idx, out):
    _run("ogc_gather_points", points, b, c, n, npoints, _f(points, "points"), _i(idx, "idx"), _f(out, "out"))
    return 1


def gather_points_grad_wrapper(b, c, n, npoints, grad_out, idx, grad_points):
    _run("ogc_gather_points_grad", grad_out, b, c, n, npoints, _f(grad_out, "grad_out"), _i(idx, "idx"),
         _f(grad_points, "grad_points"))
    return 1


def furthest_point_sampling_wrapper(b, n, m, points, temp, idx):
    _run("ogc_furthest_point_sampling", points, b, n, m, _f(points, "points"), _f(temp, "temp"),
         _i(idx, "idx"))
    return 1


def furthest_point_sampling_chain_wrapper(b, n, m, points, temp, idx, ties_in, ties_out):
    """FPS along a chain of levels (ogc_furthest_point_sampling_chain): `ties_in` (b,) int32 from the run that produced
    `points` (its first n samples in order) lets tie-free samples skip the rounds; `ties_out` (b,) int32 records this
    run's tie count.  Either may be None; so may `temp` for n <= 16384 (the minima start at 1e10 and are not returned)."""
    _run("ogc_furthest_point_sampling_chain", points, b, n, m, _f(points, "points"), 0 if temp is None else _f(temp, "temp"),
         _i(idx, "idx"),
         _opt(ties_in, torch.int32, "ties_in"), _opt(ties_out, torch.int32, "ties_out"))
    return 1


def knn_wrapper(b, n, m, k, unknown, known, dist2, idx):
    _run("ogc_knn", unknown, b, n, m, k, _f(unknown, "unknown"), _f(known, "known"), _f(dist2, "dist2"),
         _i(idx, "idx"))


def three_nn_wrapper(b, n, m, unknown, known, dist2, idx):
    _run("ogc_three_nn", unknown, b, n, m, _f(unknown, "unknown"), _f(known, "known"), _f(dist2, "dist2"),
         _i(idx, "idx"))


def three_interpolate_wrapper(b, c, m, n, points, idx, weight, out):
    _run("ogc_three_interpolate", points, b, c, m, n, _f(points, "points"), _i(idx, "idx"),
         _f(weight, "weight"), _f(out, "out"))


def three_interpolate_grad_wrapper(b, c, n, m, grad_out, idx, weight, grad_points):
    _run("ogc_three_interpolate_grad", grad_out, b, c, n, m, _f(grad_out, "grad_out"), _i(idx, "idx"),
         _f(weight, "weight"), _f(grad_points, "grad_points"))


# ogc_ball_query writes every row of idx itself (rows without a hit as zeros): callers need not pre-zero it
BALL_QUERY_WRITES_ALL_ROWS = True


# ---- fused extension (no reference counterpart at this level; see include/ogc_ops.h) ----------
def knn_clamped_wrapper(b, n, m, k, radius, unknown, known, dist, idx):
    """kNN + sqrt + radius clamp in one launch; radius < 0 disables the clamp."""
    _run("ogc_knn_clamped", unknown, b, n, m, k, float(radius), _f(unknown, "unknown"), _f(known, "known"),
         _f(dist, "dist"), _i(idx, "idx"))


class CellGrid:
    """One cell grid for several radius searches of a batch of clouds in themselves (ogc_cell_grid_build): `xyz` (b, n, 3), cells
    of edge 1.01 x `radius`.  `ball_query(radius, nsample)` / `knn_clamped(k, radius)` give what ball_query_wrapper /
    knn_clamped_wrapper give for (xyz, xyz) and any radius up to the grid's, without sorting the clouds into cells again."""

    def __init__(self, xyz, radius):
        b, n, _ = xyz.shape
        self.xyz, self.radius, self.b, self.n = xyz, float(radius), b, n
        nbytes = int(_lib.load().ogc_cell_grid_bytes(b, n))
        self.buf = torch.empty(nbytes, dtype=torch.uint8, device=xyz.device)
        _run("ogc_cell_grid_build", xyz, b, n, self.radius, _f(xyz, "xyz"), self.buf.data_ptr())

    @staticmethod
    def applies(xyz, radius):
        return xyz.is_cuda and xyz.dim() == 3 and xyz.shape[1] >= 1024 and 0.0 < float(radius) < 3.0e38

    def ball_query(self, radius, nsample, idx):
        _run("ogc_ball_query_cells", self.xyz, self.b, self.n, float(radius), nsample, _f(self.xyz, "xyz"), self.buf.data_ptr(),
             self.radius, _i(idx, "idx"))

    def knn_clamped(self, k, radius, dist, idx):
        _run("ogc_knn_clamped_cells", self.xyz, self.b, self.n, k, float(radius), _f(self.xyz, "xyz"), self.buf.data_ptr(),
             self.radius, _f(dist, "dist"), _i(idx, "idx"))


def chamfer_terms_wrapper(b, n1, n2, p, p1, pc2, idx12, idx21, dist1, dist2):
    """dist1[b, i] = |p1[b, i] - pc2[b, idx12[b, i]]|_p, dist2[b, j] = |pc2[b, j] - p1[b, idx21[b, j]]|_p (ogc_chamfer_terms)."""
    _run("ogc_chamfer_terms", p1, b, n1, n2, int(p), _f(p1, "p1"), _f(pc2, "pc2"), _i(idx12, "idx12"), _i(idx21, "idx21"),
         _f(dist1, "dist1"), _f(dist2, "dist2"))


def chamfer_terms_grad_wrapper(b, n1, n2, p, p1, pc2, idx12, idx21, g1, g2, grad_p1):
    """grad_p1 = d(sum g1 dist1 + sum g2 dist2) / d p1 with the indices held constant (ogc_chamfer_terms_grad)."""
    _run("ogc_chamfer_terms_grad", p1, b, n1, n2, int(p), _f(p1, "p1"), _f(pc2, "pc2"), _i(idx12, "idx12"), _i(idx21, "idx21"),
         _f(g1, "g1"), _f(g2, "g2"), _f(grad_p1, "grad_p1"))


def gather_xyz_pair_wrapper(b, n, m, xyz, idx, out, out_t):
    """out (b, 3, m) = xyz (b, 3, n)[:, :, idx] and out_t (b, m, 3), its transpose, in one launch (ogc_gather_xyz_pair)."""
    _run("ogc_gather_xyz_pair", xyz, b, n, m, _f(xyz, "xyz"), _i(idx, "idx"), _f(out, "out"), _f(out_t, "out_t"))


def flow_advance_wrapper(b, n, scale, cur, delta, ref, out_delta, out_new, out_new_t, out_flow):
    """d = delta * scale, new = cur + d, flow = new - ref on (b, 3, n) tensors (ogc_flow_advance); out_delta, out_new_t
    (b, n, 3) and out_flow may be None."""
    _run("ogc_flow_advance", cur, b, n, float(scale), _f(cur, "cur"), _f(delta, "delta"), 0 if ref is None else _f(ref, "ref"),
         0 if out_delta is None else _f(out_delta, "out_delta"), _f(out_new, "out_new"),
         0 if out_new_t is None else _f(out_new_t, "out_new_t"), 0 if out_flow is None else _f(out_flow, "out_flow"))


def linear_cn_wrapper(b, cin, cout, n, x, weight, bias, y):
    """y (b, cout, n) = weight (cout, cin) x (b, cin, n) + bias, cout <= 4 (ogc_linear_cn); bias may be None."""
    _run("ogc_linear_cn", x, b, cin, cout, n, _f(x, "x"), _f(weight, "weight"), 0 if bias is None else _f(bias, "bias"), _f(y, "y"))


def three_nn_weights_wrapper(b, n, mode, dist2, weight):
    """Normalised inverse-distance weights (b, n, 3) from three_nn_wrapper's squared distances (ogc_three_nn_weights): mode 0 clamps
    the distance at 1e-10 (FlowStep3D), mode 1 adds 1e-8 (the segmentation nets)."""
    _run("ogc_three_nn_weights", dist2, b, n, int(mode), _f(dist2, "dist2"), _f(weight, "weight"))


def soft_corr_flow_wrapper(b, n1, n2, c, support, epsilon, pc1, pc2, f1, f2, flow):
    """GlobalCorrLayer's soft correspondence weights and the coarse flow they imply, in one launch (ogc_soft_corr_flow)."""
    _run("ogc_soft_corr_flow", pc1, b, n1, n2, c, float(support), _f(epsilon, "epsilon"), _f(pc1, "pc1"), _f(pc2, "pc2"),
         _f(f1, "f1"), _f(f2, "f2"), _f(flow, "flow"))


def _gate_ptr(t, channel0, c, name):
    """Channels [channel0, channel0 + c) of a contiguous (b, ctot, n, s) fp32 tensor: (address, batch stride in floats)."""
    base = _f(t, name)
    ctot, n, s = t.shape[1], t.shape[2], t.shape[3]
    if channel0 < 0 or channel0 + c > ctot:
        raise ValueError("%s: channels [%d, %d) of %d" % (name, channel0, channel0 + c, ctot))
    return base + 4 * channel0 * n * s, ctot * n * s


def gru_reset_wrapper(b, c, cx, n, s, rc, hx, out, rc_channel0=0):
    """out (b, c + cx, n) = cat([sigmoid(max_s rc) * h, x]) for hx = cat([h, x]) and the un-pooled gate: channels [rc_channel0,
    rc_channel0 + c) of rc (b, >= c, n, s) (ogc_gru_reset)."""
    ptr, bs = _gate_ptr(rc, rc_channel0, c, "rc")
    _run("ogc_gru_reset", hx, b, c, cx, n, s, ptr, bs, _f(hx, "hx"), _f(out, "out"))


def gru_blend_wrapper(b, c, n, s, zc, qc, h, h_batch_stride, out, zc_channel0=0):
    """out (b, c, n) = (1 - z) * h + z * q, z = sigmoid(max_s zc), q = tanh(max_s qc) (ogc_gru_blend); zc: channels [zc_channel0,
    zc_channel0 + c) of a (b, >= c, n, s) tensor; h may be the first c channels of a wider contiguous tensor whose batch stride
    (in floats) is h_batch_stride."""
    zp, zbs = _gate_ptr(zc, zc_channel0, c, "zc")
    qp, qbs = _gate_ptr(qc, 0, c, "qc")
    _run("ogc_gru_blend", h, b, c, n, s, zp, zbs, qp, qbs, _f(h, "h"), int(h_batch_stride), _f(out, "out"))


def kabsch_rotation_wrapper(nb, S, R, valid=None):
    """R = V diag(1,1,det) U^T per 3x3 cross-covariance (ogc_kabsch_rotation); NaN matrices give the identity."""
    _run("ogc_kabsch_rotation", S, nb, _f(S, "S"), _f(R, "R"), 0 if valid is None else _i(valid, "valid"))


def group_concat_wrapper(b, c, n, npoints, nsample, xyz, new_xyz, points, idx, out):
    """out = cat([xyz[idx] - new_xyz, points[idx]], dim=1) (ogc_group_concat); points may be None when c == 0."""
    _run("ogc_group_concat", xyz, b, c, n, npoints, nsample, _f(xyz, "xyz"), _f(new_xyz, "new_xyz"),
         0 if points is None else _f(points, "points"), _i(idx, "idx"), _f(out, "out"))


def group_concat_grad_wrapper(b, c, n, npoints, nsample, grad_out, idx, grad_points):
    """grad_points += scatter of channels 3.. of grad_out (ogc_group_concat_grad)."""
    _run("ogc_group_concat_grad", grad_out, b, c, n, npoints, nsample, _f(grad_out, "grad_out"), _i(idx, "idx"),
         _f(grad_points, "grad_points"))


def group_linear_fwd_wrapper(b, m, n, npoints, nsample, groups, P, idx, rel, wx, y, stats):
    """y = P[idx] + wx . rel — the first layer of a set-abstraction MLP without the grouped tensor (ogc_group_linear_fwd);
    stats: float64, conv1x1_gn_slots() * b * groups * 2 elements, or None with groups == 0."""
    h, (yp,) = _acts((y,), ("y",))
    _run("ogc_group_linear_fwd" + h, P, b, m, n, npoints, nsample, int(groups), _f(P, "P"), _i(idx, "idx"), _f(rel, "rel"),
         _f(wx, "wx"), yp, _opt(stats, torch.float64, "stats"))


def group_linear_fwd_pt_wrapper(b, m, n, npoints, nsample, groups, Pt, idx, rel, wx, y, stats):
    """group_linear_fwd_wrapper for a bf16 y with P stored point-major, Pt (b, n, m) (ogc_group_linear_fwd_pt_h)."""
    _run("ogc_group_linear_fwd_pt_h", Pt, b, m, n, npoints, nsample, int(groups), _f(Pt, "Pt"), _i(idx, "idx"), _f(rel, "rel"),
         _f(wx, "wx"), _check(y, torch.bfloat16, "y"), _opt(stats, torch.float64, "stats"))


def group_linear_fwd_direct_wrapper(b, m, cf, n, npoints, nsample, groups, feats, idx, rel, w, y, stats):
    """ogc_group_linear_fwd for 1 .. 4 feature channels: one fused-multiply-add chain over [rel, features[idx]] per output
    (ogc_group_linear_fwd_direct); w (m, 3 + cf) fp32, y fp32."""
    h, (yp,) = _acts((y,), ("y",))
    _run("ogc_group_linear_fwd_direct" + h, feats, b, m, cf, n, npoints, nsample, int(groups), _f(feats, "feats"), _i(idx, "idx"),
         _f(rel, "rel"), _f(w, "w"), yp, _opt(stats, torch.float64, "stats"))


def group_linear_bwd_wrapper(b, m, n, npoints, nsample, grad_y, idx, rel, grad_p, dwx):
    """grad_p += scatter of grad_y, dwx += grad_y . rel in one pass over grad_y (ogc_group_linear_bwd); both zeroed by
    the caller.  Raises OgcOpsError (unsupported) outside n <= 16384, npoints * nsample >= 4096 and % 16 == 0."""
    _run("ogc_group_linear_bwd", grad_y, b, m, n, npoints, nsample, _f(grad_y, "grad_y"), _i(idx, "idx"), _f(rel, "rel"),
         _f(grad_p, "grad_p"), _f(dwx, "dwx"))


def soft_nn_target_wrapper(b, n1, n2, k, temperature, p1, p2, mask1, mask2, target):
    """Soft nearest-neighbour targets of OA-ICP without the (b, n1, n2) tensors (ogc_soft_nn_target)."""
    _run("ogc_soft_nn_target", p1, b, n1, n2, k, float(temperature), _f(p1, "p1"), _f(p2, "p2"), _f(mask1, "mask1"),
         _f(mask2, "mask2"), _f(target, "target"))


def rigid_moments_wrapper(vb, n, k, pc, pc2, mask, mom, S, means):
    """Weighted moments -> centred cross-covariances S and means per (cloud, slot) (ogc_rigid_moments)."""
    _run("ogc_rigid_moments", pc, vb, n, k, _f(pc, "pc"), _f(pc2, "pc2"), _f(mask, "mask"),
         _check(mom, torch.float64, "mom"), _f(S, "S"), _f(means, "means"))


def rigid_translation_wrapper(total, means, valid, R, t):
    """t = qbar - R pbar; invalid fits -> identity / zero (ogc_rigid_translation)."""
    _run("ogc_rigid_translation", means, total, _f(means, "means"), _i(valid, "valid"), _f(R, "R"), _f(t, "t"))


def rigid_blend_wrapper(vb, n, k, p, backward, pc, pc2, mask, R, t, grad_out, out):
    """Per-point residual of the mask-blended rigid motions, or its gradient w.r.t. the mask (ogc_rigid_blend)."""
    _run("ogc_rigid_blend", pc, vb, n, k, int(p), int(backward), _f(pc, "pc"), _f(pc2, "pc2"), _f(mask, "mask"),
         _f(R, "R"), _f(t, "t"), 0 if grad_out is None else _f(grad_out, "grad_out"), _f(out, "out"))


def mask_iou_wrapper(pb, n, k, mask1, mask2, counts, iou):
    """IoU matrices of the arg-max segmentations (ogc_mask_iou)."""
    _run("ogc_mask_iou", mask1, pb, n, k, _f(mask1, "mask1"), _f(mask2, "mask2"), _i(counts, "counts"), _f(iou, "iou"))


def matched_distance_wrapper(pb, n, k, p, backward, mask1, mask2, col12, col21, grad12, grad21, out1, out2):
    """Per-point distances between matched masks, or their gradients (ogc_matched_distance)."""
    _run("ogc_matched_distance", mask1, pb, n, k, int(p), int(backward), _f(mask1, "mask1"), _f(mask2, "mask2"),
         _i(col12, "col12"), _i(col21, "col21"), 0 if grad12 is None else _f(grad12, "grad12"),
         0 if grad21 is None else _f(grad21, "grad21"), _f(out1, "out1"), _f(out2, "out2"))


def lsap_maximize_wrapper(np_, k, score, col4row):
    """Batched maximising linear-sum assignment with scipy's tie-breaking (ogc_lsap_maximize)."""
    _run("ogc_lsap_maximize", score, np_, k, _f(score, "score"), _i(col4row, "col4row"))


def sym_eigvals_wrapper(nb, k, A, w):
    """Ascending eigenvalues of (nb, k, k) symmetric float64 matrices (ogc_sym_eigvals)."""
    _run("ogc_sym_eigvals", A, nb, k, _check(A, torch.float64, "A"), _check(w, torch.float64, "w"))


def neighbour_consistency_fwd_wrapper(b, n, c, k, p, mask, idx, out):
    """out[i] = mean_j ||mask[i] - mask[idx[i, j]]||_p, mask (b, n, c) point-major (ogc_neighbour_consistency_fwd)."""
    _run("ogc_neighbour_consistency_fwd", mask, b, n, c, k, int(p), _f(mask, "mask"), _i(idx, "idx"), _f(out, "out"))


def reverse_neighbours_wrapper(b, n, k, idx, rev_start, rev_src, rev_mult, ws):
    """CSR of incoming edges of the neighbour lists idx (b, n, k) (ogc_reverse_neighbours)."""
    _run("ogc_reverse_neighbours", idx, b, n, k, _i(idx, "idx"), _i(rev_start, "rev_start"), _i(rev_src, "rev_src"),
         _i(rev_mult, "rev_mult"), _i(ws, "ws"))


def neighbour_consistency_bwd_wrapper(b, n, c, k, p, mask, idx, rev_start, rev_src, rev_mult, grad_out, grad_mask):
    """Gradient of neighbour_consistency_fwd w.r.t. mask (ogc_neighbour_consistency_bwd)."""
    _run("ogc_neighbour_consistency_bwd", mask, b, n, c, k, int(p), _f(mask, "mask"), _i(idx, "idx"),
         _i(rev_start, "rev_start"), _i(rev_src, "rev_src"), _i(rev_mult, "rev_mult"), _f(grad_out, "grad_out"),
         _f(grad_mask, "grad_mask"))


def group_norm_fwd_wrapper(b, c, hw, groups, eps, relu, x, gamma, beta, y, mean, rstd, ws):
    """Fused GroupNorm(+ReLU) forward (ogc_group_norm_fwd); ws: float64 scratch of 2*b*groups elements."""
    _run("ogc_group_norm_fwd", x, b, c, hw, groups, float(eps), int(relu), _f(x, "x"), _f(gamma, "gamma"),
         _f(beta, "beta"), _f(y, "y"), _f(mean, "mean"), _f(rstd, "rstd"), _check(ws, torch.float64, "ws"))


def group_norm_bwd_wrapper(b, c, hw, groups, relu, x, gamma, beta, mean, rstd, grad_y, grad_x, grad_gamma, grad_beta,
                           ws):
    """Fused GroupNorm(+ReLU) backward (ogc_group_norm_bwd); ws: float64 scratch of 2*b*c + b*groups elements."""
    h, (xp, gyp, gxp) = _acts((x, grad_y, grad_x), ("x", "grad_y", "grad_x"))
    _run("ogc_group_norm_bwd" + h, x, b, c, hw, groups, int(relu), xp, _f(gamma, "gamma"), _f(beta, "beta"),
         _f(mean, "mean"), _f(rstd, "rstd"), gyp, gxp,
         _f(grad_gamma, "grad_gamma"), _f(grad_beta, "grad_beta"), _check(ws, torch.float64, "ws"))


def group_norm_maxpool_fwd_wrapper(b, c, p, s, groups, eps, relu, x, gamma, beta, out, argmax, mean, rstd, ws):
    _run("ogc_group_norm_maxpool_fwd", x, b, c, p, s, groups, float(eps), int(relu), _f(x, "x"), _f(gamma, "gamma"),
         _f(beta, "beta"), _f(out, "out"), _i(argmax, "argmax"), _f(mean, "mean"), _f(rstd, "rstd"),
         _check(ws, torch.float64, "ws"))


def group_norm_maxpool_bwd_wrapper(b, c, p, s, groups, relu, x, gamma, mean, rstd, out, argmax, grad_out, grad_x,
                                   grad_gamma, grad_beta, ws):
    _run("ogc_group_norm_maxpool_bwd", x, b, c, p, s, groups, int(relu), _f(x, "x"), _f(gamma, "gamma"),
         _f(mean, "mean"), _f(rstd, "rstd"), _f(out, "out"), _i(argmax, "argmax"), _f(grad_out, "grad_out"),
         _f(grad_x, "grad_x"), _f(grad_gamma, "grad_gamma"), _f(grad_beta, "grad_beta"),
         _check(ws, torch.float64, "ws"))


def group_norm_maxpool_bwd_ext_wrapper(b, c, p, s, groups, relu, x, x_at_argmax, gamma, mean, rstd, out, argmax, grad_out,
                                       grad_x, grad_gamma, grad_beta, ws):
    """group_norm_maxpool_bwd_wrapper with x at the arg-max positions handed in (ogc_group_norm_maxpool_bwd_ext)."""
    _run("ogc_group_norm_maxpool_bwd_ext", x, b, c, p, s, groups, int(relu), _f(x, "x"), _f(x_at_argmax, "x_at_argmax"),
         _f(gamma, "gamma"), _f(mean, "mean"), _f(rstd, "rstd"), _f(out, "out"), _i(argmax, "argmax"),
         _f(grad_out, "grad_out"), _f(grad_x, "grad_x"), _f(grad_gamma, "grad_gamma"), _f(grad_beta, "grad_beta"),
         _check(ws, torch.float64, "ws"))


def conv1x1_wgrad_wrapper(b, cin, cout, hw, x, dy, dw):
    """dw[co, ci] = sum_{b,p} dy[b, co, p] x[b, ci, p] (ogc_conv1x1_wgrad); hw % 16 == 0."""
    if dy.dtype is torch.bfloat16:   # x fp32 (relative coordinates), dy in 16 bits
        _run("ogc_conv1x1_wgrad_xf_h", x, b, cin, cout, hw, _f(x, "x"), _check(dy, torch.bfloat16, "dy"), _f(dw, "dw"))
        return
    _run("ogc_conv1x1_wgrad", x, b, cin, cout, hw, _f(x, "x"), _f(dy, "dy"), _f(dw, "dw"))


def _opt(t, dtype, name):
    return 0 if t is None else _check(t, dtype, name)


def batch_norm_fwd_wrapper(b, c, hw, eps, relu, training, momentum, x, gamma, beta, running_mean, running_var, y, mean,
                           rstd, ws, stats, slots):
    """Fused BatchNorm(+ReLU) forward (ogc_batch_norm_fwd); running statistics updated in place when training."""
    _run("ogc_batch_norm_fwd", x, b, c, hw, float(eps), int(relu), int(training), float(momentum), _f(x, "x"),
         _f(gamma, "gamma"), _f(beta, "beta"), _opt(running_mean, torch.float32, "running_mean"),
         _opt(running_var, torch.float32, "running_var"), _f(y, "y"), _opt(mean, torch.float32, "mean"),
         _opt(rstd, torch.float32, "rstd"), _opt(ws, torch.float64, "ws"), _opt(stats, torch.float64, "stats"), int(slots))


def batch_norm_bwd_wrapper(b, c, hw, relu, training, x, gamma, beta, mean, rstd, grad_y, grad_x, grad_gamma, grad_beta,
                           ws):
    _run("ogc_batch_norm_bwd", x, b, c, hw, int(relu), int(training), _f(x, "x"), _f(gamma, "gamma"), _f(beta, "beta"),
         _f(mean, "mean"), _f(rstd, "rstd"), _f(grad_y, "grad_y"), _f(grad_x, "grad_x"), _f(grad_gamma, "grad_gamma"),
         _f(grad_beta, "grad_beta"), _check(ws, torch.float64, "ws"))


def batch_norm_maxpool_fwd_wrapper(b, c, p, s, eps, relu, training, momentum, x, gamma, beta, running_mean, running_var,
                                   out, argmax, mean, rstd, ws, stats, slots):
    _run("ogc_batch_norm_maxpool_fwd", x, b, c, p, s, float(eps), int(relu), int(training), float(momentum), _f(x, "x"),
         _f(gamma, "gamma"), _f(beta, "beta"), _opt(running_mean, torch.float32, "running_mean"),
         _opt(running_var, torch.float32, "running_var"), _f(out, "out"), _i(argmax, "argmax"),
         _opt(mean, torch.float32, "mean"), _opt(rstd, torch.float32, "rstd"), _opt(ws, torch.float64, "ws"),
         _opt(stats, torch.float64, "stats"), int(slots))


def batch_norm_maxpool_bwd_wrapper(b, c, p, s, relu, training, x, gamma, mean, rstd, out, argmax, grad_out, grad_x,
                                   grad_gamma, grad_beta, ws):
    _run("ogc_batch_norm_maxpool_bwd", x, b, c, p, s, int(relu), int(training), _f(x, "x"), _f(gamma, "gamma"),
         _f(mean, "mean"), _f(rstd, "rstd"), _f(out, "out"), _i(argmax, "argmax"), _f(grad_out, "grad_out"),
         _f(grad_x, "grad_x"), _f(grad_gamma, "grad_gamma"), _f(grad_beta, "grad_beta"), _check(ws, torch.float64, "ws"))


def group_norm_coeffs_wrapper(b, c, hw, groups, eps, x, gamma, beta, stats, slots, ws, mean, rstd, a, bb):
    """GroupNorm as a per-(b, channel) affine map, not applied (ogc_group_norm_coeffs)."""
    _run("ogc_group_norm_coeffs", gamma, b, c, hw, groups, float(eps), _opt(x, torch.float32, "x"), _f(gamma, "gamma"),
         _f(beta, "beta"), _opt(stats, torch.float64, "stats"), int(slots), _opt(ws, torch.float64, "ws"),
         _f(mean, "mean"), _f(rstd, "rstd"), _f(a, "a"), _f(bb, "bb"))


def conv1x1_gemm_affine_wrapper(b, M, K, hw, relu, groups, w, inp, pa, pb, out, stats):
    """Forward conv on act(pa * in + pb), optionally with the output's GroupNorm statistics (ogc_conv1x1_gemm_affine)."""
    h, (ip, op) = _acts((inp, out), ("in", "out"))
    _run("ogc_conv1x1_gemm_affine" + h, inp, b, M, K, hw, int(relu), int(groups), _f(w, "w"), ip, _f(pa, "pa"),
         _f(pb, "pb"), op, _opt(stats, torch.float64, "stats"))


def conv1x1_gemm_affine_pool_wrapper(b, M, K, hw, relu, groups, nsample, w, inp, pa, pb, next_gamma, out, stats, yext,
                                     aext):
    """conv1x1_gemm_affine_wrapper with statistics, plus per neighbourhood the extreme of the raw output (largest where
    next_gamma >= 0, smallest where it is negative) and its neighbour index (ogc_conv1x1_gemm_affine_pool);
    hw = centres * nsample, nsample in {16, 32, 64}."""
    h, (ip, op) = _acts((inp, out), ("in", "out"))
    _run("ogc_conv1x1_gemm_affine_pool" + h, inp, b, M, K, hw, int(relu), int(groups), int(nsample), _f(w, "w"),
         ip, _f(pa, "pa"), _f(pb, "pb"), _f(next_gamma, "next_gamma"), op,
         _check(stats, torch.float64, "stats"), _f(yext, "yext"), _i(aext, "aext"))


def group_norm_pool_extremes_wrapper(b, c, p, s, groups, eps, relu, yext, aext, gamma, beta, out, argmax, mean, rstd,
                                     stats, slots):
    """max over the neighbourhood of act(GroupNorm(x)) from the extremes of x (ogc_group_norm_pool_extremes)."""
    _run("ogc_group_norm_pool_extremes", yext, b, c, p, s, groups, float(eps), int(relu), _f(yext, "yext"),
         _i(aext, "aext"), _f(gamma, "gamma"), _f(beta, "beta"), _f(out, "out"), _i(argmax, "argmax"), _f(mean, "mean"),
         _f(rstd, "rstd"), _check(stats, torch.float64, "stats"), int(slots))


def conv1x1_wgrad_affine_wrapper(b, cin, cout, hw, relu, x, pa, pb, dy, dw):
    """Weight gradient with the operand act(pa * x + pb) recomputed on load (ogc_conv1x1_wgrad_affine)."""
    h, (xp, dyp) = _acts((x, dy), ("x", "dy"))
    _run("ogc_conv1x1_wgrad_affine" + h, x, b, cin, cout, hw, int(relu), xp, _f(pa, "pa"), _f(pb, "pb"), dyp, _f(dw, "dw"))


def _rows(t, name):
    """(pointer, row stride) of a (b, l, w) fp32 operand read in place from a packed projection output: a column slice
    of a contiguous (b, l, W) tensor (unit stride along the last axis, rows W apart, samples l*W apart)."""
    if not isinstance(t, torch.Tensor) or t.device.type != "cuda" or t.dtype != torch.float32 or t.dim() != 3:
        raise RuntimeError("%s must be a 3-D float32 CUDA tensor" % name)
    ld = t.stride(1)
    if t.stride(2) != 1 or ld < t.shape[2] or (t.shape[0] > 1 and t.stride(0) != t.shape[1] * ld):
        raise RuntimeError("%s must be a column slice of a contiguous (b, l, W) tensor" % name)
    return t.data_ptr(), ld


def attention_fwd_wrapper(h, scale, q, k, v, out, prob):
    """out = softmax(scale * Q K^T) V per (sample, head) (ogc_attention_fwd).  q (b, lq, e), k / v (b, lk, e) may be
    column slices of packed projections; out (b, lq, e) and prob (b, h, lq, lk) contiguous."""
    b, lq, e = q.shape
    lk = k.shape[1]
    (qp, ldq), (kp, ldk), (vp, ldv) = _rows(q, "q"), _rows(k, "k"), _rows(v, "v")
    _run("ogc_attention_fwd", q, b, lq, lk, h, e // h, float(scale), qp, ldq, kp, ldk, vp, ldv, _f(out, "out"),
         _f(prob, "prob"))


def attention_bwd_wrapper(h, scale, q, k, v, out, prob, dout, dq, dk, dv):
    """Gradients of attention_fwd_wrapper (ogc_attention_bwd); dq / dk / dv may be column slices of packed buffers."""
    b, lq, e = q.shape
    lk = k.shape[1]
    (qp, ldq), (kp, ldk), (vp, ldv) = _rows(q, "q"), _rows(k, "k"), _rows(v, "v")
    (dqp, lddq), (dkp, lddk), (dvp, lddv) = _rows(dq, "dq"), _rows(dk, "dk"), _rows(dv, "dv")
    _run("ogc_attention_bwd", q, b, lq, lk, h, e // h, float(scale), qp, ldq, kp, ldk, vp, ldv, _f(out, "out"),
         _f(prob, "prob"), _f(dout, "dout"), dqp, lddq, dkp, lddk, dvp, lddv)


def conv1x1_wgrad_moments_wrapper(b, cin, cout, hw, relu, y_prev, pa, pb, grad_y, moments):
    h, (yp, gp) = _acts((y_prev, grad_y), ("y_prev", "grad_y"))
    _run("ogc_conv1x1_wgrad_moments" + h, y_prev, b, cin, cout, hw, int(relu), yp, _f(pa, "pa"), _f(pb, "pb"), gp,
         _f(moments, "moments"))


def gn_moments_combine_wrapper(b, cin, cout, hw, groups, moments, w, pa, pb, mean, rstd, gamma, grad_w, coef, gw, gb):
    _run("ogc_gn_moments_combine", moments, b, cin, cout, hw, groups, _f(moments, "moments"), _f(w, "w"), _f(pa, "pa"),
         _f(pb, "pb"), _f(mean, "mean"), _f(rstd, "rstd"), _f(gamma, "gamma"), _f(grad_w, "grad_w"), _f(coef, "coef"),
         _f(gw, "grad_gamma"), _f(gb, "grad_beta"))


def conv1x1_dgrad_adjoint_wrapper(b, cin, cout, hw, relu, w, grad_y, y_prev, pa, pb, coef, grad_prev):
    h, (gp, yp, op) = _acts((grad_y, y_prev, grad_prev), ("grad_y", "y_prev", "grad_prev"))
    _run("ogc_conv1x1_dgrad_adjoint" + h, grad_y, b, cin, cout, hw, int(relu), _f(w, "w"), gp, yp, _f(pa, "pa"), _f(pb, "pb"),
         _f(coef, "coef"), op)


def group_norm_maxpool_bwd_sparse_wrapper(b, c, p, s, groups, relu, x, gamma, mean, rstd, out, argmax, grad_out, coef2, inj,
                                          grad_gamma, grad_beta, ws, x_at_argmax=None):
    """coef2 (b, c, 2), inj (b, c, p, 2): the gradient of the pooled GroupNorm w.r.t. x in sparse form
    (ogc_group_norm_maxpool_bwd_sparse)."""
    h, (xp,) = _acts((x,), ("x",))
    _run("ogc_group_norm_maxpool_bwd_sparse" + h, x, b, c, p, s, groups, int(relu), xp,
         _opt(x_at_argmax, torch.float32, "x_at_argmax"), _f(gamma, "gamma"),
         _f(mean, "mean"), _f(rstd, "rstd"), _f(out, "out"), _i(argmax, "argmax"), _f(grad_out, "grad_out"),
         _f(coef2, "coef2"), _f(inj, "inj"), _f(grad_gamma, "grad_gamma"), _f(grad_beta, "grad_beta"),
         _check(ws, torch.float64, "ws"))


def conv1x1_wgrad_moments_pooled_wrapper(b, cin, cout, hw, relu, nsample, y_prev, pa, pb, y, coef2, inj, moments):
    h, (ypp, yp) = _acts((y_prev, y), ("y_prev", "y"))
    _run("ogc_conv1x1_wgrad_moments_pooled" + h, y_prev, b, cin, cout, hw, int(relu), nsample, ypp, _f(pa, "pa"),
         _f(pb, "pb"), yp, _f(coef2, "coef2"), _f(inj, "inj"), _f(moments, "moments"))


def conv1x1_dgrad_adjoint_pooled_wrapper(b, cin, cout, hw, relu, nsample, w, y, coef2, inj, y_prev, pa, pb, coef, grad_prev):
    h, (yp, ypp, op) = _acts((y, y_prev, grad_prev), ("y", "y_prev", "grad_prev"))
    _run("ogc_conv1x1_dgrad_adjoint_pooled" + h, y, b, cin, cout, hw, int(relu), nsample, _f(w, "w"), yp,
         _f(coef2, "coef2"), _f(inj, "inj"), ypp, _f(pa, "pa"), _f(pb, "pb"), _f(coef, "coef"), op)


def mlp_chain_pool_supported(c0, c1, c2, c3, nsample):
    """Is there a fused inference kernel for the MLP c0 -> c1 -> c2 [-> c3] followed by the max over nsample?"""
    return bool(_lib.load().ogc_mlp_chain_pool_supported(c0, c1, c2, c3, nsample))


def mlp_chain_pool_wrapper(x, wts, biases, out):
    """out (B, c_last, P) = max over the neighbourhood of the folded MLP of x (B, c0, P, S) (ogc_mlp_chain_pool).
    wts: transposed folded weights (rows padded to a multiple of 4), biases: folded biases; two or three layers."""
    B, c0, P, S = x.shape
    c = [w.shape[1] for w in wts] + [0]
    _run("ogc_mlp_chain_pool", x, B, c0, c[0], c[1], c[2] if len(wts) > 2 else 0, P, S, _f(x, "x"),
         _f(wts[0], "wt1"), _f(biases[0], "b1"), _f(wts[1], "wt2"), _f(biases[1], "b2"),
         _f(wts[2], "wt3") if len(wts) > 2 else None, _f(biases[2], "b3") if len(wts) > 2 else None, _f(out, "out"))


def corr_layer_pool_supported(cf, c1, c2, c3, nsample):
    return bool(_lib.load().ogc_corr_layer_pool_supported(cf, c1, c2, c3, nsample))


def corr_layer_pool_wrapper(pos1, pos2, feat1, feat2, idx, wts, biases, out):
    """FlowEmbedding after its neighbour search in one launch (ogc_corr_layer_pool): pos (B, 3, n), feat (B, cf, n),
    idx (B, n1, S) int32, folded transposed weights / biases of the three layers; out (B, c3, n1)."""
    B, cf, n1 = feat1.shape
    _run("ogc_corr_layer_pool", feat1, B, cf, wts[0].shape[1], wts[1].shape[1], wts[2].shape[1], n1, feat2.shape[2],
         idx.shape[2], _f(pos1, "pos1"), _f(pos2, "pos2"), _f(feat1, "feat1"), _f(feat2, "feat2"), _i(idx, "idx"),
         _f(wts[0], "wt1"), _f(biases[0], "b1"), _f(wts[1], "wt2"), _f(biases[1], "b2"), _f(wts[2], "wt3"),
         _f(biases[2], "b3"), _f(out, "out"))


def small_linear_fwd_wrapper(x, weight, bias, y):
    """y (rows, n_out) = x (rows, n_in) weight^T + bias (ogc_small_linear_fwd); bias may be None."""
    rows, n_in = x.shape
    _run("ogc_small_linear_fwd", x, rows, n_in, weight.shape[0], _f(x, "x"), _f(weight, "weight"),
         None if bias is None else _f(bias, "bias"), _f(y, "y"))


def small_linear_bwd_wrapper(x, weight, grad_y, grad_x, grad_weight, grad_bias):
    """grad_x / grad_weight / grad_bias of small_linear_fwd_wrapper in one launch; any of them may be None."""
    rows, n_in = x.shape
    _run("ogc_small_linear_bwd", x, rows, n_in, weight.shape[0], _f(x, "x"), _f(weight, "weight"), _f(grad_y, "grad_y"),
         None if grad_x is None else _f(grad_x, "grad_x"), None if grad_weight is None else _f(grad_weight, "grad_weight"),
         None if grad_bias is None else _f(grad_bias, "grad_bias"))


def slot_masks_fwd_wrapper(temperature, feats, slots, mask):
    """mask (b, n, k) = softmax_k(normalize(feats, 1)^T normalize(slots, 1) / temperature) (ogc_slot_masks_fwd);
    feats (b, d, n), slots (b, d, k)."""
    b, d, n = feats.shape
    _run("ogc_slot_masks_fwd", feats, b, d, n, slots.shape[2], float(temperature), _f(feats, "feats"),
         _f(slots, "slots"), _f(mask, "mask"))


def slot_masks_bwd_wrapper(temperature, feats, slots, mask, grad_mask, grad_feats, grad_slots):
    """Gradients of slot_masks_fwd_wrapper (ogc_slot_masks_bwd); the scratch is allocated here."""
    b, d, n = feats.shape
    k = slots.shape[2]
    ws = torch.empty(max(int(_lib.load().ogc_slot_masks_ws_floats(b, d, n, k)), 1), dtype=torch.float32,
                     device=feats.device)
    _run("ogc_slot_masks_bwd", feats, b, d, n, k, float(temperature), _f(feats, "feats"), _f(slots, "slots"),
         _f(mask, "mask"), _f(grad_mask, "grad_mask"), _f(grad_feats, "grad_feats"), _f(grad_slots, "grad_slots"),
         _f(ws, "ws"))


def group_norm_ws(b, c, groups, backward, device):
    """Scratch of the GroupNorm entry points: per-slice partial sums, 2*b*groups*ogc_group_norm_stats_slots() doubles
    for a forward statistics pass, 2*b*c*ogc_group_norm_bwd_slots() for the backward pass (include/ogc_ops.h)."""
    L = _lib.load()
    n = 2 * b * c * L.ogc_group_norm_bwd_slots() if backward else 2 * b * groups * L.ogc_group_norm_stats_slots()
    return torch.empty(max(n, 1), dtype=torch.float64, device=device)


_precision = "fp32"


def set_matmul_precision(mode):
    """'fp32' (default, exact) or 'bf16' operands for the 1x1-convolution kernels (ogc_set_matmul_precision);
    returns the previous mode."""
    global _precision
    if mode not in ("fp32", "bf16"):
        raise ValueError("matmul precision must be 'fp32' or 'bf16'")
    previous = "bf16" if _lib.load().ogc_set_matmul_precision(1 if mode == "bf16" else 0) else "fp32"
    _precision = mode
    return previous


def get_matmul_precision():
    return _precision


def conv1x1_gemm_stream_supported(b, m, k, hw):
    """ogc_conv1x1_gemm_stream_supported: the plain product of this shape runs on the streaming MFMA kernel."""
    return bool(_lib.load().ogc_conv1x1_gemm_stream_supported(int(b), int(m), int(k), int(hw)))


def conv1x1_gemm_stats_supported(b, m, k, hw, affine):
    """Can the forward convolution of this shape also produce the next GroupNorm's statistics?"""
    return bool(_lib.load().ogc_conv1x1_gemm_stats_supported(int(b), int(m), int(k), int(hw), 1 if affine else 0))


def conv1x1_gn_slots():
    """Number of accumulator copies conv1x1_gemm_gnstats_wrapper fills (ogc_conv1x1_gn_slots)."""
    return _lib.load().ogc_conv1x1_gn_slots()


def conv1x1_gemm_gnstats_wrapper(b, M, K, hw, groups, w, inp, out, stats):
    """Forward 1x1 convolution that also accumulates the following GroupNorm's statistics (ogc_conv1x1_gemm_gnstats);
    stats: float64, conv1x1_gn_slots() * b * groups * 2 elements."""
    _run("ogc_conv1x1_gemm_gnstats", inp, b, M, K, hw, groups, _f(w, "w"), _f(inp, "in"), _f(out, "out"),
         _check(stats, torch.float64, "stats"))


def group_norm_fwd_stats_wrapper(b, c, hw, groups, eps, relu, x, gamma, beta, y, mean, rstd, stats, slots):
    _run("ogc_group_norm_fwd_stats", x, b, c, hw, groups, float(eps), int(relu), _f(x, "x"), _f(gamma, "gamma"),
         _f(beta, "beta"), _f(y, "y"), _f(mean, "mean"), _f(rstd, "rstd"), _check(stats, torch.float64, "stats"),
         int(slots))


def group_norm_maxpool_fwd_stats_wrapper(b, c, p, s, groups, eps, relu, x, gamma, beta, out, argmax, mean, rstd, stats,
                                         slots):
    _run("ogc_group_norm_maxpool_fwd_stats", x, b, c, p, s, groups, float(eps), int(relu), _f(x, "x"),
         _f(gamma, "gamma"), _f(beta, "beta"), _f(out, "out"), _i(argmax, "argmax"), _f(mean, "mean"),
         _f(rstd, "rstd"), _check(stats, torch.float64, "stats"), int(slots))


def conv1x1_gemm_wrapper(b, M, K, hw, transpose_a, w, inp, out):
    """out[b, m, p] = sum_k A[m, k] in[b, k, p], A = w or w^T (ogc_conv1x1_gemm); hw % 64 == 0, K <= 160."""
    h, (ip, op) = _acts((inp, out), ("in", "out"))
    _run("ogc_conv1x1_gemm" + h, inp, b, M, K, hw, int(transpose_a), _f(w, "w"), ip, op)


def conv1x1_gemm_any_wrapper(b, M, K, hw, transpose_a, w, inp, out):
    """out[b, m, p] = sum_k A[m, k] in[b, k, p], A = w or w^T, any K and M (ogc_conv1x1_gemm_any); hw % 64 == 0."""
    _run("ogc_conv1x1_gemm_any", inp, b, M, K, hw, int(transpose_a), _f(w, "w"), _f(inp, "in"), _f(out, "out"))


def conv1x1_wgrad_affine_pooled_wrapper(b, cin, cout, hw, relu, nsample, x, pa, pb, y, coef2, inj, dw):
    """ogc_conv1x1_wgrad_affine with dy in the sparse form (y, coef2, inj) of group_norm_maxpool_bwd_sparse_wrapper."""
    h, (xp, yp) = _acts((x, y), ("x", "y"))
    _run("ogc_conv1x1_wgrad_affine_pooled" + h, x, b, cin, cout, hw, int(relu), nsample, xp, _f(pa, "pa"), _f(pb, "pb"),
         yp, _f(coef2, "coef2"), _f(inj, "inj"), _f(dw, "dw"))


def conv1x1_dgrad_pooled_wrapper(b, cin, cout, hw, nsample, w, y, coef2, inj, grad_z):
    """grad_z[b] = w^T . g_y[b] with g_y rebuilt from (y, coef2, inj) on load (ogc_conv1x1_dgrad_pooled)."""
    h, (yp, gp) = _acts((y, grad_z), ("y", "grad_z"))
    _run("ogc_conv1x1_dgrad_pooled" + h, y, b, cin, cout, hw, nsample, _f(w, "w"), yp, _f(coef2, "coef2"), _f(inj, "inj"), gp)

