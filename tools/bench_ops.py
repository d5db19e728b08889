"""Per-operator timings on one MI355X at the config shapes of SURVEY.md §8 (development tool).

    python tools/bench_ops.py [--ops knn,ball,...] [--iters 20]

Prints one line per case: ms per launch, algorithmic GB/s (SURVEY §8d byte counts) and, for the
all-pairs ops, G pair-evaluations/s.
"""
import argparse
import sys, os

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ogc_amd  # noqa: E402
from ogc_amd import pointnet2_cuda as nat  # noqa: E402

DEV = "cuda"


def cloud(B, N, g, scale=(60.0, 4.0, 80.0)):
    return ((torch.rand(B, N, 3, generator=g) - 0.5) * torch.tensor(scale)).to(DEV).contiguous()


def timeit(fn, iters, warm=3):
    """Median of three timed batches of `iters` back-to-back calls.  (One batch was what produced the 14.3-ms FPS row and the
    `nan` of profiles/r03_ops.txt: a first launch of a kernel instantiation — code-object load, LDS-limit attribute — inside the
    only timed batch, and a column printed for a product nobody had timed.)"""
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    reads = []
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        reads.append(e0.elapsed_time(e1) / iters)
    return sorted(reads)[1]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ops", default="knn,nn3,ball,fps,group,interp,conv,gn")
    ap.add_argument("--iters", type=int, default=10)
    a = ap.parse_args()
    ops = a.ops.split(",")
    g = torch.Generator().manual_seed(1234)
    print("device:", torch.cuda.get_device_name(0))

    if "knn" in ops:
        for (B, n, m, k) in [(1, 8192, 8192, 32), (4, 8192, 8192, 32), (16, 8192, 8192, 32), (16, 2048, 8192, 64), (16, 1024, 2048, 64),
                             (16, 512, 1024, 64), (1, 2048, 2048, 16), (1, 4096, 8192, 32), (1, 8192, 8192, 1),
                             (8, 16384, 16384, 32)]:
            pc = cloud(B, m, g)
            q = pc[:, :: m // n].contiguous() if n <= m else cloud(B, n, g)
            d2 = torch.empty(B, n, k, device=DEV)
            idx = torch.empty(B, n, k, dtype=torch.int32, device=DEV)
            ms = timeit(lambda: nat.knn_wrapper(B, n, m, k, q, pc, d2, idx), a.iters)
            byt = B * (12 * n + 12 * m + 8 * n * k)
            print("knn   B=%-3d n=%-6d m=%-6d k=%-3d %9.3f ms  %8.2f GB/s  %8.1f Gpair/s" %
                  (B, n, m, k, ms, byt / ms / 1e6, B * n * m / ms / 1e6))
    if "knnc" in ops:   # kNN + radius clamp (ogc_knn_clamped) at the C4 shapes: loss term and the three SA levels
        for (B, n, m, k, r) in [(16, 8192, 8192, 32, 1.0), (16, 2048, 8192, 64, 2.0), (16, 1024, 2048, 64, 4.0),
                                (16, 512, 1024, 64, 8.0), (4, 16384, 16384, 32, 1.0), (16, 8192, 8192, 32, -1.0)]:
            pc = cloud(B, m, g)
            q = pc[:, :: m // n].contiguous()
            d2 = torch.empty(B, n, k, device=DEV)
            idx = torch.empty(B, n, k, dtype=torch.int32, device=DEV)
            ms = timeit(lambda: nat.knn_clamped_wrapper(B, n, m, k, r, q, pc, d2, idx), a.iters)
            byt = B * (12 * n + 12 * m + 8 * n * k)
            print("knnc  B=%-3d n=%-6d m=%-6d k=%-3d r=%-4.1f %9.3f ms  %8.2f GB/s" % (B, n, m, k, r, ms, byt / ms / 1e6))
    if "nn3" in ops:
        for (B, n, m) in [(16, 8192, 2048), (16, 2048, 1024), (1, 100000, 8192)]:
            pc = cloud(B, m, g)
            q = cloud(B, n, g)
            d2 = torch.empty(B, n, 3, device=DEV)
            idx = torch.empty(B, n, 3, dtype=torch.int32, device=DEV)
            ms = timeit(lambda: nat.three_nn_wrapper(B, n, m, q, pc, d2, idx), a.iters)
            byt = B * (12 * n + 12 * m + 24 * n)
            print("nn3   B=%-3d n=%-6d m=%-6d       %9.3f ms  %8.2f GB/s  %8.1f Gpair/s" %
                  (B, n, m, ms, byt / ms / 1e6, B * n * m / ms / 1e6))
    if "ball" in ops:
        for (B, n, ns, r) in [(1, 8192, 64, 2.0), (4, 8192, 64, 2.0), (16, 8192, 64, 2.0), (8, 16384, 64, 2.0),
                              (16, 4096, 16, 2.4)]:
            pc = cloud(B, n, g)
            idx = torch.zeros(B, n, ns, dtype=torch.int32, device=DEV)
            ms = timeit(lambda: nat.ball_query_wrapper(B, n, n, r, ns, pc, pc, idx), a.iters)
            byt = B * (12 * n + 12 * n + 4 * n * ns)
            print("ball  B=%-3d M=N=%-6d ns=%-3d r=%-4.1f  %9.3f ms  %8.2f GB/s  %8.1f Gpair/s" %
                  (B, n, ns, r, ms, byt / ms / 1e6, B * n * n / ms / 1e6))
    if "fps" in ops:
        for (B, N, m) in [(16, 8192, 2048), (16, 2048, 1024), (16, 1024, 512), (1, 2048, 2048), (1, 8192, 4096),
                          (8, 16384, 4096), (1, 100000, 8192)]:
            pc = cloud(B, N, g)
            idx = torch.empty(B, m, dtype=torch.int32, device=DEV)
            temp = torch.empty(B, N, device=DEV)

            def run():
                temp.fill_(1e10)
                nat.furthest_point_sampling_wrapper(B, N, m, pc, temp, idx)
            ms = timeit(run, max(2, a.iters // 3), warm=1)
            print("fps   B=%-3d N=%-6d m=%-6d       %9.3f ms  %8.3f us/round" % (B, N, m, ms, ms * 1e3 / m))
    if "group" in ops:
        for (B, C, N, P, S) in [(16, 96, 2048, 1024, 64), (16, 128, 1024, 512, 64), (16, 3, 8192, 2048, 64),
                                (4, 10, 8192, 8192, 64), (1, 64, 2048, 2048, 16)]:
            feats = torch.randn(B, C, N, device=DEV)
            idx = torch.randint(0, N, (B, P, S), dtype=torch.int32, device=DEV)
            out = torch.empty(B, C, P, S, device=DEV)
            ms = timeit(lambda: nat.group_points_wrapper(B, C, N, P, S, feats, idx, out), a.iters)
            byt = B * (4 * C * N + 4 * P * S + 4 * C * P * S)
            gp = torch.zeros(B, C, N, device=DEV)
            ms2 = timeit(lambda: nat.group_points_grad_wrapper(B, C, N, P, S, out, idx, gp), a.iters)
            from ogc_amd import fused
            rev = fused.group_reverse(idx, N)
            ms3 = ms4 = float("nan")
            if rev is not None:
                ms3 = timeit(lambda: fused.group_reverse(idx, N), a.iters)
                ms4 = timeit(lambda: nat.group_points_grad_rev_wrapper(B, C, N, P, S, out, rev[0], rev[1], rev[2], gp), a.iters)
            print("group B=%-3d C=%-4d N=%-5d P=%-5d S=%-3d fwd %8.3f ms %8.1f GB/s | bwd (atomics) %8.3f ms %8.1f GB/s | "
                  "bwd (gather) %8.3f ms %8.1f GB/s + lists %6.3f ms once per neighbour tensor" %
                  (B, C, N, P, S, ms, byt / ms / 1e6, ms2, byt / ms2 / 1e6, ms4, byt / ms4 / 1e6, ms3))
    if "interp" in ops:
        for (B, C, M, N) in [(16, 256, 512, 1024), (16, 128, 1024, 2048), (16, 64, 2048, 8192)]:
            feats = torch.randn(B, C, M, device=DEV)
            i3 = torch.randint(0, M, (B, N, 3), dtype=torch.int32, device=DEV)
            w = torch.rand(B, N, 3, device=DEV)
            out = torch.empty(B, C, N, device=DEV)
            ms = timeit(lambda: nat.three_interpolate_wrapper(B, C, M, N, feats, i3, w, out), a.iters)
            byt = B * (4 * C * M + 24 * N + 4 * C * N)
            gp = torch.zeros(B, C, M, device=DEV)
            ms2 = timeit(lambda: nat.three_interpolate_grad_wrapper(B, C, N, M, out, i3, w, gp), a.iters)
            from ogc_amd import fused
            rev = fused.group_reverse(i3, M)
            ms3 = timeit(lambda: nat.three_interpolate_grad_rev_wrapper(B, C, N, M, out, w, rev[0], rev[1], rev[2], gp), a.iters)
            print("interp B=%-3d C=%-4d M=%-5d N=%-5d fwd %8.3f ms %8.1f GB/s | bwd (atomics) %8.3f ms %8.1f GB/s | "
                  "bwd (gather) %8.3f ms %8.1f GB/s" % (B, C, M, N, ms, byt / ms / 1e6, ms2, byt / ms2 / 1e6, ms3, byt / ms3 / 1e6))


def bench_conv_gn(ops, iters):
    if "conv" in ops:
        slots = nat.conv1x1_gn_slots()
        for (B, cin, cout, hw) in [(16, 6, 32, 131072), (16, 32, 32, 131072), (16, 32, 64, 131072), (16, 99, 64, 65536),
                                   (16, 64, 64, 65536), (16, 64, 128, 65536), (16, 131, 128, 32768), (16, 128, 128, 32768),
                                   (16, 128, 256, 32768)]:
            x = torch.randn(B, cin, hw, device=DEV)
            w = torch.randn(cout, cin, device=DEV)
            y, y2 = torch.empty(B, cout, hw, device=DEV), torch.empty(B, cout, hw, device=DEV)
            dx = torch.empty_like(x)
            dw = torch.empty(cout, cin, device=DEV)
            byt = 4.0 * B * hw * (cin + cout)
            f = timeit(lambda: nat.conv1x1_gemm_wrapper(B, cout, cin, hw, 0, w, x, y), iters)
            # (ogc_conv1x1_gemm takes K <= 160; the one wider input gradient of the step, 256 -> 128 at SA3, runs on the chunked
            # kernel, ogc_conv1x1_gemm_any)
            lib_dgrad = cout > 160
            d = (timeit(lambda: nat.conv1x1_gemm_any_wrapper(B, cin, cout, hw, 1, w, y, dx), iters) if lib_dgrad else
                 timeit(lambda: nat.conv1x1_gemm_wrapper(B, cin, cout, hw, 1, w, y, dx), iters))
            g = timeit(lambda: nat.conv1x1_wgrad_wrapper(B, cin, cout, hw, x, y, dw), iters)
            print("conv  B=%-3d %4d->%-4d hw=%-6d fwd %7.3f ms %6.0f GB/s | dgrad%s %7.3f ms %6.0f GB/s | wgrad %7.3f ms %6.0f GB/s" %
                  (B, cin, cout, hw, f, byt / f / 1e6, " (chunked)" if lib_dgrad else "", d, byt / d / 1e6, g, byt / g / 1e6))
            # variants of the same layer: GroupNorm statistics in the epilogue (offered up to K = 100) and the previous
            # layer's GroupNorm + ReLU folded into the operand load
            gamma, beta = torch.ones(cout, device=DEV), torch.zeros(cout, device=DEV)
            mean, rstd = torch.empty(B * 4, device=DEV), torch.empty(B * 4, device=DEV)
            ws = nat.group_norm_ws(B, cout, 4, False, DEV)
            gn_full = timeit(lambda: nat.group_norm_fwd_wrapper(B, cout, hw, 4, 1e-5, 1, y, gamma, beta, y2, mean, rstd, ws), iters)
            a, bb = torch.rand(B * cin, device=DEV), torch.randn(B * cin, device=DEV)
            fa = timeit(lambda: nat.conv1x1_gemm_affine_wrapper(B, cout, cin, hw, 1, 0, w, x, a, bb, y, None), iters)
            ga = timeit(lambda: nat.conv1x1_wgrad_affine_wrapper(B, cin, cout, hw, 1, x, a, bb, y, dw), iters)
            line = "      input norm folded in: fwd %+.3f, wgrad %+.3f ms" % (fa - f, ga - g)
            if cin <= 100:
                st = torch.empty(slots * B * 4 * 2, dtype=torch.float64, device=DEV)
                fs = timeit(lambda: nat.conv1x1_gemm_gnstats_wrapper(B, cout, cin, hw, 4, w, x, y, st), iters)
                gn_apply = timeit(lambda: nat.group_norm_fwd_stats_wrapper(B, cout, hw, 4, 1e-5, 1, y, gamma, beta, y2, mean,
                                                                           rstd, st, slots), iters)
                line += " | output statistics in the epilogue: %+.3f ms vs %.3f ms for the separate pass" % (fs - f, gn_full - gn_apply)
            print(line + " | GroupNorm+ReLU of the output (stats + apply): %.3f ms" % gn_full)
    if "gn" in ops:
        from ogc_amd.fused import group_norm_act, group_norm_act_maxpool
        for shape in [(16, 32, 2048, 64), (16, 64, 1024, 64), (16, 128, 512, 64), (16, 256, 512, 64)]:
            gn = torch.nn.GroupNorm(4, shape[1]).to(DEV)
            x = torch.randn(*shape, device=DEV, requires_grad=True)
            y = group_norm_act(x, gn, True)
            g = torch.randn_like(y)
            f = timeit(lambda: group_norm_act(x, gn, True), iters)
            bw = timeit(lambda: torch.autograd.grad(group_norm_act(x, gn, True), x, g), iters) - f
            nbytes = 4.0 * x.numel()
            print("gn    %-22s fwd %7.3f ms (%5.0f GB/s of 3 passes) | bwd %7.3f ms (%5.0f GB/s of 5 passes)" %
                  (str(shape), f, 3 * nbytes / f / 1e6, bw, 5 * nbytes / bw / 1e6))


if __name__ == "__main__":
    main()
    import argparse as _ap
    _a = _ap.ArgumentParser(); _a.add_argument("--ops", default="knn,nn3,ball,fps,group,interp,conv,gn"); _a.add_argument("--iters", type=int, default=10)
    _args = _a.parse_args()
    bench_conv_gn(_args.ops.split(","), _args.iters)
