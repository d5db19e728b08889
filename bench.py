"""Headline benchmark: unsupervised OGC segmentation training throughput on 8192-point KITTI-SF-shaped scenes.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = `Trainer._train_it` of the reference (train_seg.py:47-86) on config C4 (SURVEY.md §8,
config/seg/kittisf/kittisf_unsup.yaml): per GPU 4 samples x 4 views (2 frames + 2 augmented) of 8192 points,
MaskFormer3D(segnet_kitti) forward on 16 clouds, UnsupervisedOGCLoss with all three terms active, backward,
NaN-gradient check, Adam step.  Inputs are synthetic and seeded: four distinct batches, resident in HBM, rotated through the
timed steps (the same steps with the host-to-device copy of every batch inside: `ms_per_step_with_h2d`).  Weak scaling: every
rank has its own batch; gradients are averaged over RCCL by one flat ~2.4 MB all-reduce per step (utils/dist_util.py).

Rank 0 prints ONE JSON line: value = whole-job point-clouds/s.  `roofline` is measured live (HIP events on the
launch stream, inside the timed steps) for the ball-query kernel, the kernel BASELINE.json's metric names;
`cpu_baseline` is the same step on the host cores with the CPU oracle's operators, on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E peak (MI355X_MICROARCH.md)
FP32_MFMA_PEAK_TF = 157.3  # v_mfma_f32_16x16x4_f32 / 32x32x2_f32: the fp32 matrix rate = the fp32 vector rate (same guide)
# Numbers NOT measured by this run: PMC counter readings of earlier profiling passes (rocprofv3 --pmc cannot run inside the
# timed region; `roofline.traffic` is the one such field the contract asks for).  They are READ from the tracked files under
# profiles/ that hold them — the newest round's — when the line is printed: every number under `offline` is a line of the file
# its `source` names, or absent (with the reason) when that file is missing or does not parse.
def _newest_profile(stem):
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_%s.txt" % stem)))
    return files[-1] if files else None


def _parse_op_pmc(path):
    """tools/pmc_op2.sh output: '<kernel[:60]> calls N median X us avg Y us min .. max ..' lines (files of round 4 and before:
    no median, 'avg' is taken) and '<kernel[:40]> COUNTER v1 v2 ..' lines (one value per profiled launch).
    -> {kernel: {"avg_us": the median launch duration, COUNTER: median, ...}}"""
    import re
    out = {}
    for line in open(path):
        m = re.match(r"^(.{1,60}?)\s+calls\s+(\d+)\s+(?:median|avg)\s+([0-9.]+)\s+us", line)
        if m:
            out.setdefault(m.group(1).strip()[:40], {})["avg_us"] = float(m.group(3))   # (the counter lines carry 40 characters)
            continue
        m = re.match(r"^(.{1,42}?)\s+([A-Z][A-Z0-9_a-z]+)\s+((?:[-+0-9.e]+\s*)+)$", line)
        if m:
            vals = sorted(float(v) for v in m.group(3).split())
            out.setdefault(m.group(1).strip()[:40], {})[m.group(2)] = vals[len(vals) // 2]
    return out


def _kernel(table, *parts):
    for name, row in table.items():
        if all(p in name for p in parts):
            return name, row
    raise KeyError("no kernel matching %r" % (parts,))


def load_offline():
    off = {}
    rel = lambda p: os.path.relpath(p, ROOT)  # noqa: E731
    path = _newest_profile("ball_query_pmc")
    try:
        if path is None:
            raise FileNotFoundError("profiles/rNN_ball_query_pmc.txt")
        t = _parse_op_pmc(path)
        qn, q = _kernel(t, "ball_query_cells_kernel")
        bn, b = _kernel(t, "grid_build")
        kib = q["FETCH_SIZE"] + q["WRITE_SIZE"] + b["FETCH_SIZE"] + b["WRITE_SIZE"]
        alg = 16 * (12 * 8192 + 12 * 8192 + 4 * 8192 * 64)
        off["ball_query_traffic_bytes"] = {
            "shape": [16, 8192, 8192, 64], "bytes": int(kib * 1024),
            "source": "%s (median FETCH_SIZE + WRITE_SIZE of %s + %s over the profiled launches, separate rocprofv3 --pmc passes; "
                      "FETCH_SIZE as reported — these kernels issue 12-16 byte gathers, not the wide streams the x2 gfx950 correction "
                      "applies to)" % (rel(path), bn, qn)}
        off["ball_query_kernels_us"] = {
            qn: q["avg_us"], bn: b["avg_us"],
            "query_plus_half_build_frac_of_hbm_peak": round(alg / ((q["avg_us"] + b["avg_us"] / 2) * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
            "query_alone_frac_of_hbm_peak": round(alg / (q["avg_us"] * 1e-6) / (HBM_PEAK_GBS * 1e9), 4),
            "source": "%s (rocprofv3 --kernel-trace, idle GPU, 16 x 8192 points, r = 2, 64 samples)" % rel(path)}
        if "SQ_INSTS_VALU" in q:
            peak = 256 * 4 * 2.4e9 / 2 / 1e9
            off["ball_query_valu_issue"] = {
                "kernel": qn, "wave_insts_valu": q["SQ_INSTS_VALU"], "kernel_us": q["avg_us"], "peak_ginst_s": peak,
                "frac_of_issue_peak": round(q["SQ_INSTS_VALU"] / (q["avg_us"] * 1e-6) / (peak * 1e9), 3),
                "lds_bank_conflict_share": (round(q["SQ_LDS_BANK_CONFLICT"] / q["SQ_LDS_IDX_ACTIVE"], 3)
                                            if q.get("SQ_LDS_IDX_ACTIVE") else None),
                "note": "peak = 256 CUs x 4 SIMDs x 2.4 GHz / 2 cycles per wave64 VALU on a SIMD (one wavefront alone issues every 4)",
                "source": rel(path)}
    except Exception as err:  # noqa: BLE001
        off["ball_query_traffic_bytes"] = {"bytes": None, "shape": None, "source": rel(path) if path else None,
                                           "error": "%s: %s" % (type(err).__name__, str(err)[:160])}
    path = _newest_profile("step_hbm_traffic")
    try:
        if path is None:
            raise FileNotFoundError("profiles/rNN_step_hbm_traffic.txt")
        row = [ln for ln in open(path) if ln.startswith("ALL KERNELS")][-1].split()
        off["step_traffic_mib"] = {
            "fetch_reported": float(row[-2]), "write": float(row[-1]),
            "source": "%s (per-kernel FETCH_SIZE / WRITE_SIZE table of one C4 step, mean of whole timed steps; the estimate doubles "
                      "the reported fetch: MI355X_MICROARCH.md)" % rel(path)}
    except Exception as err:  # noqa: BLE001
        off["step_traffic_mib"] = {"fetch_reported": None, "write": None, "source": rel(path) if path else None,
                                   "error": "%s: %s" % (type(err).__name__, str(err)[:160])}
    return off


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU (config batch_size: 4)")
    ap.add_argument("--npoint", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--timed-only", action="store_true",
                    help="print the line after the timed steps: no extras (graph replay, operator tables, copies), no CPU baseline — "
                         "what the rocprofv3 --pmc passes of tools/pmc_step.sh run")
    return ap.parse_args()


def cpu_baseline(npoint):
    """The same train step on the host with the CPU oracle's operators ("port"), 1 sample x 4 views."""
    import ogc_amd.pointnet2.pointnet2 as api
    from oracle import oracle as orc
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion, train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    orc.build()
    saved = api._native
    api._native = orc.Pointnet2CudaCPU()
    try:
        torch.manual_seed(10)
        net = MaskFormer3D(n_slot=10, n_point=npoint, transformer_embed_dim=128)
        crit = build_criterion(KITTI_LOSS)
        opt = torch.optim.Adam(net.parameters(), lr=1e-3)
        batch = make_scene_batch(1, npoint, 10, seed=1234, aug=True)
        times = []
        for _ in range(2):   # (two steps, the faster one reported: the first also pays the process's first touch of the CPU paths)
            t0 = time.time()
            train_step(net, crit, opt, batch, 1000, True)
            times.append(time.time() - t0)
        dt = min(times)
    finally:
        api._native = saved
    cores = min(torch.get_num_threads(), os.cpu_count() or 1)
    return {"value": round(4 / dt, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "sample": "2 train steps of 1 sample x 4 views (4 clouds) of %d pts, the faster one: torch-CPU layers + oracle operators "
                      "(OpenMP), %s s" % (npoint, " / ".join("%.1f" % t for t in times))}


def cpu_ops(pc_dev):
    """BASELINE.md 3 / SURVEY 8d: each operator of the path on the host (the oracle: single thread and OpenMP over all cores)
    and on the GPU, on the SAME tensors — one 8192-point cloud of the timed batch — median of 5 runs each.  B = 1 keeps the
    single-thread scans to seconds; at B = 1 the GPU operators are launch-latency-bound (tools/bench_ops.py has the batched
    table), so the ratio understates the batched GPU rate."""
    import statistics
    import numpy as np
    from oracle import oracle as orc
    from ogc_amd import pointnet2_cuda as nat
    pc1 = pc_dev[:1].contiguous()
    N = pc1.shape[1]
    host = pc1.cpu().numpy()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    teams = sorted({t for t in (8, 32, cores) if t <= cores})  # OpenMP team sizes tried (the operators are milliseconds long:
                                                               # waking 256 threads costs more than most of them take)

    def med(fn, reps=5):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)

    def gpu_med(fn, reps=5):
        ts = []
        fn()
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        return statistics.median(ts)

    fps_idx = orc.fps(host, 2048)
    centres = np.take_along_axis(host, fps_idx[:, :, None].astype(np.int64), 1)
    cen_dev = torch.from_numpy(centres).to(pc1.device)
    feats = np.random.default_rng(0).standard_normal((1, 64, 2048), dtype=np.float32)
    nn_d, nn_i = orc.three_nn(host, centres)
    w = (1.0 / (nn_d + 1e-8)); w = (w / w.sum(-1, keepdims=True)).astype(np.float32)
    kidx = orc.knn(32, host, host)[1]
    f_dev, i3_dev, w_dev = torch.from_numpy(feats).to(pc1.device), torch.from_numpy(nn_i).to(pc1.device), torch.from_numpy(w).to(pc1.device)
    k_dev = torch.from_numpy(kidx).to(pc1.device)
    m10 = np.random.default_rng(1).standard_normal((1, 10, N), dtype=np.float32)
    m_dev = torch.from_numpy(m10).to(pc1.device)
    d32 = torch.empty(1, N, 32, device=pc1.device); i32 = torch.empty(1, N, 32, dtype=torch.int32, device=pc1.device)
    i64 = torch.zeros(1, N, 64, dtype=torch.int32, device=pc1.device)
    d3 = torch.empty(1, N, 3, device=pc1.device); i3 = torch.empty(1, N, 3, dtype=torch.int32, device=pc1.device)
    tmp = torch.empty(1, N, device=pc1.device); fi = torch.empty(1, 2048, dtype=torch.int32, device=pc1.device)
    out_i = torch.empty(1, 64, N, device=pc1.device); out_g = torch.empty(1, 10, N, 32, device=pc1.device)

    def fps_gpu():
        tmp.fill_(1e10)
        nat.furthest_point_sampling_wrapper(1, N, 2048, pc1, tmp, fi)

    table = {
        "furthest_point_sampling (8192 -> 2048)": (lambda: orc.fps(host, 2048), fps_gpu),
        "knn (8192 <- 8192, k=32)": (lambda: orc.knn(32, host, host), lambda: nat.knn_wrapper(1, N, N, 32, pc1, pc1, d32, i32)),
        "ball_query (8192 x 8192, r=2, nsample=64)": (lambda: orc.ball_query(2.0, 64, host, host),
                                                       lambda: nat.ball_query_wrapper(1, N, N, 2.0, 64, pc1, pc1, i64)),
        "three_nn (8192 <- 2048)": (lambda: orc.three_nn(host, centres), lambda: nat.three_nn_wrapper(1, N, 2048, pc1, cen_dev, d3, i3)),
        "three_interpolate (C=64, 2048 -> 8192)": (lambda: orc.three_interpolate(feats, nn_i, w),
                                                   lambda: nat.three_interpolate_wrapper(1, 64, 2048, N, f_dev, i3_dev, w_dev, out_i)),
        "group_points (C=10, 8192 x 32)": (lambda: orc.group(m10, kidx), lambda: nat.group_points_wrapper(1, 10, N, N, 32, m_dev, k_dev, out_g)),
    }
    rows = {}
    prev = orc.get_threads()
    try:
        for name, (cpu_fn, gpu_fn) in table.items():
            orc.set_threads(1)
            one = med(cpu_fn)
            best = None
            for t in teams:
                orc.set_threads(t)
                cpu_fn()  # (the team is created on the first call)
                ms = med(cpu_fn)
                if best is None or ms < best[0]:
                    best = (ms, t)
            rows[name] = {"cpu_1_thread_ms": round(one, 3), "cpu_openmp_ms": round(best[0], 3), "openmp_threads": best[1],
                          "gpu_ms": round(gpu_med(gpu_fn), 4)}
    finally:
        orc.set_threads(prev)
    return {"ops": rows, "cores_available": cores, "openmp_teams_tried": teams, "reps": 5, "statistic": "median",
            "tensors": "one cloud (B = 1) of the timed batch; GPU times are host-synchronised wall times of one call (launch latency included)",
            "kind": "port (oracle/ogc_oracle.c, the CPU restatement of pointnet2/src/*.cu; the reference has no CPU path)"}


def _event_pair_floor(n=64):
    """What a pair of HIP events reads with NOTHING between them on the current stream (ms): the floor of every per-launch
    event measurement of LaunchTimer (two marker packets).  Minimum over `n` pairs on an otherwise idle GPU."""
    pairs = []
    torch.cuda.synchronize()
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        e1.record()
        pairs.append((e0, e1))
    torch.cuda.synchronize()
    return min(a.elapsed_time(b) for a, b in pairs)


def _time(fn, iters=20, warm=3):
    """Mean duration (ms) of `fn` over back-to-back launches on an idle GPU: events on torch's current stream, which is
    the stream the operators launch on (ogc_amd/pointnet2_cuda.py::_stream)."""
    for _ in range(warm):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters


def steps_with_h2d(model, crit, opt, host_batches, it, train_step, dev, steps=12, warm=3):
    """The timed loop again with the loader's half of the boundary inside: every batch is copied from pinned host memory into one
    of three device slots on a copy stream, two steps before the step that trains on it (the step in between prefetches its
    geometry).  A slot is overwritten only after the step that trained on it has finished (event on the launch stream)."""
    copy = torch.cuda.Stream(device=dev)
    slots = [tuple(torch.empty(t.shape, dtype=t.dtype, device=dev) for t in host_batches[0]) for _ in range(3)]
    done, ready = [None] * 3, [None] * 3

    def upload(j):
        s = j % 3
        if done[s] is not None:
            copy.wait_event(done[s])
        with torch.cuda.stream(copy):
            for d, h in zip(slots[s], host_batches[j % len(host_batches)]):
                d.copy_(h, non_blocking=True)
            ready[s] = torch.cuda.Event()
            ready[s].record(copy)

    main = torch.cuda.current_stream(dev)
    upload(0)
    upload(1)
    main.wait_event(ready[0])
    pre, t1 = None, None
    for j in range(warm + steps):
        if j == warm:
            torch.cuda.synchronize()
            t1 = time.perf_counter()
        upload(j + 2)
        main.wait_event(ready[(j + 1) % 3])
        pre = train_step(model, crit, opt, slots[j % 3], it, True, sync=False, prefetched=pre, next_batch=slots[(j + 1) % 3]).prefetched
        done[j % 3] = torch.cuda.Event()
        done[j % 3].record(main)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t1) / steps * 1e3
    nbytes = sum(t.numel() * t.element_size() for t in host_batches[0])
    return round(ms, 3), ("%d steps after the timed region; each batch (%.2f MB: clouds, flows, labels, valid flags) is copied from "
                          "pinned host memory on a copy stream two steps ahead of its use" % (steps, nbytes / 1e6))


def measure_extras(pc, a):
    """Live readings, taken after the timed region on the idle GPU, for the other kernels the north star names: k-NN
    against the HBM roofline, a dense SharedMLP GEMM against the fp32 MFMA peak, and the FlowStep3D correlation layer."""
    from ogc_amd import pointnet2_cuda as nat
    out = {}
    B, N, _ = pc.shape
    k, r = KITTI_K, KITTI_R
    d = torch.empty(B, N, k, device=pc.device)
    i = torch.empty(B, N, k, dtype=torch.int32, device=pc.device)
    alg = B * (12 * N + 12 * N + 8 * N * k)                       # 12n + 12m + 8nk per cloud
    ms_c = _time(lambda: nat.knn_clamped_wrapper(B, N, N, k, r, pc, pc, d, i))
    ms_p = _time(lambda: nat.knn_wrapper(B, N, N, k, pc, pc, d, i))
    out["roofline_knn"] = {
        "kernel": "ogc_knn_clamped (grid_build_split_kernel + knn_cells_kernel<%d> + knn_grid_kernel<1> for the rows it leaves): the smoothness "
                  "term's k-NN, k=%d clamped at %g m" % (k, k, r),
        "bound": "hbm", "achieved": round(alg / ms_c / 1e6, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(alg / ms_c / 1e6 / HBM_PEAK_GBS, 5), "avg_ms": round(ms_c, 4), "algorithmic_bytes": alg,
        "shape": {"B": B, "n": N, "m": N, "k": k, "radius": r},
        "unclamped_ogc_knn": {"avg_ms": round(ms_p, 4), "achieved": round(alg / ms_p / 1e6, 2),
                              "frac": round(alg / ms_p / 1e6 / HBM_PEAK_GBS, 5),
                              "kernel": "ogc_knn (grid_build_split_kernel + knn_wave_kernel<0>: a wavefront per query, threshold selection "
                                        "+ 64-lane bitonic network; knn_grid_kernel for rows it marks)"},
        "note": "idle GPU, 20 back-to-back launches; the radius-limited search stops once the scanned cells cover the clamp "
                "radius (neighbours beyond it are replaced by the nearest one in the output).  Both searches are bound by "
                "instruction issue, not by HBM (SURVEY 8d): the HBM fraction is reported because the north star asks for it"}
    # the set-abstraction groupers' searches (queries = the sampled centres, k = 64, clamped at the level's radius) by themselves:
    # in the step they ride the geometry side stream underneath the dense kernels, and kernel_ms reads their latency THERE
    iso = {}
    for (nq, nm, rad) in ((N // 4, N, 2.0), (N // 8, N // 4, 4.0), (N // 16, N // 8, 8.0)):
        q, m_ = pc[:, :nq].contiguous(), pc[:, :nm].contiguous()
        dq = torch.empty(B, nq, 64, device=pc.device)
        iq = torch.empty(B, nq, 64, dtype=torch.int32, device=pc.device)
        iso[str((B, nq, nm, 64))] = round(_time(lambda: nat.knn_clamped_wrapper(B, nq, nm, 64, rad, q, m_, dq, iq)), 4)
    out["kernel_ms_isolated"] = {"ogc_knn_clamped": iso,
                                 "note": "the same operator calls as kernel_ms.ogc_knn_clamped, 20 back to back on the idle GPU"}
    # dense per-group MLP GEMM on the fp32 matrix pipe: SA3's 128 -> 128 layer on 16 x 32768 positions
    Bc, cin, cout, hw = B, 128, 128, 32768
    x = torch.randn(Bc, cin, hw, device=pc.device)
    w = torch.randn(cout, cin, device=pc.device)
    y = torch.empty(Bc, cout, hw, device=pc.device)
    ms_g = _time(lambda: nat.conv1x1_gemm_wrapper(Bc, cout, cin, hw, 0, w, x, y))
    tf = 2.0 * Bc * hw * cin * cout / ms_g / 1e9
    out["mfma"] = {"kernel": "conv1x1_gemm_stream_kernel (v_mfma_f32_16x16x4_f32), SA3 layer 128 -> 128 on %d x %d positions" % (Bc, hw),
                   "bound": "mfma", "achieved": round(tf, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                   "frac": round(tf / FP32_MFMA_PEAK_TF, 4), "avg_ms": round(ms_g, 4),
                   "hbm_gbs": round(4.0 * Bc * hw * (cin + cout) / ms_g / 1e6, 1),
                   "note": "MFMA utilisation of the widest hand-written GEMM of the step = useful fp32 flops / fp32 MFMA peak "
                           "(this layer sits where the fp32 MFMA roof and the HBM roof meet; narrower layers are HBM-bound)"}
    # the chunked kernel (csrc/gemm_chunk.hip) on the widest product of the step: SA3's 256-channel input gradient
    wt = torch.randn(256, 128, device=pc.device)
    gy = torch.randn(Bc, 256, hw, device=pc.device)
    ms_c = _time(lambda: nat.conv1x1_gemm_any_wrapper(Bc, 128, 256, hw, 1, wt, gy, y))
    tfc = 2.0 * Bc * hw * 128 * 256 / ms_c / 1e9
    out["mfma"]["gemm_chunk_kernel"] = {"shape": "128 <- 256 channels (input gradient of SA3's last layer) on %d x %d positions" % (Bc, hw),
                                        "achieved": round(tfc, 2), "frac": round(tfc / FP32_MFMA_PEAK_TF, 4), "avg_ms": round(ms_c, 4)}
    del x, y, gy
    # FlowStep3D correlation layer at config C3's level-2 shape (BASELINE config 3): B = 1, 2048 points, k = 16
    from ogc_amd.utils.flowstep3d_util import FlowEmbedding
    torch.manual_seed(0)
    fe = FlowEmbedding(radius=1.5, nsample=16, in_channel=64, mlp=[128, 128, 128]).to(pc.device).eval()
    p1 = pc[:1, :2048].transpose(1, 2).contiguous()
    p2 = (pc[:1, :2048] + 0.05 * torch.randn(1, 2048, 3, device=pc.device)).transpose(1, 2).contiguous()
    f1, f2 = torch.randn(1, 64, 2048, device=pc.device), torch.randn(1, 64, 2048, device=pc.device)
    with torch.no_grad():
        ms_fe = _time(lambda: fe(p1, p2, f1, f2))
        dk = torch.empty(1, 2048, 16, device=pc.device)
        ik = torch.empty(1, 2048, 16, dtype=torch.int32, device=pc.device)
        q, kn = p1.transpose(1, 2).contiguous(), p2.transpose(1, 2).contiguous()
        ms_k = _time(lambda: nat.knn_clamped_wrapper(1, 2048, 2048, 16, 1.5, q, kn, dk, ik))
    flops = 2.0 * 2048 * 16 * (131 * 128 + 128 * 128 + 128 * 128)
    algk = 12 * 2048 + 12 * 2048 + 8 * 2048 * 16
    out["corr_layer"] = {"layer": "FlowEmbedding (utils/flowstep3d_util.py:27-66): kNN(16) + clamp 1.5 m -> group -> 131->128->128->128 "
                                  "(BatchNorm eval) -> max, B=1, 2048 points",
                         "avg_ms": round(ms_fe, 4),
                         "knn": {"avg_ms": round(ms_k, 4), "achieved_gbs": round(algk / ms_k / 1e6, 3), "peak_gbs": HBM_PEAK_GBS,
                                 "frac": round(algk / ms_k / 1e6 / HBM_PEAK_GBS, 6)},
                         "mfma": {"tflops_over_whole_layer": round(flops / ms_fe / 1e9, 3), "peak": FP32_MFMA_PEAK_TF,
                                  "frac": round(flops / ms_fe / 1e9 / FP32_MFMA_PEAK_TF, 5), "flops": flops},
                         "note": "BASELINE.md 4: k-NN GB/s and MFMA TF/s against peak; at B = 1 the layer is a chain of small "
                                 "launches (latency-bound), not a bandwidth- or MFMA-bound kernel"}
    return out


def config2_reading(dev, steps=12, warm=4):
    """BASELINE config 2 next to the headline: config/ogcdr_unsup_synthetic.yaml (segnet_ogcdr, 8 samples x 4 views x 4096 points,
    `matmul_precision: bf16` = bf16 operands + 16-bit activations inside the set-abstraction MLPs) through the same train_step loop,
    `steps` steps after `warm`, fresh weights; the operand precision is restored afterwards."""
    import yaml
    from ogc_amd import fused
    from ogc_amd.pointnet2 import pointnet2 as api
    from ogc_amd.train_seg import build_segnet
    from ogc_amd.train_step import build_criterion, make_optimizer, train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "config", "ogcdr_unsup_synthetic.yaml")) as f:
        cfg = yaml.safe_load(f)
    prev = api._native.get_matmul_precision()
    api._native.set_matmul_precision(cfg.get("matmul_precision", "fp32"))
    try:
        torch.manual_seed(cfg["random_seed"])
        net = build_segnet(cfg).to(dev)
        crit = build_criterion(cfg["loss"])
        opt = make_optimizer(net.parameters(), lr=cfg["lr"])
        batch = make_scene_batch(cfg["batch_size"], cfg["segnet"]["n_point"], cfg["segnet"]["n_slot"], seed=1, outdoor=False, aug=True,
                                 device=dev)
        torch.cuda.reset_peak_memory_stats()
        pre, pend = None, None
        for i in range(warm + steps):
            if i == warm:
                torch.cuda.synchronize()
                t0 = time.perf_counter()
            pend = train_step(net, crit, opt, batch, 10 ** 6, True, sync=False, prefetched=pre, next_batch=batch)
            pre = pend.prefetched
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        losses, stepped = pend.result()
        clouds = cfg["batch_size"] * batch[1].shape[1]
        return {"workload": "C2 OGC-DR train_seg unsup: segnet_ogcdr, %d samples x %d views x %d pts, matmul_precision %s" %
                            (cfg["batch_size"], batch[1].shape[1], cfg["segnet"]["n_point"], cfg.get("matmul_precision", "fp32")),
                "ms_per_step": round(ms, 3), "point_clouds_per_s": round(clouds / ms * 1e3, 1), "steps": steps, "warmup": warm,
                "dtype": "bf16 operands and bf16-stored activations (ogc_amd.fused.ACT16=%s), fp32 accumulation / statistics / parameters" % fused.ACT16,
                "optimizer_stepped": bool(stepped), "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                "act16_leave": int(fused.GATE_MISSES.get("act16_leave", 0)),
                "loss_sum": round(float(losses["sum"]), 5)}
    finally:
        api._native.set_matmul_precision(prev)


def oa_icp_reading(dev, iters=20, reps=5):
    """Object-aware ICP at its own shape (reference oa_icp.py:41-84, :175: the KITTI-SF refinement round — B = 4 scenes of 8192
    points, K = 10 slots, 20 iterations): ms per call of ogc_amd.oa_icp.object_aware_icp, and the soft-NN step's pair rate
    (B N^2 (query, candidate) pairs per iteration) against the fp32 MFMA peak its two dot products run on (5 + K padded to 8 + 12
    reduction terms -> 2 * 20 flop per pair)."""
    from ogc_amd.oa_icp import object_aware_icp
    from ogc_amd.utils.synthetic import make_scene_batch
    B, N, K = 4, 8192, 10
    pcs, segms, flows, _ = make_scene_batch(B, N, K, seed=5, aug=False, device=dev)
    pc1, pc2, flow = pcs[:, 0].contiguous(), pcs[:, 1].contiguous(), flows[:, 0].contiguous()
    eye = torch.eye(K, device=dev)
    g = torch.Generator(device=dev).manual_seed(7)
    mask1 = (4 * eye[segms[:, 0].long().to(dev) % K] + torch.randn(B, N, K, device=dev, generator=g)).softmax(-1)
    mask2 = (4 * eye[segms[:, 1].long().to(dev) % K] + torch.randn(B, N, K, device=dev, generator=g)).softmax(-1)
    noisy = flow + 0.05 * torch.randn(B, N, 3, device=dev, generator=g)
    with torch.no_grad():
        object_aware_icp(pc1, pc2, noisy, mask1, mask2, icp_iter=iters, temperature=0.01)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            out = object_aware_icp(pc1, pc2, noisy, mask1, mask2, icp_iter=iters, temperature=0.01)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / reps * 1e3
        # the soft-NN step alone (ogc_soft_nn_target, the pack of the candidates included)
        from ogc_amd import pointnet2_cuda as nat
        target = torch.empty(B, N, 3, device=dev)
        q = (pc1 + noisy).contiguous()
        m1c, m2c = mask1.contiguous(), mask2.contiguous()
        step_ms = _time(lambda: nat.soft_nn_target_wrapper(B, N, N, K, 0.01, q, pc2, m1c, m2c, target))
    pairs = B * N * N
    tflops = pairs * 2 * 20 / (step_ms * 1e-3) / 1e12
    return {"workload": "object_aware_icp, B = %d, N = %d, K = %d, %d iterations (oa_icp.py:175 round 1)" % (B, N, K, iters),
            "ms_per_call": round(ms, 3), "ms_per_iteration": round(ms / iters, 4),
            "soft_nn_step": {"kernel": "ogc_soft_nn_target (soft_nn_pack_kernel + soft_nn_mfma_kernel<3>)", "avg_ms": round(step_ms, 4),
                             "gpairs_per_s": round(pairs / (step_ms * 1e-3) / 1e9, 1),
                             "bound": "mfma", "achieved": round(tflops, 2), "peak": FP32_MFMA_PEAK_TF, "unit": "TFLOP/s",
                             "frac": round(tflops / FP32_MFMA_PEAK_TF, 4),
                             "note": "2 x 20 flop per pair on v_mfma_f32_16x16x4_f32 (8 of them carry zeros: 5 distance + 10 mask "
                                     "terms padded to 8 + 12); the vector unit does max / sqrt / scale / 2 exp2 / 5 fma per pair"},
            "epe_vs_truth": round(float((out - flow).norm(dim=-1).mean()), 4), "epe_of_input": round(float((noisy - flow).norm(dim=-1).mean()), 4)}


def config3_flow_train_reading(dev, steps=8, warm=3):
    """BASELINE config 3 as train_flow.py runs it (reference train_flow.py:33-86): FlowStep3D (flownet_kitti) on 8192-point pairs,
    batch 4, 4 refinement iterations, UnsupervisedFlowStep3DLoss (Chamfer + smoothness on every iteration), backward, NaN rule,
    Adam — ogc_amd.train_step.flow_train_step."""
    from ogc_amd.losses.flow_loss_unsup import ChamferLoss, SmoothLoss, UnsupervisedFlowStep3DLoss
    from ogc_amd.models.flownet_kitti import FlowStep3D
    from ogc_amd.train_step import flow_train_step, make_optimizer
    from ogc_amd.utils.synthetic import make_scene_batch
    B, N, iters = 4, 8192, 4
    # (this step's launch thread needs ~32 of its ~37 ms: what the earlier readings left behind — cached blocks, collectable
    # cycles — is cleared first, and the collector does not run inside the timed steps)
    import gc
    gc.collect()
    torch.cuda.empty_cache()
    torch.manual_seed(10)
    net = FlowStep3D(npoint=N, loc_flow_nn=16, loc_flow_rad=1.5).to(dev)
    crit = UnsupervisedFlowStep3DLoss(ChamferLoss(2), SmoothLoss(3., 1., {'k': 4, 'radius': 0.5, 'loss_norm': 1},
                                                                  {'k': 8, 'radius': 1.0, 'loss_norm': 1}),
                                      weights=[0.75, 0.25], iters_w=[0.8, 0.2, 0.4, 0.6])
    opt = make_optimizer(net.parameters(), lr=1e-3)
    pcs, _, flows, _ = make_scene_batch(B, N, 10, seed=1, aug=False, device=dev)
    # two distinct resident batches, alternating: the batch a step prefetches the sampling chains of really is the next one
    batches = [(pcs, None, flows, None)]
    pcs_b, _, flows_b, _ = make_scene_batch(B, N, 10, seed=2, aug=False, device=dev)
    batches.append((pcs_b, None, flows_b, None))
    torch.cuda.reset_peak_memory_stats()
    pend, ahead = None, None
    collecting = gc.isenabled()
    try:
        for i in range(warm + steps):
            if i == warm:
                torch.cuda.synchronize()
                gc.disable()
                t0 = time.perf_counter()
            pend = flow_train_step(net, crit, opt, batches[i % 2], iters, sync=False, prefetched=ahead, next_batch=batches[(i + 1) % 2])
            ahead = pend.prefetched
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
    finally:
        if collecting:
            gc.enable()
    losses, stepped = pend.result()
    return {"workload": "C3 KITTI-SF train_flow unsup: flownet_kitti (FlowStep3D), %d pairs x %d pts, iters = %d, fwd + loss + bwd + Adam"
                        % (B, N, iters),
            "ms_per_step": round(ms, 3), "point_cloud_pairs_per_s": round(B / ms * 1e3, 1), "steps": steps, "warmup": warm,
            "dtype": "f32", "optimizer_stepped": bool(stepped), "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
            "loss_sum": round(float(losses["sum"]), 5)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        from ogc_amd.utils.dist_util import pin_rank_to_cores
        pin_rank_to_cores()  # before the GPU runtime starts its helper threads: they inherit the block
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    # OGC_BENCH_SHARE_GPU=1 (tests only): every rank on cuda:0 — the multi-rank control flow of this file (sharding, barrier + MAX
    # timing, one line from rank 0, nobody left alone in a collective) exercised on a one-GPU box, over OGC_BENCH_BACKEND=gloo
    # (RCCL refuses two ranks on one device)
    if os.environ.get("OGC_BENCH_SHARE_GPU") == "1":
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:  # a 1-process torchrun launch still exercises RCCL init + the DDP wrapper
        # no device_id: binding the group to the device at init (eager communicator) slows EVERY later launch of this
        # process on this stack — the same un-wrapped step takes 17.5 ms instead of 14.9 (tools/ddp_cost.py)
        dist.init_process_group(os.environ.get("OGC_BENCH_BACKEND", "nccl"))
    assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)

    import ogc_amd  # noqa: F401  (fails loudly if libogc_ops.so is missing)
    from ogc_amd import pointnet2_cuda as nat
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
    from ogc_amd.utils.synthetic import make_scene_batch
    global KITTI_K, KITTI_R
    KITTI_K = KITTI_LOSS["smooth_loss_params"]["knn_loss_params"]["k"]
    KITTI_R = KITTI_LOSS["smooth_loss_params"]["knn_loss_params"]["radius"]

    torch.manual_seed(10)  # random_seed: 10 in the reference YAMLs; identical init on every rank
    net = MaskFormer3D(n_slot=10, n_point=a.npoint, use_xyz=True, n_transformer_layer=2,
                       transformer_embed_dim=128, transformer_input_pos_enc=False).to(dev)
    model = net
    if dist.is_initialized():
        # one flat gradient all-reduce per step instead of DistributedDataParallel's per-parameter hooks (its host
        # overhead would make the launch thread the bottleneck): ogc_amd/utils/dist_util.py
        from ogc_amd.utils.dist_util import FlatDataParallel, always_reduce
        always_reduce(True)  # a one-process launch still goes through RCCL
        model = FlatDataParallel(net)
    crit = build_criterion(KITTI_LOSS)
    opt = make_optimizer(net.parameters(), lr=1e-3, weight_decay=0.0)
    # NB distinct synthetic batches, generated on the host (pinned) and uploaded BEFORE the timed region: the timed steps rotate
    # through them, so the batch a step prefetches geometry for really is the next one (different clouds, different FPS tie
    # records), and `value` is measured with the inputs resident in HBM.  ms_per_step_with_h2d (an extra, after the timed region)
    # repeats the steps with every batch copied from pinned host memory on a copy stream two steps ahead.
    NB = 4
    host_batches = [tuple(t.pin_memory() for t in make_scene_batch(a.batch, a.npoint, 10, seed=1234 + rank + 101 * j, outdoor=True,
                                                                     aug=True))
                    for j in range(NB)]
    batches = [tuple(t.to(dev) for t in hb) for hb in host_batches]
    batch = batches[0]
    clouds_per_step = a.batch * 4

    def sync():
        if dist.is_initialized():
            dist.barrier(device_ids=[local])
        torch.cuda.synchronize()

    it = 1000  # it*b >= every start_step -> dynamic + smooth + invariance all active
    # The trainer holds the next batch while a step runs (ogc_amd/train_seg.py reads one batch ahead), so each step
    # also queues the network's coordinate-only work (FPS / kNN / 3-NN) of the FOLLOWING step on a side stream and consumes
    # the plan made during the previous step: one plan is computed per step, inside the timed region, none is reused.
    pre = None
    step_no = 0
    for _ in range(a.warmup):
        pre = train_step(model, crit, opt, batches[step_no % NB], it, True, sync=False, prefetched=pre,
                         next_batch=batches[(step_no + 1) % NB]).prefetched
        step_no += 1
    sync()
    if hasattr(model, "time_collectives"):
        model.time_collectives(True)
    timed_ops = {"ogc_ball_query", "ogc_ball_query_cells", "ogc_cell_grid_build", "ogc_knn_clamped_cells", "ogc_knn_clamped",
                 "ogc_furthest_point_sampling", "ogc_furthest_point_sampling_chain"}
    # (the event pairs of these ~12 launches per step cost the step < 0.05 ms: measured with the roofline's three alone)
    with nat.LaunchTimer(timed_ops) as timer:
        t0 = time.perf_counter()
        mark = bool(os.environ.get("OGC_BENCH_MARK"))  # profiling aid: a marker kernel per step (tools/prof_summary.py)
        for _ in range(a.steps):
            if mark:
                torch.cuda._sleep(1000)
            # sync=False: the step's scalars (losses, NaN flag) travel to the host asynchronously and are read after
            # the timed region; every step still computes and copies them
            pending = train_step(model, crit, opt, batches[step_no % NB], it, True, sync=False, prefetched=pre,
                                 next_batch=batches[(step_no + 1) % NB])
            pre = pending.prefetched
            step_no += 1
        issued = time.perf_counter() - t0   # the launch thread is done queueing the K steps here; the GPU is still running them
        sync()
        elapsed = time.perf_counter() - t0
    loss_dict, stepped = pending.result()
    # train_step follows the reference in swallowing a RuntimeError of backward() (train_seg.py:75-78): a bench line for steps
    # that did not run their backward pass and optimizer would be worthless, so it must not be printed
    assert stepped and all(v == v for v in loss_dict.values()), "the timed steps did not step the optimizer: %r" % (loss_dict,)
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    isolated_ms, extras, pair_floor_ms = None, {}, 0.0
    if a.timed_only:
        a.no_cpu_baseline = True
    # The extras are single-GPU readings taken by rank 0 alone after the timed region; some of them run further training steps,
    # i.e. gradient collectives the other ranks would not join: with more than one rank the line carries the timed region only.
    if rank == 0 and not a.timed_only and world == 1:
        def graph_reading():
            # the same step replayed as ONE HIP graph (ogc_amd/graph_step.py; `--hip-graph` of the training driver): what a step
            # costs when the launch thread is out of the picture.  A second network / optimizer (the capture needs step counts
            # on the device); reported next to the eager headline, never instead of it.
            try:
                from ogc_amd.graph_step import GraphedTrainStep
                torch.manual_seed(10)
                net_g = MaskFormer3D(n_slot=10, n_point=a.npoint, use_xyz=True, n_transformer_layer=2,
                                     transformer_embed_dim=128, transformer_input_pos_enc=False).to(dev)
                gs = GraphedTrainStep(net_g, build_criterion(KITTI_LOSS),
                                      make_optimizer(net_g.parameters(), lr=1e-3, weight_decay=0.0, capturable=True), batch, it, True)
                for _ in range(25):
                    gs.step(batch)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for _ in range(20):
                    pg = gs.step(batch)
                torch.cuda.synchronize()
                extras["ms_per_step_hip_graph"] = round((time.perf_counter() - t1) / 20 * 1e3, 3)
                ld, ok = pg.result()
                extras["hip_graph_note"] = ("the step as one replayed HIP graph (train_seg --hip-graph), 20 replays right after the timed "
                                            "region; stepped=%s, loss sum %.4f" % (bool(ok), ld.get("sum", float("nan"))))
                del gs, net_g
            except Exception as err:  # an extra reading must not cost the headline its line
                extras["ms_per_step_hip_graph"] = None
                extras["hip_graph_note"] = "not measured: %s" % (str(err)[:200],)
        # The graph reading comes BEFORE the other extras (OGC_BENCH_GRAPH_FIRST=0: after them, as until round 4).  The host-to-device
        # reading below creates a copy stream, the fifth stream of the process; ROCm maps streams onto four hardware queues by
        # default, the copy stream's queue is then shared with one of the replay's streams and the replay measured 13.2-13.3 ms
        # instead of 11.2-11.4 (bisected with OGC_BENCH_SKIP; GPU_MAX_HW_QUEUES=8 / 16 make both readings WORSE: 16.6 / 18.6 ms, 3
        # costs the eager step 0.3 ms, 2 dead-locks the step's side streams).  tools/graph_step.py measures the same pair alone.
        graph_first = os.environ.get("OGC_BENCH_GRAPH_FIRST", "1") != "0"
        graph_extras = {}
        if graph_first and world == 1 and not dist.is_initialized():
            graph_reading()
            graph_extras, extras = extras, {}
        # the same ball-query call on an otherwise idle GPU (in the step it shares the chip with the dense kernels)
        from ogc_amd.pointnet2.pointnet2 import ball_query
        pc = torch.cat([batch[0][:, v] for v in range(4)]).contiguous()
        bl = KITTI_LOSS["smooth_loss_params"]["ball_q_loss_params"]
        isolated_ms = _time(lambda: ball_query(bl["radius"], bl["k"], pc, pc))
        # ... and at four times the clouds per launch (the views of all four resident batches): what the operator reaches once its
        # fixed costs — two dependent launches of ~5 us each before the first candidate is tested — are spread over more work
        pc64 = torch.cat([torch.cat([bt[0][:, v] for v in range(4)]) for bt in batches]).contiguous()
        ms64 = _time(lambda: ball_query(bl["radius"], bl["k"], pc64, pc64))
        alg64 = pc64.shape[0] * (24 * pc64.shape[1] + 4 * pc64.shape[1] * bl["k"])
        at64 = {
            "kernel": "ogc_ball_query (own grid build + query), %d clouds of %d points per launch, idle GPU, 20 back to back" % (pc64.shape[0], pc64.shape[1]),
            "avg_ms": round(ms64, 4), "algorithmic_bytes": alg64, "achieved": round(alg64 / (ms64 * 1e-3) / 1e9, 2),
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(alg64 / (ms64 * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
            "note": "the same operator as roofline.isolated at 4 x the batch: per cloud the launch pair costs a quarter"}
        del pc64
        pair_floor_ms = _event_pair_floor()
        skip = set(filter(None, os.environ.get("OGC_BENCH_SKIP", "").split(",")))  # (development: leave extras out)
        extras = measure_extras(pc, a) if "ops" not in skip else {}
        extras["roofline_at_64_clouds"] = at64
        # The other configurations BEFORE the host-to-device reading: that one leaves a copy stream and pinned buffers behind, after
        # which every launch of the process costs the launch thread a little more — invisible at C4 (GPU-bound), 2 ms on the
        # FlowStep3D step, whose launch thread needs ~32 of its ~37 ms (36.7-36.9 before it, 38.6-38.9 after it, round 6).
        if "c2" not in skip:
            try:
                extras["config2_ogcdr_bf16"] = config2_reading(dev)
            except Exception as err:  # an extra reading must not cost the headline its line
                extras["config2_ogcdr_bf16"] = {"error": str(err)[:200]}
        if "c3" not in skip:
            try:
                extras["config3_flow_train"] = config3_flow_train_reading(dev)
            except Exception as err:
                extras["config3_flow_train"] = {"error": str(err)[:200]}
        if "icp" not in skip:
            try:
                extras["oa_icp"] = oa_icp_reading(dev)
            except Exception as err:
                extras["oa_icp"] = {"error": str(err)[:200]}
        if "h2d" not in skip:
            extras["ms_per_step_with_h2d"], extras["h2d_note"] = steps_with_h2d(model, crit, opt, host_batches, it, train_step, dev)
        # steps with the FPS chain shortcut off: every encoder level runs all its sampling rounds, as it must for clouds
        # with duplicated points (synthetic uniform clouds are tie-free, so levels 2-3 cost ~10 us in the headline)
        import ogc_amd.utils.pointnet2_util as sa_util
        sa_util.FPS_CHAIN_SHORTCUT = False
        try:
            pre2 = None
            for j in range((2 + 8) if "fps" not in skip else 0):
                if j == 2:
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                pre2 = train_step(model, crit, opt, batches[j % NB], it, True, sync=False, prefetched=pre2,
                                  next_batch=batches[(j + 1) % NB]).prefetched
            torch.cuda.synchronize()
            if "fps" not in skip:
                extras["ms_per_step_all_fps_rounds"] = round((time.perf_counter() - t1) / 8 * 1e3, 3)
        finally:
            sa_util.FPS_CHAIN_SHORTCUT = True
        if not graph_first and world == 1 and not dist.is_initialized():
            graph_reading()
        extras.update(graph_extras)
    if rank == 0:
        offline = load_offline()
        durs = timer.durations_ms()
        # In the step the ball query runs on the cell grid it shares with the loss's k-NN (ogc_cell_grid_build once per step for
        # both searches, ogc_ball_query_cells / ogc_knn_clamped_cells on it): the operator's time is its query launch plus its
        # HALF of the build.  `query_only` and the stand-alone operator (its own build + query, idle GPU) are reported next to it.
        bq = durs.get("ogc_ball_query_cells", [])
        shared = bool(bq)
        if not shared:
            bq = durs.get("ogc_ball_query", [])
        roof = None
        if bq:
            ms_q = sum(d for d, _ in bq) / len(bq)
            builds = durs.get("ogc_cell_grid_build", [])
            ms_b = sum(d for d, _ in builds) / len(builds) if builds else 0.0
            ms = ms_q + 0.5 * ms_b if shared else ms_q
            dims = bq[0][1]
            b_, n_, ns_ = (dims[0], dims[1], dims[3]) if shared else (dims[0], dims[1], dims[4])
            m_ = n_
            alg = b_ * (12 * m_ + 12 * n_ + 4 * m_ * ns_)            # 12M + 12N + 4M*nsample per cloud
            gbs = alg / (ms * 1e-3) / 1e9
            roof = {"kernel": ("ogc_ball_query_cells (ball_query_cells_kernel<64, 1>) on the cell grid shared with the loss's k-NN, plus "
                               "half of ogc_cell_grid_build (grid_build_split_kernel<8>)") if shared else
                              "ogc_ball_query (grid_build_split_kernel<8> + ball_query_cells_kernel<64, 1>)",
                    "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5),
                    "traffic": (offline["ball_query_traffic_bytes"]["bytes"]
                                if [b_, n_, m_, ns_] == offline["ball_query_traffic_bytes"]["shape"] else None),
                    "traffic_source": offline["ball_query_traffic_bytes"].get("source"),
                    "launches": len(bq), "avg_ms": round(ms, 4), "algorithmic_bytes": alg,
                    "query_only": {"avg_ms": round(ms_q, 4), "achieved": round(alg / (ms_q * 1e-3) / 1e9, 2),
                                   "frac": round(alg / (ms_q * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)},
                    "grid_build": {"avg_ms": round(ms_b, 4), "launches_per_step": len(builds) / max(a.steps, 1),
                                   "shared_by": "ogc_knn_clamped_cells (k-NN of the smoothness term) and ogc_ball_query_cells"},
                    # per-launch event pairs read ~4.6 us with nothing between them (two marker packets): the same launches
                    # with that floor removed, which is what the kernel trace of this command shows (profiles/)
                    "event_pair_floor_ms": round(pair_floor_ms, 5),
                    "floor_removed": (lambda q, bld: {"avg_ms": round(q + 0.5 * bld, 4), "query_ms": round(q, 4), "build_ms": round(bld, 4),
                                                      "achieved": round(alg / ((q + 0.5 * bld) * 1e-3) / 1e9, 2),
                                                      "frac": round(alg / ((q + 0.5 * bld) * 1e-3) / 1e9 / HBM_PEAK_GBS, 5)})(
                        max(ms_q - pair_floor_ms, 1e-6), max(ms_b - pair_floor_ms, 0.0) if shared else 0.0),
                    "isolated": None if isolated_ms is None else {
                        "avg_ms": round(isolated_ms, 4), "achieved": round(alg / (isolated_ms * 1e-3) / 1e9, 2),
                        "frac": round(alg / (isolated_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                        "note": "the stand-alone operator ogc_ball_query (its own grid build + query), 20 back-to-back "
                                "launches on an idle GPU after the timed region"},
                    "shape": {"B": b_, "N": n_, "M": m_, "nsample": ns_},
                    "note": "algorithmic bytes = B*(12M + 12N + 4M*nsample) (SURVEY 8d) / mean duration (HIP events on the launch "
                            "stream, inside the timed steps) of the query launch + half of the grid build it shares with the k-NN "
                            "search; the exact cell-list search tests ~N/60 candidates per centre, so the all-pairs figure of "
                            "8*B*N*M flop no longer describes the work done.  A pair of HIP events reads event_pair_floor_ms with "
                            "NOTHING between them (measured live, idle GPU): achieved / frac / avg_ms are the raw readings, "
                            "floor_removed the same launches with that floor taken off, which is what the rocprofv3 kernel "
                            "statistics of this command show (profiles/rNN_bench_kernel_stats.csv of the newest round; tools/bench_ops.py has the "
                            "idle-GPU table)",
                    "all_pairs_equivalent_tpairs_per_s": round(b_ * n_ * m_ / (ms * 1e-3) / 1e12, 2)}
        others = {}
        for name in ("ogc_knn_clamped", "ogc_knn_clamped_cells", "ogc_cell_grid_build", "ogc_furthest_point_sampling",
                     "ogc_furthest_point_sampling_chain"):
            if name in durs:
                per = {}
                for d, dims in durs[name]:
                    per.setdefault(str(dims[:4]), []).append(d)
                others[name] = {k: round(sum(v) / len(v), 4) for k, v in per.items()}
        out = {
            "metric": "point-clouds/sec (8192 pts) segnet fwd+bwd",
            "value": round(clouds_per_step * a.steps * world / elapsed, 3),
            "unit": "point-clouds/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(elapsed / a.steps * 1e3, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "C4 KITTI-SF train_seg unsup: segnet_kitti fwd + OGC loss (dynamic+smooth+invariance) "
                                   "+ bwd + Adam, %d samples x 4 views x %d pts per GPU" % (a.batch, a.npoint),
                       "clouds_per_step_per_gpu": clouds_per_step, "n_point": a.npoint, "n_slot": 10,
                       "parallelism": "dp%d" % world, "optimizer_stepped": bool(stepped),
                       "peak_hbm_gib": round(torch.cuda.max_memory_allocated() / 2 ** 30, 2),
                       "loss": {k: round(v, 5) for k, v in loss_dict.items()}},
            "roofline": roof,
            "kernel_ms": others,
            "launch_thread_ms_per_step": round(issued / a.steps * 1e3, 3),
        }
        out.update(extras)
        out["fps_note"] = ("levels 2-3 of the encoder cost ~10 us in ms_per_step because the synthetic uniform clouds are free "
                           "of exact fp32 distance ties (the chain shortcut, DESIGN.md); ms_per_step_all_fps_rounds is the same "
                           "step with the shortcut off, i.e. what clouds with duplicated points pay (FPS runs on a side stream)")
        out["offline"] = offline
        st = offline["step_traffic_mib"]
        if (a.batch, a.npoint) == (4, 8192) and st["fetch_reported"] is not None:
            ms_step = elapsed / a.steps * 1e3
            out["offline"]["step_traffic_est_gbs"] = round((2 * st["fetch_reported"] + st["write"]) * 2 ** 20 / (ms_step * 1e-3) / 1e9, 1)
        if dist.is_initialized():
            out["collective"] = {"backend": dist.get_backend(), "op": "all_reduce(SUM) of one flat fp32 gradient buffer per step",
                                 "payload_bytes": model.payload_bytes(), "world": world,
                                 "avg_ms": round(model.collective_ms() or 0.0, 4),
                                 "note": "events on the launch stream around the collective, rank 0, inside the timed steps"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.npoint)
            try:
                out["cpu_ops"] = cpu_ops(torch.cat([batch[0][:, v] for v in range(4)]))
            except Exception as err:  # an extra table must not cost the headline its line
                out["cpu_ops"] = {"error": str(err)[:200]}
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        sync()  # (nobody tears the group down while another rank may still be inside a collective)
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
