"""Headline benchmark: unsupervised OGC segmentation training throughput on 8192-point KITTI-SF-shaped scenes.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        bench.py --gpus N --steps K --warmup W

One "step" = `Trainer._train_it` of the reference (train_seg.py:47-86) on config C4 (SURVEY.md §8,
config/seg/kittisf/kittisf_unsup.yaml): per GPU 4 samples x 4 views (2 frames + 2 augmented) of 8192 points,
MaskFormer3D(segnet_kitti) forward on 16 clouds, UnsupervisedOGCLoss with all three terms active, backward,
NaN-gradient check, Adam step.  Inputs are synthetic, seeded and already resident in HBM.  Weak scaling: every
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
# HBM traffic of one ogc_ball_query call measured offline with rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate
# passes, tools/pmc_op.sh -> profiles/r01_ball_query_pmc_v2.txt): key = (B, N, M, nsample) -> bytes summed over the two
# kernels of the operator (grid_build_kernel + ball_query_grid_kernel; KiB as reported).  FETCH_SIZE is taken as
# reported (the gfx950 x2 correction of MI355X_MICROARCH.md applies to wide coalesced streams; these kernels issue
# 12-16 byte gathers); WRITE_SIZE of the query kernel equals the output size exactly.
PMC_TRAFFIC_BYTES = {(16, 8192, 8192, 64): int((8653.8 + 855.6 + 32768 + 2199.5) * 1024)}
FP32_VALU_PEAK_TF = 157.3  # fp32 vector peak
# all kernels of one C4 step (16 clouds x 8192 points), MiB as reported by the counters (profiles/r01_step_hbm_traffic_v20.txt)
STEP_FETCH_MIB, STEP_WRITE_MIB = 14922.5, 12373.3


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=4, help="samples per GPU (config batch_size: 4)")
    ap.add_argument("--npoint", type=int, default=8192)
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
        t0 = time.time()
        train_step(net, crit, opt, batch, 1000, True)
        dt = time.time() - t0
    finally:
        api._native = saved
    cores = min(torch.get_num_threads(), os.cpu_count() or 1)
    return {"value": round(4 / dt, 4), "unit": "point-clouds/s", "cores": cores, "kind": "port",
            "sample": "1 train step, 1 sample x 4 views (4 clouds) of %d pts: torch-CPU layers + oracle operators "
                      "(OpenMP), %.1f s" % (npoint, dt)}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert torch.cuda.is_available(), "bench.py needs an MI355X (no CPU fallback)"
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    under_launcher = "RANK" in os.environ and "MASTER_ADDR" in os.environ
    if world > 1 or under_launcher:  # a 1-process torchrun launch still exercises RCCL init + the DDP wrapper
        # no device_id: binding the group to the device at init (eager communicator) slows EVERY later launch of this
        # process on this stack — the same un-wrapped step takes 17.5 ms instead of 14.9 (tools/ddp_cost.py)
        dist.init_process_group("nccl")
    assert world == a.gpus, "--gpus %d but WORLD_SIZE=%d" % (a.gpus, world)

    import ogc_amd  # noqa: F401  (fails loudly if libogc_ops.so is missing)
    from ogc_amd import pointnet2_cuda as nat
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    from ogc_amd.train_step import KITTI_LOSS, build_criterion, make_optimizer, train_step
    from ogc_amd.utils.synthetic import make_scene_batch

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
    batch = make_scene_batch(a.batch, a.npoint, 10, seed=1234 + rank, outdoor=True, aug=True, device=dev)
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
    for _ in range(a.warmup):
        pre = train_step(model, crit, opt, batch, it, True, sync=False, prefetched=pre, next_batch=batch).prefetched
    sync()
    if hasattr(model, "time_collectives"):
        model.time_collectives(True)
    with nat.LaunchTimer({"ogc_ball_query", "ogc_knn_clamped", "ogc_furthest_point_sampling",
                          "ogc_furthest_point_sampling_chain"}) as timer:
        t0 = time.perf_counter()
        mark = bool(os.environ.get("OGC_BENCH_MARK"))  # profiling aid: a marker kernel per step (tools/prof_summary.py)
        for _ in range(a.steps):
            if mark:
                torch.cuda._sleep(1000)
            # sync=False: the step's scalars (losses, NaN flag) travel to the host asynchronously and are read after
            # the timed region; every step still computes and copies them
            pending = train_step(model, crit, opt, batch, it, True, sync=False, prefetched=pre, next_batch=batch)
            pre = pending.prefetched
        sync()
        elapsed = time.perf_counter() - t0
    loss_dict, stepped = pending.result()
    t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist.is_initialized():
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())

    isolated_ms = None
    if rank == 0:
        # the same ball-query call on an otherwise idle GPU (in the step it shares the chip with the dense kernels)
        from ogc_amd.pointnet2.pointnet2 import ball_query
        pc = torch.cat([batch[0][:, v] for v in range(4)]).contiguous()
        bl = KITTI_LOSS["smooth_loss_params"]["ball_q_loss_params"]
        for _ in range(3):
            ball_query(bl["radius"], bl["k"], pc, pc)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(20):
            ball_query(bl["radius"], bl["k"], pc, pc)
        e1.record()
        torch.cuda.synchronize()
        isolated_ms = e0.elapsed_time(e1) / 20
    if rank == 0:
        durs = timer.durations_ms()
        bq = durs.get("ogc_ball_query", [])
        roof = None
        if bq:
            ms = sum(d for d, _ in bq) / len(bq)
            b_, n_, m_, _r, ns_ = bq[0][1][:5]
            alg = b_ * (12 * m_ + 12 * n_ + 4 * m_ * ns_)            # 12M + 12N + 4M*nsample per cloud
            gbs = alg / (ms * 1e-3) / 1e9
            roof = {"kernel": "ogc_ball_query (grid_build_kernel + ball_query_grid_kernel)",
                    "bound": "hbm", "achieved": round(gbs, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(gbs / HBM_PEAK_GBS, 5),
                    "traffic": PMC_TRAFFIC_BYTES.get((b_, n_, m_, ns_)),
                    "launches": len(bq), "avg_ms": round(ms, 4), "algorithmic_bytes": alg,
                    "isolated": {"avg_ms": round(isolated_ms, 4), "achieved": round(alg / (isolated_ms * 1e-3) / 1e9, 2),
                                 "frac": round(alg / (isolated_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 5),
                                 "note": "same call, 20 back-to-back launches on an idle GPU after the timed region"},
                    "shape": {"B": b_, "N": n_, "M": m_, "nsample": ns_},
                    "note": "algorithmic bytes = B*(12M + 12N + 4M*nsample) (SURVEY 8d) / mean duration of the whole "
                            "operator call (HIP events on the launch stream, inside the timed steps); the exact "
                            "cell-list search tests ~N/60 candidates per centre, so the all-pairs figure of 8*B*N*M "
                            "flop no longer describes the work done; in the step the next batch's network geometry plan "
                            "(FPS / kNN on a side stream) shares the chip with it (tools/bench_ops.py has the idle-GPU table)",
                    "all_pairs_equivalent_tpairs_per_s": round(b_ * n_ * m_ / (ms * 1e-3) / 1e12, 2),
                    # offline (rocprofv3 --pmc SQ_INSTS_VALU + --kernel-trace, profiles/r01_ball_query_pmc_v2.txt):
                    # what actually bounds the search kernel is VALU issue, not HBM
                    "valu_issue_offline": {"kernel": "ball_query_grid_kernel", "wave_insts": 12.84e6, "kernel_us": 29.4,
                                           "achieved_ginst_s": round(12.84e6 / 29.4e-6 / 1e9, 1),
                                           "peak_ginst_s": round(256 * 4 * 2.4 / 4 * 1e3 / 1e3, 1),
                                           "frac": round(12.84e6 / 29.4e-6 / (256 * 4 * 2.4e9 / 4), 3),
                                           "shape": "B=16, N=M=8192, nsample=64"}}
        others = {}
        for name in ("ogc_knn_clamped", "ogc_furthest_point_sampling", "ogc_furthest_point_sampling_chain"):
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
        }
        if (a.batch, a.npoint) == (4, 8192):
            # HBM-side bytes of one whole step, measured offline (tools/pmc_step.sh, separate FETCH_SIZE / WRITE_SIZE
            # passes -> profiles/r01_step_hbm_traffic_v20.txt); the average rate uses THIS run's per-rank step time
            ms_step = elapsed / a.steps * 1e3
            out["step_traffic"] = {
                "fetch_mib_reported": STEP_FETCH_MIB, "write_mib": STEP_WRITE_MIB,
                "est_gbs": round((2 * STEP_FETCH_MIB + STEP_WRITE_MIB) * 2 ** 20 / (ms_step * 1e-3) / 1e9, 1),
                "note": "PMC counters summed over all kernels of a step; est_gbs doubles the reported fetch volume "
                        "(gfx950 tallies wide coalesced reads at half their bytes, MI355X_MICROARCH.md) and divides "
                        "by this run's step time: the step as a whole against the %.0f GB/s HBM peak" % HBM_PEAK_GBS}
        if dist.is_initialized():
            out["collective"] = {"backend": dist.get_backend(), "op": "all_reduce(SUM) of one flat fp32 gradient buffer per step",
                                 "payload_bytes": model.payload_bytes(), "world": world,
                                 "avg_ms": round(model.collective_ms() or 0.0, 4),
                                 "note": "events on the launch stream around the collective, rank 0, inside the timed steps"}
        if world == 1 and not a.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(a.npoint)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
