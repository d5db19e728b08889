"""One round of scene-flow refinement by object-aware ICP (counterpart of the reference's
oa_icp.py:87-236 written against this repo's layers).

    python -m ogc_amd.oa_icp_round config.yaml --round R [--save] [--flow-root DIR] [--synthetic N_SCENES]

Kept from the reference so that rounds chain with `train_seg`:
  * the segmentation YAML (config/seg/*/*.yaml) and the checkpoint `<save_path>_R<round>/best.pth.tar`
    (`{'model_state': ...}`, oa_icp.py:137-139);
  * frame pairs `view_sels = [[0, 1], [1, 0]]` of every scene (oa_icp.py:160), flows predicted for both directions;
  * input flows from `<flow-root>/flow_preds/<predflow_path>[_R<round-1>]/<scene id>/flow{1,2}.npy`
    (oa_icp.py:143-146, datasets/dataset_kittisf.py:125-137), refined flows written to
    `<flow-root>/flow_preds/<saveflow_path>_R<round>/<scene id>/flow{1,2}.npy` (oa_icp.py:183-186);
  * ICP iterations per round {1: 20, 2: 10, 3: 5, 4: 3} (oa_icp.py:175);
  * the three reported flows: input, weighted-Kabsch, object-aware ICP, each with EPE / AccS / AccR / Outlier
    (metrics/flow_metric.py:4-25, computed on the device).
  * with `--frames 4` the SAPIEN / OGC-DR variant: six ordered pairs of the four frames of a scene
    (`[[0,1],[1,0],[1,2],[2,1],[2,3],[3,2]]`, oa_icp.py:148), flows stored as `<dir>/<scene id>.npy` (6, N, 3) next to
    `<dir>.json` {"view_sel": ...} (oa_icp.py:187-191, datasets/dataset_ogcdr.py:147-157) — ogc_amd/utils/flow_store.py.
With `--data-root DIR` the scenes and the input flows are read from, and the refined flows written to, a directory tree in the
reference's layout (ogc_amd/datasets.py: `kittisf` -> <DIR>/data/<id>/..., split file `--mapping`; `ogcdr` -> <DIR>/data/<id>/...,
<DIR>/data/<split>.lst) — the tree the reference's own oa_icp.py works on; tests/golden/flow_store.npz holds what that script
wrote for a three-scene tree and tests/test_flow_store.py replays it here.  Without it the scenes are the seeded synthetic ones
of `train_seg`, and when no predicted flow exists on disk for a scene the ground-truth flow plus noise stands in for the flow
network's prediction.
"""
import argparse
import importlib
import json
import os

import numpy as np
import torch
import yaml

from .metrics.flow_metric import flow_metrics
from .oa_icp import object_aware_icp, weighted_kabsch
from .train_seg import SEGNETS, SyntheticScenes, SyntheticSequenceScenes
from .utils import flow_store

ICP_ITERS = {1: 20, 2: 10, 3: 5, 4: 3}


def flow_dir(flow_root, name, round_):
    return os.path.join(flow_root, "flow_preds", name if round_ is None else "%s_R%d" % (name, round_))


def load_flows(directory, scene_id, fallback):
    """(flow 0->1, flow 1->0) of a scene from `<directory>/<scene id>/flow{1,2}.npy`, or `fallback()`."""
    paths = [os.path.join(directory, scene_id, "flow%d.npy" % v) for v in (1, 2)]
    if all(os.path.exists(p) for p in paths):
        return [torch.from_numpy(np.load(p)) for p in paths]
    return fallback()


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("config")
    ap.add_argument("--round", type=int, default=1)
    ap.add_argument("--test_batch_size", type=int, default=8)
    ap.add_argument("--save", action="store_true")
    ap.add_argument("--saveflow_path", default="flowstep3d")
    ap.add_argument("--flow-root", default="ckpt/synthetic_data")
    ap.add_argument("--synthetic", type=int, default=16, help="number of synthetic scenes")
    ap.add_argument("--device", default="cuda")
    ap.add_argument("--frames", type=int, default=2, help="2: frame pairs (KITTI layout); 4: SAPIEN / OGC-DR sequences")
    ap.add_argument("--data-root", default=None, help="scenes and flows in the reference's directory layout (ogc_amd/datasets.py)")
    ap.add_argument("--split", default="train")
    ap.add_argument("--mapping", default=None, help="kittisf: the split file listing the scene ids (default <data-root>/<split>.txt)")
    args = ap.parse_args(argv)
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    device = torch.device(args.device)

    seg = cfg["segnet"]
    MaskFormer3D = importlib.import_module("ogc_amd.models." + SEGNETS[cfg["dataset"]]).MaskFormer3D
    segnet = MaskFormer3D(n_slot=seg["n_slot"], n_point=seg["n_point"], use_xyz=seg["use_xyz"],
                          n_transformer_layer=seg["n_transformer_layer"],
                          transformer_embed_dim=seg["transformer_embed_dim"],
                          transformer_input_pos_enc=seg["transformer_input_pos_enc"]).to(device)
    weight_path = os.path.join(cfg["save_path"] + "_R%d" % args.round, "best.pth.tar")
    segnet.load_state_dict(torch.load(weight_path, map_location=device)["model_state"])
    segnet.eval()

    if args.data_root is not None:
        return refine_data_root(args, cfg, segnet, device)

    predflow = cfg.get("predflow_path", "flowstep3d")
    in_dir = flow_dir(args.flow_root, predflow, args.round - 1 if args.round > 1 else None)
    out_dir = flow_dir(args.flow_root, args.saveflow_path, args.round)
    if args.save:
        os.makedirs(out_dir, exist_ok=True)
    outdoor = cfg["dataset"] in ("kittisf", "waymo")
    sequences = args.frames > 2
    if sequences:
        assert not outdoor and args.frames == 4, "sequences are the SAPIEN / OGC-DR sample format (4 frames)"
        view_sels = flow_store.SEQUENCE_PAIRS
        scenes = SyntheticSequenceScenes(args.synthetic, seg["n_point"], seg["n_slot"], view_sels, args.frames, seed=1000)
        in_meta = flow_store.read_meta(in_dir)
        if args.save:
            flow_store.write_meta(out_dir, view_sels)
    else:
        view_sels = [[0, 1], [1, 0]]
        scenes = SyntheticScenes(args.synthetic, seg["n_point"], seg["n_slot"], outdoor, seed=1000)
    n_pair = len(view_sels)
    icp_iter = ICP_ITERS[args.round]
    noise = 0.05 if outdoor else 0.005

    thresh = 0.05 if outdoor else 0.01  # epe_norm_thresh per dataset (oa_icp.py:112,116,124)
    sums, count = {"input": 0.0, "kabsch": 0.0, "oa_icp": 0.0}, 0
    per_batch = max(args.test_batch_size // n_pair, 1)  # all pairs of a scene stay in one batch (oa_icp.py:179-180)
    n_scene = args.synthetic
    for start in range(0, n_scene, per_batch):
        ids = list(range(start, min(start + per_batch, n_scene)))
        pc1, pc2, gt, pred = [], [], [], []
        for i in ids:
            g = torch.Generator().manual_seed(77 + i)
            if sequences:
                pc, _, seq_flows = scenes.scene(i)
                for a, b_ in view_sels:
                    truth = scenes.pair_flows(seq_flows, a, b_)[0]
                    stored = flow_store.load_pair(in_dir, "%06d" % i, (a, b_), in_meta) if in_meta is not None else None
                    guess = torch.from_numpy(np.asarray(stored[0])) if stored is not None else \
                        truth + noise * torch.randn(truth.shape, generator=g)
                    pc1.append(pc[a]); pc2.append(pc[b_]); gt.append(truth); pred.append(guess)
                continue
            pcs, _, flows, _ = scenes[i]
            guess = load_flows(in_dir, "%06d" % i, lambda: [flows[v] + noise * torch.randn(flows[v].shape, generator=g)
                                                            for v in (0, 1)])
            for a, b_ in view_sels:
                pc1.append(pcs[a]); pc2.append(pcs[b_]); gt.append(flows[a]); pred.append(guess[a])
        pc1, pc2 = torch.stack(pc1).to(device), torch.stack(pc2).to(device)
        gt, pred = torch.stack(gt).to(device), torch.stack(pred).float().to(device)
        with torch.no_grad():
            mask1, mask2 = segnet(pc1, pc1), segnet(pc2, pc2)
            kabsch = weighted_kabsch(pc1, pred, mask1)
            refined = object_aware_icp(pc1, pc2, pred, mask1, mask2, icp_iter=icp_iter)
        for key, flow in (("input", pred), ("kabsch", kabsch), ("oa_icp", refined)):
            sums[key] = sums[key] + flow_metrics(gt, flow, epe_norm_thresh=thresh) * pc1.shape[0]
        count += pc1.shape[0]
        if args.save:
            host = refined.cpu().numpy()
            for j, i in enumerate(ids):
                if sequences:
                    flow_store.save_sequence(out_dir, "%06d" % i, host[n_pair * j:n_pair * (j + 1)])
                else:
                    flow_store.save_pair(out_dir, "%06d" % i, host[2 * j], host[2 * j + 1])
    report = {"round": args.round, "icp_iter": icp_iter, "pairs": count,
              "metrics": {k: dict(zip(("EPE", "AccS", "AccR", "Outlier"), [round(x, 5) for x in (v / max(count, 1)).tolist()]))
                          for k, v in sums.items()},
              "saved_to": out_dir if args.save else None}
    print(json.dumps(report), flush=True)
    return report


def refine_data_root(args, cfg, segnet, device):
    """The round on a directory tree in the reference's layout (oa_icp.py:141-229): two data sets over the same ordered frame
    pairs — ground-truth flows for the report, predicted flows of the previous round as input — batches of whole scenes,
    refined flow of the first frame of every pair saved through the data set's writer."""
    from .datasets import KITTISceneFlowDataset, OGCDynamicRoomDataset, SapienDataset
    root = args.data_root
    predflow = "flowstep3d" if args.round <= 1 else "flowstep3d_R%d" % (args.round - 1)       # oa_icp.py:143-146
    decentralize = (cfg.get("data") or {}).get("decentralize", False)
    if cfg["dataset"] == "kittisf":
        view_sels, thresh = [[0, 1], [1, 0]], 0.05
        mapping = args.mapping or os.path.join(root, args.split + ".txt")
        kw = dict(data_root=root, mapping_path=mapping, downsampled=True, view_sels=view_sels, decentralize=decentralize)
        test_set, test_set_predflow = KITTISceneFlowDataset(**kw), KITTISceneFlowDataset(predflow_path=predflow, **kw)
    elif cfg["dataset"] == "ogcdr":
        view_sels, thresh = [list(v) for v in flow_store.SEQUENCE_PAIRS], 0.01
        kw = dict(data_root=root, split=args.split, view_sels=view_sels, decentralize=decentralize)
        test_set, test_set_predflow = OGCDynamicRoomDataset(**kw), OGCDynamicRoomDataset(predflow_path=predflow, **kw)
    elif cfg["dataset"] == "sapien":
        # oa_icp.py:105-113: the test split lives in mbs-sapien, the others in mbs-shapepart (a root that already IS one of the
        # two is taken as it is)
        sub = os.path.join(root, "mbs-sapien" if args.split == "test" else "mbs-shapepart")
        root = sub if os.path.isdir(sub) else root
        view_sels, thresh = [list(v) for v in flow_store.SEQUENCE_PAIRS], 0.01
        kw = dict(data_root=root, split=args.split, view_sels=view_sels, decentralize=decentralize)
        test_set, test_set_predflow = SapienDataset(**kw), SapienDataset(predflow_path=predflow, **kw)
    else:
        raise KeyError("no reader for dataset %r" % cfg["dataset"])
    n_frame, batch_size = len(view_sels), args.test_batch_size
    icp_iter = ICP_ITERS[args.round]
    out_dir = flow_dir(root, args.saveflow_path, args.round)
    if args.save:
        assert batch_size % n_frame == 0, "Frame pairs of one scene should be in the same batch"   # oa_icp.py:179-180
        os.makedirs(out_dir, exist_ok=True)
        if cfg["dataset"] in ("ogcdr", "sapien"):
            flow_store.write_meta(out_dir, view_sels)
    sums, count = {"input": 0.0, "kabsch": 0.0, "oa_icp": 0.0}, 0
    n_batch = 0
    for offset, start in enumerate(range(0, len(test_set), batch_size)):
        sids = range(start, min(start + batch_size, len(test_set)))
        samples, preds = [test_set[i] for i in sids], [test_set_predflow[i] for i in sids]
        pc1 = torch.from_numpy(np.stack([s[0][0] for s in samples])).to(device)
        pc2 = torch.from_numpy(np.stack([s[0][1] for s in samples])).to(device)
        gt = torch.from_numpy(np.stack([s[2][0] for s in samples])).to(device)
        pred = torch.from_numpy(np.stack([s[2][0] for s in preds])).to(device)
        with torch.no_grad():
            mask1, mask2 = segnet(pc1, pc1), segnet(pc2, pc2)
            kabsch = weighted_kabsch(pc1, pred, mask1)
            refined = object_aware_icp(pc1, pc2, pred, mask1, mask2, icp_iter=icp_iter)
        # the reference averages per-BATCH metrics (AverageMeter over batches, oa_icp.py:214-221): kept
        for key, flow in (("input", pred), ("kabsch", kabsch), ("oa_icp", refined)):
            sums[key] = sums[key] + flow_metrics(gt, flow, epe_norm_thresh=thresh)
        n_batch += 1
        count += pc1.shape[0]
        if args.save:
            test_set._save_predflow(refined, save_root=out_dir, batch_size=batch_size, n_frame=n_frame, offset=offset)
    report = {"round": args.round, "icp_iter": icp_iter, "pairs": count,
              "metrics": {k: dict(zip(("EPE", "AccS", "AccR", "Outlier"), [round(x, 6) for x in (v / max(n_batch, 1)).tolist()]))
                          for k, v in sums.items()},
              "saved_to": out_dir if args.save else None}
    print(json.dumps(report), flush=True)
    return report


if __name__ == "__main__":
    main()
