"""Multi-frame co-segmentation by voting (reference: vote.py:17-131) and its evaluation driver (:134-352).

The masks a network predicts for the T frames of a sequence are made consistent over time: every frame collects the
masks of the frames inside its time window, carried over by soft point correspondences (softmax of the negative
distance between the flow-warped frame and the other frame, chained over intermediate frames), aligned in slot order
by a Hungarian match, and averaged.

`pairwise_correspondence`, `collect_correspondences`, `match_mask_by_cost` and `mask_voting` keep the reference's
signatures and results.  What differs underneath:
  * `mask_voting` never forms the chained (N, N) correspondence products (O(N^3) each, vote.py:52-58): only their action
    on the (N, K) masks is needed, and  normalise(A @ B) @ X = (A @ (B @ X)) / clamp(A @ (B @ 1))  gets it with
    matrix-times-thin-matrix products — the normalisers ride along as an extra column.
  * the slot alignment runs on the device (ogc_lsap_maximize, scipy's tie-breaking) instead of a host round trip per
    frame pair, and its cross-entropy cost is two (K, N) x (N, K) products instead of an (N, K, K) tensor.
Run as a module it evaluates a trained segmentation network with and without voting on synthetic sequences
(`python -m ogc_amd.vote <config> --round R`), printing the reference's metrics (AP, PQ, F1, Pre, Rec, mIoU, RI).
"""
import argparse
import os

import numpy as np
import torch

from .losses.seg_loss_unsup import _assign_columns


def pairwise_correspondence(pc1, pc2, flow, temperature=0.01):
    """pc1, pc2, flow (B, N, 3) -> soft correspondence (B, N1, N2), rows sum to one.  Reference: vote.py:17-28."""
    return (-torch.cdist(pc1 + flow, pc2) / temperature).softmax(-1)


def _adjacent(pc, flows):
    """{(t, t+1), (t+1, t)} -> (N, N) correspondence of every adjacent frame pair (vote.py:47-49)."""
    adj = {}
    for t in range(pc.size(0) - 1):
        adj[(t, t + 1)] = pairwise_correspondence(pc[t:t + 1], pc[t + 1:t + 2], flows[t:t + 1, 0])[0]
        adj[(t + 1, t)] = pairwise_correspondence(pc[t + 1:t + 2], pc[t:t + 1], flows[t:t + 1, 1])[0]
    return adj


def collect_correspondences(pc, flows):
    """pc (T, N, 3), flows (T-1, 2, N, 3) [forward flow of frame t, backward flow of frame t+1] -> dict
    '%d_%d' % (a, b) -> (1, N, N) for all frame pairs, as the reference builds it (vote.py:31-60): identity, adjacent
    pairs from the flows, longer spans by chaining and re-normalising the rows."""
    n_frame, n_point, _ = pc.size()
    corrs = {}
    eye = torch.eye(n_point, device=pc.device).unsqueeze(0)
    for t in range(n_frame):
        corrs['%d_%d' % (t, t)] = eye
    for (a, b), m in _adjacent(pc, flows).items():
        corrs['%d_%d' % (a, b)] = m.unsqueeze(0)
    for interval in range(2, n_frame):
        for t in range(0, n_frame - interval):
            corr = torch.bmm(corrs['%d_%d' % (t, t + interval - 1)], corrs['%d_%d' % (t + interval - 1, t + interval)])
            corrs['%d_%d' % (t, t + interval)] = corr / corr.sum(-1, keepdim=True).clamp(1e-10)
            corr = torch.bmm(corrs['%d_%d' % (t + interval, t + interval - 1)], corrs['%d_%d' % (t + interval - 1, t)])
            corrs['%d_%d' % (t + interval, t)] = corr / corr.sum(-1, keepdim=True).clamp(1e-10)
    return corrs


def _carry(adj, t, v, x):
    """corrs['t_v'] @ x without the (N, N) chain:  x (N, c) lives on frame v, the result on frame t.
    Forward spans are nested from the left, corr(t, v) = norm(corr(t, v-1) @ adj(v-1, v)); backward spans from the
    right, corr(t, v) = norm(adj(t, t-1) @ corr(t-1, v))   (vote.py:53-58)."""
    if abs(t - v) == 1:
        return adj[(t, v)] @ x
    ones = torch.ones(x.size(0), 1, dtype=x.dtype, device=x.device)
    c = x.size(1)
    if v > t:
        z = _carry(adj, t, v - 1, adj[(v - 1, v)] @ torch.cat([x, ones], 1))
    else:
        z = adj[(t, t - 1)] @ _carry(adj, t - 1, v, torch.cat([x, ones], 1))
    return z[:, :c] / z[:, c:].clamp(1e-10)


def match_mask_by_cost(mask1, mask2, measure='ce'):
    """mask1, mask2 (N, K) -> mask2 with its slots re-ordered to match mask1: Hungarian on the mean binary cross
    entropy (input mask1, target mask2) or on the soft IoU.  Reference: vote.py:63-92."""
    n_object = mask1.shape[-1]
    if measure == 'ce':
        # mean_n BCE(mask1[n, i], mask2[n, j]) with torch's clamp of the logarithms at -100
        log_p = torch.log(mask1).clamp_min(-100.0)
        log_q = torch.log1p(-mask1).clamp_min(-100.0)
        cost = -(log_p.t() @ mask2 + log_q.t() @ (1.0 - mask2)) / mask1.shape[0]
        col_ind = _assign_columns(-cost)
    else:
        intersection = mask1.t() @ mask2
        union = mask1.sum(0).unsqueeze(1) + mask2.sum(0).unsqueeze(0)
        col_ind = _assign_columns(intersection / union.clamp(1e-10))
    perm = torch.eye(n_object, dtype=torch.float32, device=mask2.device)[col_ind]
    return torch.einsum('ij,nj->ni', perm, mask2)


def mask_voting(pc, mask, flows, time_window_size=3):
    """pc (T, N, 3), mask (T, N, K), flows (T-1, 2, N, 3) -> voted masks (T, N, K).  Reference: vote.py:95-131."""
    n_frame = pc.size(0)
    adj = _adjacent(pc, flows)
    voted = []
    for t in range(n_frame):
        votes = []
        for v in range(max(0, t - time_window_size), min(n_frame, t + time_window_size + 1)):
            if v == t:
                votes.append(mask[t])
            else:
                votes.append(match_mask_by_cost(mask[t], _carry(adj, t, v, mask[v])))
        vote = torch.stack(votes, 0).mean(0)
        voted.append(vote / vote.sum(-1, keepdim=True).clamp(1e-10))
    return torch.stack(voted, 0)


def vote_batch(pc, mask, flows, n_frame, time_window_size=3):
    """The evaluation loop's inner step (vote.py:302-310): the batch holds whole scenes of `n_frame` consecutive
    frames; `flows` is the loader's (B, 2, N, 3) [forward, backward] tensor of which the last entry of every scene is
    redundant."""
    out = []
    for sid in range(pc.size(0) // n_frame):
        lo, hi = n_frame * sid, n_frame * (sid + 1)
        out.append(mask_voting(pc[lo:hi], mask[lo:hi], flows[lo:hi - 1].contiguous(), time_window_size))
    return torch.cat(out, 0)


def evaluate_voting(segnet, sequences, n_frame, time_window_size=3, ignore_npoint_thresh=0, device="cuda"):
    """Metrics of the raw and the voted masks over an iterable of (pc (T,N,3), segm (T,N), flows (T-1,2,N,3))
    sequences.  Returns {'raw': {...}, 'voted': {...}} with AP, PQ, F1, Pre, Rec, mIoU, RI (vote.py:284-352)."""
    from .metrics.seg_metric import ClusteringMetrics, accumulate_eval_results, calculate_AP, calculate_PQ_F1
    clustering = ClusteringMetrics()
    acc = {k: {'iou': [], 'matched': [], 'conf': [], 'n_gt': 0, 'miou': [], 'ri': []} for k in ('raw', 'voted')}
    segnet.eval()
    for pc, segm, flows in sequences:
        pc, segm, flows = pc.to(device), segm.to(device), flows.to(device)
        with torch.no_grad():
            mask = segnet(pc, pc).detach()
            both = {'raw': mask, 'voted': mask_voting(pc, mask, flows, time_window_size)}
        for key, m in both.items():
            a = acc[key]
            iou, matched, conf, n_gt = accumulate_eval_results(segm, m, ignore_npoint_thresh)
            a['iou'].append(iou); a['matched'].append(matched); a['conf'].append(conf); a['n_gt'] += n_gt
            scan = clustering(m, segm.long(), ignore_npoint_thresh)
            a['miou'].append(np.mean(scan['iou'])); a['ri'].append(np.mean(scan['ri']))
    out = {}
    for key, a in acc.items():
        iou, matched, conf = (np.concatenate(a[k]) for k in ('iou', 'matched', 'conf'))
        pq, f1, pre, rec = calculate_PQ_F1(iou, matched, a['n_gt'])
        out[key] = {'AP': calculate_AP(matched, conf, a['n_gt']), 'PQ': pq, 'F1': f1, 'Pre': pre, 'Rec': rec,
                    'mIoU': float(np.mean(a['miou'])), 'RI': float(np.mean(a['ri']))}
    return out


def main(argv=None):
    import yaml
    from .train_seg import build_segnet
    from .utils.synthetic import make_sequence
    ap = argparse.ArgumentParser(description="evaluate a trained segmentation network with multi-frame voting")
    ap.add_argument('config', type=str)
    ap.add_argument('--round', type=int, default=0)
    ap.add_argument('--time_window_size', type=int, default=3)
    ap.add_argument('--n_frame', type=int, default=4)
    ap.add_argument('--n_sequence', type=int, default=4)
    ap.add_argument('--device', type=str, default='cuda')
    args = ap.parse_args(argv)
    with open(args.config) as f:
        cfg = yaml.safe_load(f)
    segnet = build_segnet(cfg).to(args.device)
    weight_path = os.path.join(cfg['save_path'] + '_R%d' % args.round, 'best.pth.tar')   # where train_seg writes
    segnet.load_state_dict(torch.load(weight_path, map_location=args.device)['model_state'])
    print('Loaded weights from', weight_path)
    outdoor = cfg['dataset'] in ('kittisf', 'kittidet', 'waymo')
    seqs = (make_sequence(args.n_frame, cfg['segnet']['n_point'], cfg['segnet']['n_slot'], seed=10_000 + i,
                          outdoor=outdoor) for i in range(args.n_sequence))
    res = evaluate_voting(segnet, seqs, args.n_frame, args.time_window_size,
                          ignore_npoint_thresh=50 if outdoor else 0, device=args.device)
    for key in ('raw', 'voted'):
        print('%-5s ' % key + '  '.join('%s %.4f' % kv for kv in res[key].items()))
    return res


if __name__ == '__main__':
    main()
