"""Instance-segmentation metrics (reference: metrics/seg_metric.py:8-161): per-prediction IoU / matched flag /
confidence accumulated over batches, then AP (MS-COCO 101-point), PQ, F1, precision, recall.

`ClusteringMetrics` (mean IoU under the best one-to-one matching, Rand index; :167-243) completes the set.

The reference evaluates every sample on the host with numpy loops over (GT instance, predicted instance) pairs — inside
the training loop (train_seg.py:171).  Here the per-batch part is a handful of tensor ops on the device (one bincount
gives all intersections) and ONE device->host copy of the small per-prediction table; the curve arithmetic of AP / PQ
runs on those few hundred numbers on the host, as in the reference.
"""
import numpy as np
import torch


def _prediction_table(segm, mask, ignore_npoint_thresh=0):
    """segm (B, N) integer labels, mask (B, N, K) soft masks -> device tensors
    iou (B, K), confidence (B, K), valid_pred (B, K) bool, n_gt (B,)  — eval_segm (seg_metric.py:38-95) for all samples
    at once.  Labels may be arbitrary integers; they are ranked per sample on the device."""
    segm, mask = segm.detach(), mask.detach()
    B, N, K = mask.shape
    dev = mask.device
    pred = mask.argmax(dim=2)                                                    # (B, N)
    # compress GT labels per sample to 0..G-1 without leaving the device: rank among the sorted distinct labels
    sorted_lab, order = segm.sort(dim=1)
    new_run = torch.ones_like(sorted_lab, dtype=torch.bool)
    new_run[:, 1:] = sorted_lab[:, 1:] != sorted_lab[:, :-1]
    rank_sorted = new_run.long().cumsum(dim=1) - 1
    gt = torch.empty_like(rank_sorted).scatter_(1, order, rank_sorted)           # (B, N) in [0, G_b)
    G = N  # upper bound on instances per sample; rows beyond G_b stay empty
    G = int(min(G, 4 * K + 64)) if N > 4 * K + 64 else G
    gt = gt.clamp_max(G - 1)  # (labels beyond the bound cannot occur for real data: G >> number of objects)
    flat = (torch.arange(B, device=dev).view(B, 1) * G + gt) * K + pred
    inter = torch.bincount(flat.reshape(-1), minlength=B * G * K).view(B, G, K).to(torch.float64)
    gt_sizes = inter.sum(dim=2)                                                  # (B, G)
    pred_sizes = inter.sum(dim=1)                                                # (B, K)
    present_gt = gt_sizes > 0
    ignore_gt = present_gt & (gt_sizes < ignore_npoint_thresh)
    ignored_area = (inter * ignore_gt.unsqueeze(2)).sum(dim=1)                   # (B, K)
    present_pred = pred_sizes > 0
    invalid_pred = ignored_area / pred_sizes.clamp_min(1) > 0.5
    pred_sizes_kept = pred_sizes - ignored_area
    valid_pred = present_pred & (pred_sizes_kept > 0) & ~invalid_pred
    keep_gt = present_gt & ~ignore_gt
    # confidence (:78-82).  The reference indexes the points by the COMPACT id of a prediction but the mask by its position
    # among the VALID predictions (`mask[segm_pred == j, j]` after `mask = mask[:, valid_pred]`): once a prediction has
    # been dropped, the j-th valid column is averaged over the points of the j-th PRESENT prediction.  Reproduced as is.
    cross = torch.zeros(B, K, K, dtype=torch.float64, device=dev)
    cross.scatter_add_(1, pred.unsqueeze(2).expand(-1, -1, K), mask.to(torch.float64))   # [b, s, k] = sum_{pred == s} mask[:, k]
    present_first = torch.sort((~present_pred).to(torch.int8), dim=1, stable=True).indices     # present slots, ascending
    filtered_index = (valid_pred.long().cumsum(dim=1) - 1).clamp_min(0)                         # position among valid ones
    source = torch.gather(present_first, 1, filtered_index)                                     # slot whose points are used
    conf_sum = torch.gather(cross, 1, source.unsqueeze(1)).squeeze(1)                           # [b, k] = cross[b, source[b,k], k]
    confidence = conf_sum / torch.gather(pred_sizes, 1, source).clamp_min(1)
    union = gt_sizes.unsqueeze(2) + pred_sizes_kept.unsqueeze(1) - inter
    iou = torch.where(keep_gt.unsqueeze(2), inter / union.clamp_min(1e-300), torch.full_like(inter, -1.0))
    pred_iou = iou.max(dim=1).values.clamp_min(0.0)                              # (B, K)
    return pred_iou, confidence, valid_pred, keep_gt.sum(dim=1)


def accumulate_eval_results(segm, mask, ignore_npoint_thresh=0):
    """Reference signature (seg_metric.py:8-35): -> Pred_IoU (N',), Pred_Matched (N',), Confidence (N',) numpy arrays over
    all valid predictions of the batch (sample-major, slot order) and N_GT_Inst (int)."""
    pred_iou, confidence, valid, n_gt = _prediction_table(segm, mask, ignore_npoint_thresh)
    table = torch.cat([pred_iou.reshape(-1), confidence.reshape(-1), valid.reshape(-1).to(torch.float64),
                       n_gt.to(torch.float64)]).cpu().numpy()                    # the only copy to the host
    nk = pred_iou.numel()
    keep = table[2 * nk:3 * nk] > 0
    iou = table[:nk][keep]
    return iou, (iou >= 0.5).astype(float), table[nk:2 * nk][keep], int(round(table[3 * nk:].sum()))


def calculate_AP(Pred_Matched, Confidence, N_GT_Inst, eps=1e-10):
    """MS-COCO style average precision over 101 recall thresholds (seg_metric.py:101-143, without the plot)."""
    order = np.argsort(-Confidence, kind='mergesort')
    matched = Pred_Matched[order]
    tp, fp = np.cumsum(matched), np.cumsum(1 - matched)
    precisions = tp / np.maximum(tp + fp, eps)
    recalls = tp / N_GT_Inst
    precisions = np.maximum.accumulate(precisions[::-1])[::-1] if len(precisions) else precisions
    thresholds = np.linspace(0, 1, 101, endpoint=True)
    at = np.searchsorted(recalls, thresholds, side='left')
    queried = np.where(at < len(precisions), precisions[np.minimum(at, max(len(precisions) - 1, 0))] if len(precisions) else 0.0, 0.0)
    return float(np.mean(queried))


def calculate_PQ_F1(Pred_IoU, Pred_Matched, N_GT_Inst, eps=1e-10):
    """Panoptic quality, F1, precision, recall (seg_metric.py:146-161)."""
    tp = Pred_Matched.sum()
    tp_iou = Pred_IoU[Pred_Matched > 0].sum()
    fp = Pred_Matched.shape[0] - tp
    fn = N_GT_Inst - tp
    pq = tp_iou / max(tp + 0.5 * fp + 0.5 * fn, eps)
    pre = tp / max(tp + fp, eps)
    rec = tp / max(tp + fn, eps)
    f1 = (2 * pre * rec) / max(pre + rec, eps)
    return pq, f1, pre, rec


class ClusteringMetrics:
    """Per-scan mean IoU (Hungarian-matched) and Rand index of hard predictions against GT labels starting at 0
    (reference: seg_metric.py:167-243, adapted there from MultiBodySync).  ``__call__(mask, segm, thresh)`` ->
    {'iou': [per sample], 'ri': [per sample]}.

    The reference compares two (B, N, N) same-cluster matrices for the Rand index (N = 8192: 268 MB each per sample).
    Both metrics are functions of the (GT, prediction) contingency table alone — of the N_v^2 ordered pairs of valid
    points, the ones on which the two partitions agree number  N_v^2 - sum_g a_g^2 - sum_p b_p^2 + 2 sum_gp n_gp^2
    — so one bincount on the device and one small copy to the host replace them."""
    IOU = 1
    RI = 2

    def __init__(self, spec=None):
        self.spec = [self.IOU, self.RI] if spec is None else spec

    def forward(self, mask, segm, ignore_npoint_thresh=0):
        from scipy.optimize import linear_sum_assignment
        n_batch, k_mask = mask.shape[0], mask.shape[-1]
        gt = segm.reshape(n_batch, -1).detach().long()                               # (B, N)
        pred = mask.reshape(n_batch, -1, k_mask).detach().argmax(dim=-1)             # (B, N)
        n_data = gt.shape[-1]
        n_gt_segms = (gt.max(dim=1).values + 1).cpu().numpy()
        k = int(max(k_mask, n_gt_segms.max()))
        flat = (torch.arange(n_batch, device=gt.device).view(-1, 1) * k + gt) * k + pred
        table = torch.bincount(flat.reshape(-1), minlength=n_batch * k * k).view(n_batch, k, k).cpu().numpy()
        out = {}
        ious, ris = [], []
        for b in range(n_batch):
            n = table[b].astype(np.int64)                                            # [gt, prediction]
            gt_sizes = n.sum(1)
            nonsmall = gt_sizes >= ignore_npoint_thresh
            if ignore_npoint_thresh > 0:
                n = n * nonsmall[:, None]                                            # points of small GT objects drop out of both
            if self.IOU in self.spec:
                matching = n.astype(np.float32)
                union = (n.sum(1).astype(np.float32)[:, None] + n.sum(0).astype(np.float32)[None, :]) - matching
                iou = matching / (union + np.float32(1e-8))
                rows = iou[:n_gt_segms[b]]
                if ignore_npoint_thresh > 0:
                    rows = rows[nonsmall[:n_gt_segms[b]]]
                r, c = linear_sum_assignment(rows, maximize=True)
                ious.append(np.mean(rows[r, c]))
            if self.RI in self.spec:
                n_valid = int(n.sum()) if ignore_npoint_thresh > 0 else n_data
                agree = n_valid * n_valid - int((n.sum(1) ** 2).sum()) - int((n.sum(0) ** 2).sum()) + 2 * int((n ** 2).sum())
                ris.append(agree / (n_valid * n_valid) if n_valid else float('nan'))
        if self.IOU in self.spec:
            out["iou"] = ious
        if self.RI in self.spec:
            out["ri"] = ris
        return out

    __call__ = forward
