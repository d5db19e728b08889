"""Host-side checks of multi-frame voting beyond the golden fixture: the chain-free propagation equals the reference's
dense chained correspondences for every frame pair and window, two-frame sequences (KITTI-SF's [[0, 1], [1, 0]]), and
the clustering metrics on labels with gaps and perfect predictions."""
import numpy as np
import pytest
import torch


@pytest.fixture()
def cpu_ops(monkeypatch, oracle):
    import ogc_amd.pointnet2.pointnet2 as api
    monkeypatch.setattr(api, "_native", oracle.Pointnet2CudaCPU())
    return api


def _dense_voting(vote, pc, mask, flows, window):
    """mask_voting written directly on collect_correspondences, as the reference does (vote.py:95-131)."""
    corrs = vote.collect_correspondences(pc, flows)
    T = pc.shape[0]
    out = []
    for t in range(T):
        votes = []
        for v in range(max(0, t - window), min(T, t + window + 1)):
            votes.append(mask[t] if v == t else vote.match_mask_by_cost(mask[t], corrs["%d_%d" % (t, v)][0] @ mask[v]))
        m = torch.stack(votes).mean(0)
        out.append(m / m.sum(-1, keepdim=True).clamp(1e-10))
    return torch.stack(out)


@pytest.mark.parametrize("T,window", [(2, 3), (3, 1), (5, 2), (5, 4)])
def test_voting_equals_dense_chains(cpu_ops, T, window):
    from ogc_amd import vote
    from ogc_amd.utils.synthetic import make_sequence
    pc, segm, flows = make_sequence(T, 96, 4, seed=T * 10 + window, outdoor=False)
    g = torch.Generator().manual_seed(1)
    mask = (3.0 * torch.eye(4)[segm] + torch.randn(T, 96, 4, generator=g)).softmax(-1)
    got = vote.mask_voting(pc, mask, flows, time_window_size=window)
    want = _dense_voting(vote, pc, mask, flows, window)
    assert got.shape == (T, 96, 4)
    assert torch.allclose(got, want, rtol=1e-4, atol=1e-6)
    assert torch.allclose(got.sum(-1), torch.ones(T, 96), atol=1e-5)


def test_vote_batch_splits_scenes(cpu_ops):
    from ogc_amd import vote
    from ogc_amd.utils.synthetic import make_sequence
    seqs = [make_sequence(2, 64, 3, seed=s, outdoor=False) for s in (1, 2)]
    pc = torch.cat([s[0] for s in seqs])
    mask = torch.rand(4, 64, 3).softmax(-1)
    # the loader's flows: one (forward, backward) entry per frame, the last of every scene redundant
    flows = torch.cat([torch.cat([s[2], s[2]]) for s in seqs])
    got = vote.vote_batch(pc, mask, flows, n_frame=2, time_window_size=3)
    for i, s in enumerate(seqs):
        want = vote.mask_voting(s[0], mask[2 * i:2 * i + 2], s[2], time_window_size=3)
        assert torch.allclose(got[2 * i:2 * i + 2], want)


def test_clustering_metrics_edge_cases():
    from ogc_amd.metrics.seg_metric import ClusteringMetrics
    segm = torch.tensor([[0, 0, 0, 2, 2, 2, 2, 2], [1, 1, 0, 0, 0, 0, 1, 1]])        # sample 0 has no label 1
    onehot = torch.eye(3)[segm]
    res = ClusteringMetrics()(onehot, segm)                                             # perfect prediction
    assert np.allclose(res["ri"], [1.0, 1.0])
    assert np.isclose(res["iou"][1], 1.0)
    assert np.isclose(res["iou"][0], 2.0 / 3.0)   # the empty label 1 is a GT row with IoU 0, as in the reference
    worst = torch.eye(3)[torch.zeros_like(segm)]                                        # everything in one slot
    res = ClusteringMetrics()(worst, segm)
    # Rand index of "one cluster" against sizes (3, 5): agreeing ordered pairs = 9 + 25 of 64
    assert np.isclose(res["ri"][0], 34.0 / 64.0)
    res = ClusteringMetrics()(onehot, segm, ignore_npoint_thresh=4)                     # the 3-point object is ignored
    assert np.isclose(res["ri"][0], 1.0) and np.isclose(res["iou"][0], 1.0)
