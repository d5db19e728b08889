"""Furthest point sampling along a chain of levels (ogc_furthest_point_sampling_chain): the shortcut for tie-free parents
and the fall-back for everything else both return what plain FPS (and the CPU oracle) return on the same cloud."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _api():
    import ogc_amd  # noqa: F401
    from ogc_amd.pointnet2 import pointnet2 as api
    return api


def lattice(n_side):
    g = torch.arange(n_side, dtype=torch.float32)
    return torch.stack(torch.meshgrid(g, g, g, indexing="ij"), -1).reshape(-1, 3)


def chain(api, pc, sizes):
    """Levels of a chain -> list of (idx via the chain entry point, idx via the plain operator, ties)."""
    out, ties, cur = [], None, pc
    for m in sizes:
        idx, t = api.furthest_point_sample_chain(cur, m, ties)
        plain = api.furthest_point_sample(cur, m)
        out.append((idx, plain, t))
        cur = api.gather_nd(cur, idx.long()).contiguous()
        ties = t
    return out


def test_tie_free_clouds_skip_the_rounds_and_match(oracle):
    api = _api()
    g = torch.Generator().manual_seed(3)
    pc = ((torch.rand(4, 8192, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous()
    levels = chain(api, pc, [2048, 1024, 512])
    for lvl, (idx, plain, ties) in enumerate(levels):
        assert torch.equal(idx, plain), "level %d" % lvl
        assert ties.dtype == torch.int32 and bool((ties >= idx.shape[1]).all())   # no tie before the last round
    for idx, _, _ in levels[1:]:  # the shortcut's answer: the first m samples of the parent, in order
        assert torch.equal(idx, torch.arange(idx.shape[1], device="cuda", dtype=torch.int32).expand_as(idx))
    # level 2 against the oracle on the gathered centres (the plain operator is pinned to it elsewhere)
    centres = api.gather_nd(pc, levels[0][0].long()).contiguous()
    want = oracle.fps(centres.cpu().numpy(), 1024)
    np.testing.assert_array_equal(levels[1][0].cpu().numpy(), want)


@pytest.mark.parametrize("kind", ["lattice", "duplicates", "identical"])
def test_clouds_with_ties_are_sampled_for_real(kind, oracle):
    api = _api()
    if kind == "lattice":
        pc = lattice(16).unsqueeze(0)                                   # 4096 points, masses of exact ties
    elif kind == "duplicates":
        g = torch.Generator().manual_seed(5)
        base = torch.rand(1, 700, 3, generator=g)
        pc = torch.cat([base, base[:, :324]], 1)                        # 1024 points, 324 of them twice
    else:
        pc = torch.ones(1, 512, 3)
    pc = pc.cuda().contiguous()
    n = pc.shape[1]
    levels = chain(api, pc, [n // 2, n // 4, n // 8])
    assert int(levels[0][2][0]) < n // 4                                 # the parent run saw a tie early ...
    cur = pc
    for lvl, (idx, plain, ties) in enumerate(levels):                    # ... so every level was really sampled
        assert torch.equal(idx, plain), "%s level %d" % (kind, lvl)
        np.testing.assert_array_equal(idx.cpu().numpy(), oracle.fps(cur.cpu().numpy(), idx.shape[1]))
        cur = api.gather_nd(cur, idx.long()).contiguous()


def test_mixed_batch_decides_per_sample(oracle):
    api = _api()
    g = torch.Generator().manual_seed(9)
    rnd = torch.rand(4096, 3, generator=g) * 15.0
    pc = torch.stack([rnd, lattice(16), rnd.flip(0).contiguous()]).cuda().contiguous()
    levels = chain(api, pc, [1024, 512])
    ties = levels[0][2].cpu().tolist()
    assert ties[0] >= 512 and ties[1] < 512 and ties[2] >= 512
    for idx, plain, _ in levels:
        assert torch.equal(idx, plain)
    ar = torch.arange(512, device="cuda", dtype=torch.int32)
    assert torch.equal(levels[1][0][0], ar) and torch.equal(levels[1][0][2], ar)
    assert not torch.equal(levels[1][0][1], ar)                          # the lattice is NOT an ordered prefix


def test_large_clouds_never_claim_to_be_tie_free():
    api = _api()
    pc = torch.rand(1, 20000, 3, generator=torch.Generator().manual_seed(1)).cuda().contiguous()
    idx, ties = api.furthest_point_sample_chain(pc, 64, None)
    assert torch.equal(idx, api.furthest_point_sample(pc, 64)) and int(ties[0]) == 0


def test_segnet_plan_uses_the_chain():
    import ogc_amd  # noqa: F401
    from ogc_amd.models.segnet_kitti import MaskFormer3D
    torch.manual_seed(0)
    net = MaskFormer3D(n_slot=10, n_point=2048, transformer_embed_dim=128).cuda()
    pc = ((torch.rand(2, 2048, 3) - 0.5) * torch.tensor([60.0, 4.0, 80.0])).cuda().contiguous()
    sa_plans, _ = net._plan(pc)
    ar = lambda m: torch.arange(m, device="cuda").expand(2, m)  # noqa: E731
    assert bool((sa_plans[0]["ties"] >= 512).all())
    assert torch.equal(sa_plans[1]["new_inds"], ar(sa_plans[1]["new_inds"].shape[1]))
    assert torch.equal(sa_plans[2]["new_inds"], ar(sa_plans[2]["new_inds"].shape[1]))


@pytest.mark.parametrize("N,first", [(8192, 2048), (16384, 4096), (12001, 3000)])
def test_first_tie_round_of_the_bucketed_rounds(oracle, N, first):
    """4097 .. 16384-point clouds go through fps_bucket_kernel (64 or 128 buckets), which finds ties from its bucket records
    instead of the lanes' registers: the first tied round it reports must be the one the plain rounds report (run in a second
    process with OGC_FPS_BUCKETS=0), and the chain built on it must match plain sampling level by level."""
    import os, subprocess, sys, tempfile
    api = _api()
    g = torch.Generator().manual_seed(11)
    pc = ((torch.rand(3, N, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0]))
    pc[1, 5000] = pc[1, 17]                     # one duplicate: a tie when the second of the pair would be picked ... never
    pc[2, :N // 2] = torch.round(pc[2, :N // 2])    # lattice half: early ties
    pc = pc.cuda().contiguous()
    levels = chain(api, pc, [first, 512])
    for lvl, (idx, plain, ties) in enumerate(levels):
        assert torch.equal(idx, plain), "level %d" % lvl
    want = oracle.fps(pc.cpu().numpy(), first)
    np.testing.assert_array_equal(levels[0][0].cpu().numpy(), want)
    with tempfile.TemporaryDirectory() as d:
        np.save(os.path.join(d, "pc.npy"), pc.cpu().numpy())
        code = ("import numpy as np, torch, ogc_amd\n"
                "from ogc_amd.pointnet2 import pointnet2 as api\n"
                "pc = torch.from_numpy(np.load(r'%s')).cuda()\n"
                "idx, t = api.furthest_point_sample_chain(pc, %d, None)\n"
                "np.save(r'%s', t.cpu().numpy()); np.save(r'%s', idx.cpu().numpy())\n"
                % (os.path.join(d, "pc.npy"), first, os.path.join(d, "t.npy"), os.path.join(d, "i.npy")))
        env = dict(os.environ, OGC_FPS_BUCKETS="0",
                   PYTHONPATH=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        subprocess.run([sys.executable, "-c", code], check=True, env=env, timeout=300)
        np.testing.assert_array_equal(levels[0][2].cpu().numpy(), np.load(os.path.join(d, "t.npy")))
        np.testing.assert_array_equal(levels[0][0].cpu().numpy(), np.load(os.path.join(d, "i.npy")))


@pytest.mark.parametrize("kind", ["tie_free", "duplicates", "lattice"])
def test_sampling_a_level_again_in_full_continues_the_chain(kind):
    """FlowStep3D's coarse set-abstraction layers sample a level of the pyramid AGAIN with npoint == n (a permutation).  With the
    level's ties the chain entry point skips the rounds the parent run decided without a tie; the indices are those of a full run."""
    api = _api()
    if kind == "tie_free":
        g = torch.Generator().manual_seed(11)
        pc = ((torch.rand(2, 4096, 3, generator=g) - 0.5) * torch.tensor([60.0, 4.0, 80.0]))
    elif kind == "duplicates":
        g = torch.Generator().manual_seed(12)
        base = torch.rand(2, 1500, 3, generator=g)
        pc = torch.cat([base, base[:, :548]], 1)                        # 2048 points, 548 of them twice: ties late in the run
    else:
        pc = lattice(16).unsqueeze(0).repeat(2, 1, 1)
    pc = pc.cuda().contiguous()
    m1 = pc.shape[1] // 2
    idx1, ties1 = api.furthest_point_sample_chain(pc, m1, None)
    level = api.gather_nd(pc, idx1.long()).contiguous()
    for m in (m1, m1 // 2):
        again, _ = api.furthest_point_sample_chain(level, m, ties1)
        assert torch.equal(again, api.furthest_point_sample(level, m)), (kind, m)
    if kind == "tie_free":
        assert torch.equal(again, torch.arange(m1 // 2, device="cuda", dtype=torch.int32).expand_as(again))


def test_flowstep3d_set_abstraction_takes_the_noted_chain():
    """PointNetSetAbstraction inside a geometry_memo scope: a cloud noted as a chain level is sampled through the chain entry
    point (same indices and features as without the note)."""
    _api()
    from ogc_amd.utils.flowstep3d_util import PointNetSetAbstraction, geometry_memo
    api = _api()
    torch.manual_seed(0)
    g = torch.Generator().manual_seed(21)
    base = torch.rand(2, 1800, 3, generator=g)
    pc = torch.cat([base, base[:, :248]], 1).cuda().contiguous()        # (2, 2048, 3) with duplicates
    idx1, ties1 = api.furthest_point_sample_chain(pc, 1024, None)
    level = api.gather_nd(pc, idx1.long()).permute(0, 2, 1).contiguous()   # (2, 3, 1024)
    sa = PointNetSetAbstraction(npoint=1024, radius=None, nsample=8, in_channel=3, mlp=[16, 16], group_all=False,
                                return_fps=True).cuda().eval()
    with torch.no_grad():
        with geometry_memo():
            xyz_a, feat_a, idx_a = sa(level, level)
        with geometry_memo():
            geometry_memo.note_chain(level, ties1)
            xyz_b, feat_b, idx_b = sa(level, level)
    assert torch.equal(idx_a, idx_b) and torch.equal(xyz_a, xyz_b) and torch.equal(feat_a, feat_b)
