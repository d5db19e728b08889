"""CPU tests: the oracle (oracle/ogc_oracle.c) against independent numpy statements."""
import numpy as np
import pytest

from refimpl import ball_query_np, fps_np, knn_np


def _cloud(rng, B, N, scale=(60, 4, 80)):
    return ((rng.random((B, N, 3), dtype=np.float32) - 0.5) * np.array(scale, np.float32)).astype(np.float32)


@pytest.mark.parametrize("n,m,k", [(37, 50, 5), (64, 64, 16), (100, 7, 10), (33, 300, 64), (5, 260, 200), (9, 1, 3)])
def test_knn_matches_stable_sort(oracle, n, m, k):
    rng = np.random.default_rng(n * 1000 + m)
    u, kn = _cloud(rng, 2, n), _cloud(rng, 2, m)
    kn[:, m // 2] = kn[:, 0]  # exact duplicate -> distance ties
    d2, idx = oracle.knn(k, u, kn)
    d2r, idxr = knn_np(k, u, kn)
    assert np.array_equal(idx, idxr)
    assert np.array_equal(d2, d2r)


def test_knn_self_is_first_and_sorted(oracle):
    rng = np.random.default_rng(0)
    pc = _cloud(rng, 1, 300)
    d2, idx = oracle.knn(8, pc, pc)
    assert np.array_equal(idx[0, :, 0], np.arange(300))
    assert (np.diff(d2, axis=-1) >= 0).all()


def test_knn_rejects_k_over_200(oracle):
    pc = np.zeros((1, 4, 3), np.float32)
    with pytest.raises(ValueError):
        oracle.knn(201, pc, pc)


def test_knn_nonfinite_never_selected(oracle):
    u = np.zeros((1, 2, 3), np.float32)
    kn = np.zeros((1, 4, 3), np.float32)
    kn[0, 1, 0] = np.inf
    kn[0, 2, 1] = np.nan
    d2, idx = oracle.knn(4, u, kn)
    assert idx[0, 0].tolist() == [0, 3, 0, 0]
    assert d2[0, 0, :2].tolist() == [0.0, 0.0] and np.isinf(d2[0, 0, 2:]).all()


def test_three_nn_equals_knn3(oracle):
    rng = np.random.default_rng(3)
    u, kn = _cloud(rng, 2, 129), _cloud(rng, 2, 77)
    kn[:, 5] = kn[:, 70]
    a = oracle.three_nn(u, kn)
    b = oracle.knn(3, u, kn)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # m < 3
    a = oracle.three_nn(u, kn[:, :2])
    assert (a[1][..., 2] == 0).all() and np.isinf(a[0][..., 2]).all()


@pytest.mark.parametrize("n,m,ns,r", [(200, 64, 16, 8.0), (512, 100, 64, 20.0), (64, 64, 4, 0.01), (300, 33, 8, 1000.0)])
def test_ball_query(oracle, n, m, ns, r):
    rng = np.random.default_rng(n + m)
    xyz, new = _cloud(rng, 2, n), _cloud(rng, 2, m)
    got = oracle.ball_query(r, ns, xyz, new)
    assert np.array_equal(got, ball_query_np(r, ns, xyz, new))


def test_ball_query_self_and_empty(oracle):
    rng = np.random.default_rng(1)
    xyz = _cloud(rng, 1, 128)
    idx = oracle.ball_query(1e-6, 8, xyz, xyz)  # only the point itself is inside
    assert np.array_equal(idx, np.repeat(np.arange(128, dtype=np.int32)[None, :, None], 8, axis=2))
    far = xyz + 1000.0
    assert (oracle.ball_query(1.0, 8, xyz, far) == 0).all()


@pytest.mark.parametrize("N,m", [(37, 20), (64, 64), (700, 64), (1024, 128), (2048, 96), (1500, 1500)])
def test_fps_tie_order_is_bitreversed_tid(oracle, N, m):
    rng = np.random.default_rng(N)
    xyz = _cloud(rng, 2, N, scale=(1, 1, 1))
    # many exact duplicates => many ties in the running min-distance
    dup = rng.integers(0, N, size=N // 3)
    xyz[:, dup] = xyz[:, (dup * 7 + 1) % N]
    got = oracle.fps(xyz, m)
    assert np.array_equal(got, fps_np(xyz, m))
    assert (got[:, 0] == 0).all()


def test_fps_grid_ties(oracle):
    # integer lattice: massive ties between distinct points
    g = np.stack(np.meshgrid(np.arange(8), np.arange(8), np.arange(4), indexing="ij"), -1).reshape(1, -1, 3)
    xyz = g.astype(np.float32)
    got = oracle.fps(xyz, 64)
    assert np.array_equal(got, fps_np(xyz, 64))
    assert len(set(got[0].tolist())) == 64


def test_fps_block_size(oracle):
    for n, bs in [(1, 1), (2, 2), (3, 2), (37, 32), (512, 512), (700, 512), (1023, 512), (1024, 1024), (8192, 1024), (100000, 1024)]:
        assert oracle.fps_block_size(n) == bs


def test_fps_permutation_when_m_equals_n(oracle):
    rng = np.random.default_rng(5)
    xyz = _cloud(rng, 1, 256)
    idx = oracle.fps(xyz, 256)
    assert sorted(idx[0].tolist()) == list(range(256))


def test_gather_group_interpolate(oracle):
    rng = np.random.default_rng(7)
    B, C, N, P, S = 2, 5, 50, 11, 4
    feats = rng.standard_normal((B, C, N)).astype(np.float32)
    gi = rng.integers(0, N, (B, P)).astype(np.int32)
    out = oracle.gather(feats, gi)
    assert np.array_equal(out, np.take_along_axis(feats, gi[:, None, :].repeat(C, 1), 2))
    idx = rng.integers(0, N, (B, P, S)).astype(np.int32)
    g = oracle.group(feats, idx)
    ref = np.stack([feats[b][:, idx[b]] for b in range(B)])
    assert np.array_equal(g, ref)
    # grads = transposes of the forward maps
    go = rng.standard_normal(g.shape).astype(np.float32)
    gg = oracle.group_grad(go, idx, N)
    ref = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(ref[b, c], idx[b].ravel(), go[b, c].ravel().astype(np.float64))
    assert np.allclose(gg, ref, rtol=1e-5, atol=1e-6)
    go = rng.standard_normal(out.shape).astype(np.float32)
    g2 = oracle.gather_grad(go, gi, N)
    ref = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(ref[b, c], gi[b], go[b, c].astype(np.float64))
    assert np.allclose(g2, ref, rtol=1e-5, atol=1e-6)
    # three_interpolate
    i3 = rng.integers(0, N, (B, P, 3)).astype(np.int32)
    w = rng.random((B, P, 3)).astype(np.float32)
    o = oracle.three_interpolate(feats, i3, w)
    ref = np.stack([(feats[b][:, i3[b]] * w[b][None]) for b in range(B)])
    ref = (ref[..., 0] + ref[..., 1]) + ref[..., 2]
    assert np.array_equal(o, ref.astype(np.float32))
    go = rng.standard_normal(o.shape).astype(np.float32)
    g3 = oracle.three_interpolate_grad(go, i3, w, N)
    ref = np.zeros((B, C, N), np.float64)
    for b in range(B):
        for c in range(C):
            np.add.at(ref[b, c], i3[b].ravel(), (go[b, c][:, None] * w[b]).ravel().astype(np.float64))
    assert np.allclose(g3, ref, rtol=1e-5, atol=1e-6)
