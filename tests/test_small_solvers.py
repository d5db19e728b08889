"""The loss-side solvers: device LSAP (scipy's tie-breaking) and the symmetric eigenvalue kernel.

CPU: the C restatement of scipy's linear_sum_assignment procedure (oracle/ogc_oracle.c) against scipy itself — the
reference's own dependency (losses/seg_loss_unsup.py:234-239) — on inputs built to tie.
GPU: the HIP kernels against scipy / numpy on the same inputs.
"""
import numpy as np
import pytest
import torch
from scipy.optimize import linear_sum_assignment

from oracle import oracle as orc


def lsap_cases():
    rng = np.random.default_rng(5)
    cases = []
    for k in (1, 2, 3, 5, 10, 16, 33):
        cases.append(rng.random((40, k, k), dtype=np.float32))                       # generic, no ties
        cases.append(rng.integers(0, 3, (60, k, k)).astype(np.float32))               # small integers: many ties
        cases.append((rng.random((60, k, k)) < 0.3).astype(np.float32))               # 0/1
        m = rng.random((60, k, k), dtype=np.float32)                                  # IoU-like: empty slots = zero rows/cols
        m[rng.random((60, k)) < 0.4] = 0
        m = m.transpose(0, 2, 1).copy()
        m[rng.random((60, k)) < 0.4] = 0
        cases.append(m)
        cases.append(np.zeros((2, k, k), np.float32))                                 # constant -> identity (scipy #11602)
        cases.append(np.round(rng.random((40, k, k), dtype=np.float32) * 4) / 4)      # quarter steps
    return cases


def scipy_cols(score):
    return np.stack([linear_sum_assignment(m, maximize=True)[1] for m in score]).astype(np.int32)


def test_oracle_lsap_matches_scipy():
    for score in lsap_cases():
        np.testing.assert_array_equal(orc.lsap_maximize(score), scipy_cols(score))


def test_oracle_lsap_iou_of_hard_masks():
    # the matrices the loss really builds: IoU of two arg-max segmentations with unused slots
    rng = np.random.default_rng(9)
    for n_slot, used in ((10, 4), (10, 10), (8, 1), (20, 7)):
        a = rng.integers(0, used, (12, 500))
        b = (a + (rng.random((12, 500)) < 0.2) * rng.integers(0, n_slot, (12, 500))) % n_slot
        eye = np.eye(n_slot, dtype=np.float32)
        oa, ob = eye[a], eye[b]
        inter = np.einsum('bng,bnp->bgp', oa, ob)
        union = oa.sum(1)[:, :, None] + ob.sum(1)[:, None, :] - inter
        iou = (inter / np.clip(union, 1e-10, None)).astype(np.float32)
        np.testing.assert_array_equal(orc.lsap_maximize(iou), scipy_cols(iou))


def test_oracle_lsap_nonfinite():
    score = np.random.default_rng(1).random((3, 4, 4), dtype=np.float32)
    score[1, 2, 2] = np.nan
    out = orc.lsap_maximize(score)
    assert (out[1] == -1).all()
    np.testing.assert_array_equal(out[[0, 2]], scipy_cols(score[[0, 2]]))


@pytest.mark.gpu
def test_hip_lsap_matches_scipy():
    from ogc_amd import pointnet2_cuda as nat
    for score in lsap_cases():
        s = torch.from_numpy(score).cuda()
        out = torch.empty(score.shape[:2], dtype=torch.int32, device="cuda")
        nat.lsap_maximize_wrapper(score.shape[0], score.shape[1], s, out)
        np.testing.assert_array_equal(out.cpu().numpy(), scipy_cols(score))
    big = np.round(np.random.default_rng(2).random((5000, 10, 10), dtype=np.float32) * 3) / 3
    out = torch.empty((5000, 10), dtype=torch.int32, device="cuda")
    nat.lsap_maximize_wrapper(5000, 10, torch.from_numpy(big).cuda(), out)
    np.testing.assert_array_equal(out.cpu().numpy(), orc.lsap_maximize(big))
    k64 = np.random.default_rng(3).integers(0, 5, (7, 64, 64)).astype(np.float32)
    out = torch.empty((7, 64), dtype=torch.int32, device="cuda")
    nat.lsap_maximize_wrapper(7, 64, torch.from_numpy(k64).cuda(), out)
    np.testing.assert_array_equal(out.cpu().numpy(), scipy_cols(k64))
    bad = np.ones((2, 3, 3), np.float32)
    bad[0, 0, 0] = np.nan
    out = torch.empty((2, 3), dtype=torch.int32, device="cuda")
    nat.lsap_maximize_wrapper(2, 3, torch.from_numpy(bad).cuda(), out)
    assert (out[0] == -1).all() and sorted(out[1].tolist()) == [0, 1, 2]


@pytest.mark.gpu
def test_hip_sym_eigvals():
    from ogc_amd import pointnet2_cuda as nat
    rng = np.random.default_rng(4)
    for k in (1, 2, 3, 10, 20, 64):
        m = rng.standard_normal((9, 300, k))
        m[:, :, k // 2:] *= 1e-4                                   # wide spectrum
        gram = np.einsum('bnk,bnl->bkl', m, m)
        gram[0] = 0.0                                              # all-zero matrix
        if k > 2:
            gram[1] = np.diag(np.arange(k, dtype=np.float64))      # already diagonal, repeated structure
            gram[2][:, -1] = gram[2][-1, :] = 0.0                  # rank deficient
        A = torch.from_numpy(np.tril(gram) + 7.0 * np.triu(np.ones((k, k)), 1)).cuda()   # upper triangle is ignored
        w = torch.empty((9, k), dtype=torch.float64, device="cuda")
        nat.sym_eigvals_wrapper(9, k, A, w)
        ref = np.linalg.eigvalsh(gram)
        scale = np.abs(ref).max(axis=1, keepdims=True) + 1e-300
        assert np.abs(w.cpu().numpy() - ref).max() <= 1e-13 * scale.max(), k
        assert (np.abs(w.cpu().numpy() - ref) / scale).max() < 1e-13
    A = torch.zeros((2, 3, 3), dtype=torch.float64, device="cuda")
    A[0, 1, 0] = float("nan")
    w = torch.empty((2, 3), dtype=torch.float64, device="cuda")
    nat.sym_eigvals_wrapper(2, 3, A, w)
    assert torch.isnan(w[0]).all() and (w[1] == 0).all()
