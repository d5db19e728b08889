"""Independent numpy statements of the operator semantics (second opinion for the oracle).

These are NOT the oracle: they are written from the semantic summary,
in a different style (sorting / masks instead of loops), and are only used to cross-check
oracle/ogc_oracle.c on small inputs.
"""
import numpy as np


def sqdist_matrix(a, b):
    """(n,3),(m,3) -> (n,m) fp32, the reference's rounding sequence: ((dx*dx)+(dy*dy))+(dz*dz)."""
    a = a.astype(np.float32)[:, None, :]
    b = b.astype(np.float32)[None, :, :]
    d = a - b
    sq = d * d
    return (sq[..., 0] + sq[..., 1]) + sq[..., 2]


def knn_np(k, unknown, known):
    B, n, _ = unknown.shape
    m = known.shape[1]
    d2 = np.full((B, n, k), np.inf, np.float32)
    idx = np.zeros((B, n, k), np.int32)
    for b in range(B):
        D = sqdist_matrix(unknown[b], known[b])
        order = np.argsort(D, axis=1, kind="stable")  # ties -> lower index first
        kk = min(k, m)
        sel = order[:, :kk]
        dsel = np.take_along_axis(D, sel, 1)
        ok = np.isfinite(dsel)
        # non-finite distances are never inserted; finite ones are a prefix after sorting
        for q in range(n):
            c = int(ok[q].sum())
            d2[b, q, :c] = dsel[q, :c]
            idx[b, q, :c] = sel[q, :c]
    return d2, idx


def ball_query_np(radius, nsample, xyz, new_xyz):
    B, n, _ = xyz.shape
    m = new_xyz.shape[1]
    r2 = np.float32(radius) * np.float32(radius)
    idx = np.zeros((B, m, nsample), np.int32)
    for b in range(B):
        D = sqdist_matrix(new_xyz[b], xyz[b])
        for q in range(m):
            hits = np.nonzero(D[q] < r2)[0][:nsample]
            if len(hits):
                idx[b, q, :] = hits[0]
                idx[b, q, :len(hits)] = hits
    return idx


def _bitrev(v, bits):
    r = 0
    for _ in range(bits):
        r = (r << 1) | (v & 1)
        v >>= 1
    return r


def fps_np(xyz, m):
    """FPS with the reference's tie order stated as a sort key:
    among equal maxima the winner minimises (bitrev(k % bs), k // bs), bs = min(1024, 2^floor(log2 N))."""
    B, N, _ = xyz.shape
    bs = min(1024, 1 << int(np.floor(np.log2(N))))
    bits = int(np.log2(bs))
    ks = np.arange(N)
    rank = np.array([_bitrev(int(k % bs), bits) for k in ks], dtype=np.int64) * (N // bs + 1) + ks // bs
    out = np.zeros((B, m), np.int32)
    for b in range(B):
        temp = np.full(N, 1e10, np.float32)
        old = 0
        for j in range(1, m):
            d = sqdist_matrix(xyz[b], xyz[b, old:old + 1])[:, 0]
            temp = np.minimum(d, temp)
            mx = temp.max()
            cand = np.nonzero(temp == mx)[0]
            old = int(cand[np.argmin(rank[cand])])
            out[b, j] = old
    return out
