"""ctypes loader for the CPU oracle (oracle/ogc_oracle.c).

TEST INFRASTRUCTURE ONLY — imported by tests/, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py``; never by anything under ``ogc_amd/``.

Two faces:

* numpy functions (``fps``, ``knn``, ``ball_query`` ...) that allocate their outputs with the
  pre-conditions the reference's Python sets (temp = 1e10, idx = 0, grads = 0;
  pointnet2/pointnet2.py:33,251,73,181,224);
* ``Pointnet2CudaCPU`` — an object exposing the ten ``*_wrapper`` names of the reference's
  native module (pointnet2/src/pointnet2_api.cpp:10-25) on CPU torch tensors, so that the
  host-side layers (and, in tests/golden/make_golden.py, the reference's own Python) can be
  driven without a GPU.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libogc_oracle.so")

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)
_int = ctypes.c_int


def build(force=False):
    """Compile libogc_oracle.so with gcc (seconds)."""
    src = os.path.join(_HERE, "ogc_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libogc_oracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_LIB_PATH)
        sig = {
            "oracle_furthest_point_sampling": [_int, _int, _int, _f32p, _f32p, _i32p],
            "oracle_gather_points": [_int, _int, _int, _int, _f32p, _i32p, _f32p],
            "oracle_gather_points_grad": [_int, _int, _int, _int, _f32p, _i32p, _f32p],
            "oracle_knn": [_int, _int, _int, _int, _f32p, _f32p, _f32p, _i32p],
            "oracle_three_nn": [_int, _int, _int, _f32p, _f32p, _f32p, _i32p],
            "oracle_three_interpolate": [_int, _int, _int, _int, _f32p, _i32p, _f32p, _f32p],
            "oracle_three_interpolate_grad": [_int, _int, _int, _int, _f32p, _i32p, _f32p, _f32p],
            "oracle_group_points": [_int, _int, _int, _int, _int, _f32p, _i32p, _f32p],
            "oracle_group_points_grad": [_int, _int, _int, _int, _int, _f32p, _i32p, _f32p],
            "oracle_ball_query": [_int, _int, _int, ctypes.c_float, _int, _f32p, _f32p, _i32p],
            "oracle_lsap_maximize": [_int, _int, _f32p, _i32p],
        }
        for name, argtypes in sig.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _int
        L.oracle_set_threads.argtypes = [_int]
        L.oracle_set_fmad.argtypes = [_int]
        L.oracle_get_fmad.restype = _int
        L.oracle_get_threads.restype = _int
        L.oracle_fps_block_size.argtypes = [_int]
        L.oracle_fps_block_size.restype = _int
        _lib = L
    return _lib


def set_threads(n):
    lib().oracle_set_threads(int(n))


def get_threads():
    return lib().oracle_get_threads()


class fmad:
    """`with oracle.fmad():` — the distance expression contracted the way `nvcc --fmad=true` contracts it (see ogc_oracle.c);
    for counting what the contraction changes, never for parity."""

    def __enter__(self):
        self._prev = lib().oracle_get_fmad()
        lib().oracle_set_fmad(1)

    def __exit__(self, *exc):
        lib().oracle_set_fmad(self._prev)


def fps_block_size(n):
    return lib().oracle_fps_block_size(int(n))


def _f(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def _check(rc, name):
    if rc != 0:
        raise ValueError("%s: invalid argument (rc=%d)" % (name, rc))


# ---------------------------------------------------------------- numpy API
def fps(xyz, m, return_temp=False):
    xyz, px = _f(xyz)
    B, N, _ = xyz.shape
    temp = np.full((B, N), 1e10, dtype=np.float32)
    idx = np.empty((B, m), dtype=np.int32)
    _check(lib().oracle_furthest_point_sampling(B, N, m, px, temp.ctypes.data_as(_f32p),
                                                idx.ctypes.data_as(_i32p)), "fps")
    return (idx, temp) if return_temp else idx


def gather(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    M = idx.shape[1]
    out = np.empty((B, C, M), dtype=np.float32)
    _check(lib().oracle_gather_points(B, C, N, M, pp, pi, out.ctypes.data_as(_f32p)), "gather")
    return out


def gather_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, M = grad_out.shape
    gp = np.zeros((B, C, N), dtype=np.float32)
    _check(lib().oracle_gather_points_grad(B, C, N, M, pg, pi, gp.ctypes.data_as(_f32p)), "gather_grad")
    return gp


def knn(k, unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.empty((B, N, k), dtype=np.float32)
    idx = np.empty((B, N, k), dtype=np.int32)
    _check(lib().oracle_knn(B, N, M, k, pu, pk, d2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p)), "knn")
    return d2, idx


def three_nn(unknown, known):
    unknown, pu = _f(unknown)
    known, pk = _f(known)
    B, N, _ = unknown.shape
    M = known.shape[1]
    d2 = np.empty((B, N, 3), dtype=np.float32)
    idx = np.empty((B, N, 3), dtype=np.int32)
    _check(lib().oracle_three_nn(B, N, M, pu, pk, d2.ctypes.data_as(_f32p), idx.ctypes.data_as(_i32p)), "three_nn")
    return d2, idx


def three_interpolate(points, idx, weight):
    points, pp = _f(points)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, M = points.shape
    N = idx.shape[1]
    out = np.empty((B, C, N), dtype=np.float32)
    _check(lib().oracle_three_interpolate(B, C, M, N, pp, pi, pw, out.ctypes.data_as(_f32p)), "three_interpolate")
    return out


def three_interpolate_grad(grad_out, idx, weight, M):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    weight, pw = _f(weight)
    B, C, N = grad_out.shape
    gp = np.zeros((B, C, M), dtype=np.float32)
    _check(lib().oracle_three_interpolate_grad(B, C, N, M, pg, pi, pw, gp.ctypes.data_as(_f32p)),
           "three_interpolate_grad")
    return gp


def group(points, idx):
    points, pp = _f(points)
    idx, pi = _i(idx)
    B, C, N = points.shape
    _, P, S = idx.shape
    out = np.empty((B, C, P, S), dtype=np.float32)
    _check(lib().oracle_group_points(B, C, N, P, S, pp, pi, out.ctypes.data_as(_f32p)), "group")
    return out


def group_grad(grad_out, idx, N):
    grad_out, pg = _f(grad_out)
    idx, pi = _i(idx)
    B, C, P, S = grad_out.shape
    gp = np.zeros((B, C, N), dtype=np.float32)
    _check(lib().oracle_group_points_grad(B, C, N, P, S, pg, pi, gp.ctypes.data_as(_f32p)), "group_grad")
    return gp


def ball_query(radius, nsample, xyz, new_xyz):
    xyz, px = _f(xyz)
    new_xyz, pn = _f(new_xyz)
    B, N, _ = xyz.shape
    M = new_xyz.shape[1]
    idx = np.zeros((B, M, nsample), dtype=np.int32)
    _check(lib().oracle_ball_query(B, N, M, float(radius), nsample, pn, px, idx.ctypes.data_as(_i32p)),
           "ball_query")
    return idx


def lsap_maximize(score):
    """(P, K, K) float32 scores -> (P, K) int32 columns, scipy.optimize.linear_sum_assignment(maximize=True)[1]."""
    score, ps = _f(score)
    P, K, _ = score.shape
    out = np.zeros((P, K), dtype=np.int32)
    _check(lib().oracle_lsap_maximize(P, K, ps, out.ctypes.data_as(_i32p)), "lsap_maximize")
    return out


# ---------------------------------------------------------------- pointnet2_cuda-shaped face
class Pointnet2CudaCPU:
    """The ten pybind names of the reference's native module, on CPU torch tensors.

    Signatures follow pointnet2/src/pointnet2_api.cpp:10-25 (argument order of each wrapper in
    ball_query.cpp:16, group_points.cpp:11,26, sampling.cpp:11,24,38, interpolate.cpp:14,26,39,55).
    Tensors are filled in place, like the reference's wrappers do.
    """

    @staticmethod
    def _fp(t):
        assert t.dtype.is_floating_point and t.element_size() == 4 and t.is_contiguous() and t.device.type == "cpu"
        return ctypes.cast(t.data_ptr(), _f32p)

    @staticmethod
    def _ip(t):
        assert (not t.dtype.is_floating_point) and t.element_size() == 4 and t.is_contiguous() and t.device.type == "cpu"
        return ctypes.cast(t.data_ptr(), _i32p)

    def ball_query_wrapper(self, b, n, m, radius, nsample, new_xyz, xyz, idx):
        _check(lib().oracle_ball_query(b, n, m, float(radius), nsample, self._fp(new_xyz), self._fp(xyz),
                                       self._ip(idx)), "ball_query")
        return 1

    def group_points_wrapper(self, b, c, n, npoints, nsample, points, idx, out):
        _check(lib().oracle_group_points(b, c, n, npoints, nsample, self._fp(points), self._ip(idx),
                                         self._fp(out)), "group_points")
        return 1

    def group_points_grad_wrapper(self, b, c, n, npoints, nsample, grad_out, idx, grad_points):
        _check(lib().oracle_group_points_grad(b, c, n, npoints, nsample, self._fp(grad_out), self._ip(idx),
                                              self._fp(grad_points)), "group_points_grad")
        return 1

    def gather_points_wrapper(self, b, c, n, npoints, points, idx, out):
        _check(lib().oracle_gather_points(b, c, n, npoints, self._fp(points), self._ip(idx), self._fp(out)),
               "gather_points")
        return 1

    def gather_points_grad_wrapper(self, b, c, n, npoints, grad_out, idx, grad_points):
        _check(lib().oracle_gather_points_grad(b, c, n, npoints, self._fp(grad_out), self._ip(idx),
                                               self._fp(grad_points)), "gather_points_grad")
        return 1

    def furthest_point_sampling_wrapper(self, b, n, m, points, temp, idx):
        _check(lib().oracle_furthest_point_sampling(b, n, m, self._fp(points), self._fp(temp), self._ip(idx)),
               "fps")
        return 1

    def knn_wrapper(self, b, n, m, k, unknown, known, dist2, idx):
        _check(lib().oracle_knn(b, n, m, k, self._fp(unknown), self._fp(known), self._fp(dist2),
                                self._ip(idx)), "knn")

    def three_nn_wrapper(self, b, n, m, unknown, known, dist2, idx):
        _check(lib().oracle_three_nn(b, n, m, self._fp(unknown), self._fp(known), self._fp(dist2),
                                     self._ip(idx)), "three_nn")

    def three_interpolate_wrapper(self, b, c, m, n, points, idx, weight, out):
        _check(lib().oracle_three_interpolate(b, c, m, n, self._fp(points), self._ip(idx), self._fp(weight),
                                              self._fp(out)), "three_interpolate")

    def three_interpolate_grad_wrapper(self, b, c, n, m, grad_out, idx, weight, grad_points):
        _check(lib().oracle_three_interpolate_grad(b, c, n, m, self._fp(grad_out), self._ip(idx),
                                                   self._fp(weight), self._fp(grad_points)),
               "three_interpolate_grad")
