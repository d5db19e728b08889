"""ctypes binding of libogc_ops.so (C ABI declared in include/ogc_ops.h).

The library is HIP-only.  There is deliberately no CPU fallback: if the shared object is missing
or a tensor is not resident on a HIP device, the call raises.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# OGC_FMAD=1: the variant whose search kernels contract the squared distance as the reference's nvcc build does (include/ogc_ops.h,
# ogc_distance_contracted) — for comparing index tensors against outputs of the real CUDA binary; never the default
FMAD = os.environ.get("OGC_FMAD", "0") == "1"
# OGC_DETERMINISTIC=1: gradients summed in a fixed order (include/ogc_ops.h: ogc_set_deterministic) — a test mode, slower
DETERMINISTIC = os.environ.get("OGC_DETERMINISTIC", "0") == "1"
LIB_PATH = os.path.join(_HERE, "csrc", "libogc_ops_fmad.so" if FMAD else "libogc_ops.so")

_vp = ctypes.c_void_p
_int = ctypes.c_int
_flt = ctypes.c_float
_dbl = ctypes.c_double
_ll = ctypes.c_longlong

# name -> argtypes (stream is always the trailing void*)
SIGNATURES = {
    "ogc_set_deterministic": [_int],
    "ogc_group_linear_fwd_direct": [_int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_get_deterministic": [],
    "ogc_furthest_point_sampling": [_int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_furthest_point_sampling_chain": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_gather_points": [_int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_gather_points_grad": [_int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_knn": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_three_nn": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_three_interpolate": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_three_interpolate_grad": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_points": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_group_points_grad": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_group_reverse_chunk": [_int, _int, _int],
    "ogc_group_reverse": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_three_interpolate_grad_rev": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_view_means": [_int, _vp, _vp, _vp, _vp, _vp],
    "ogc_view_means_grad": [_int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_sum_ranges": [_int, _vp, _vp, _vp, _ll, _vp, _vp],
    "ogc_three_interpolate_grad_rev_bs": [_int, _int, _int, _int, _vp, _ll, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_points_grad_rev": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_points_grad_rev_dwx": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_ball_query": [_int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp],
    "ogc_knn_clamped": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp],
    "ogc_cell_grid_bytes": [_int, _int],
    "ogc_cell_grid_build": [_int, _int, _flt, _vp, _vp, _vp],
    "ogc_ball_query_cells": [_int, _int, _flt, _int, _vp, _vp, _flt, _vp, _vp],
    "ogc_knn_clamped_cells": [_int, _int, _int, _flt, _vp, _vp, _flt, _vp, _vp, _vp],
    "ogc_chamfer_terms": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_chamfer_terms_grad": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_gather_xyz_pair": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_flow_advance": [_int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_linear_cn": [_int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_three_nn_weights": [_int, _int, _int, _vp, _vp, _vp],
    "ogc_soft_corr_flow": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_gru_reset": [_int, _int, _int, _int, _int, _vp, _ll, _vp, _vp, _vp],
    "ogc_gru_blend": [_int, _int, _int, _int, _vp, _ll, _vp, _ll, _vp, _ll, _vp, _vp],
    "ogc_zero_arena_begin": [_vp, _int, _vp],
    "ogc_zero_arena_end": [],
    "ogc_adam_max_tensors": [],
    "ogc_adam_chunk": [],
    "ogc_adam_step": [_int, _int, _vp, _vp, _vp, _vp, _vp, _dbl, _dbl, _dbl, _dbl, _dbl, _vp],
    "ogc_kabsch_rotation": [_int, _vp, _vp, _vp, _vp],
    "ogc_lsap_maximize": [_int, _int, _vp, _vp, _vp],
    "ogc_rigid_moments": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_rigid_translation": [_int, _vp, _vp, _vp, _vp, _vp],
    "ogc_rigid_blend": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_mask_iou": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_matched_distance": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_soft_nn_target": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_concat": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_linear_fwd": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_linear_bwd": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_concat_grad": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_neighbour_consistency_fwd": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_reverse_neighbours": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_neighbour_consistency_bwd": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_sym_eigvals": [_int, _int, _vp, _vp, _vp],
    "ogc_group_norm_fwd": [_int, _int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_norm_bwd": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_norm_maxpool_fwd": [_int, _int, _int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_gemm": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_attention_fwd": [_int, _int, _int, _int, _int, _flt, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp],
    "ogc_attention_bwd": [_int, _int, _int, _int, _int, _flt, _vp, _int, _vp, _int, _vp, _int, _vp, _vp, _vp, _vp, _int,
                          _vp, _int, _vp, _int, _vp],
    "ogc_conv1x1_gemm_affine_pool": [_int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _vp],
    "ogc_group_norm_pool_extremes": [_int, _int, _int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                     _int, _vp],
    "ogc_small_linear_fwd": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_small_linear_bwd": [_int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_wgrad_moments": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_gn_moments_combine": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_dgrad_adjoint": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_norm_maxpool_bwd_sparse": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                          _vp, _vp, _vp, _vp],
    "ogc_group_norm_maxpool_bwd_ext": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                       _vp, _vp, _vp],
    "ogc_conv1x1_wgrad_moments_pooled": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_dgrad_adjoint_pooled": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_mlp_chain_pool_supported": [_int, _int, _int, _int, _int],
    "ogc_mlp_chain_pool": [_int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_corr_layer_pool_supported": [_int, _int, _int, _int, _int],
    "ogc_corr_layer_pool": [_int, _int, _int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                            _vp, _vp, _vp],
    "ogc_slot_masks_ws_floats": [_int, _int, _int, _int],
    "ogc_slot_masks_fwd": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp],
    "ogc_slot_masks_bwd": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_gn_slots": [],
    "ogc_conv1x1_gemm_stats_supported": [_int, _int, _int, _int, _int],
    "ogc_conv1x1_gemm_stream_supported": [_int, _int, _int, _int],
    "ogc_conv1x1_gemm_any": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_set_matmul_precision": [_int],
    "ogc_group_norm_stats_slots": [],
    "ogc_group_norm_bwd_slots": [],
    "ogc_group_norm_coeffs": [_int, _int, _int, _int, _flt, _vp, _vp, _vp, _vp, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_gemm_affine": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_wgrad_affine": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_wgrad_affine_pooled": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_conv1x1_dgrad_pooled": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_batch_norm_fwd": [_int, _int, _int, _flt, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                           _int, _vp],
    "ogc_batch_norm_bwd": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ogc_batch_norm_maxpool_fwd": [_int, _int, _int, _int, _flt, _int, _int, _flt, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp, _vp, _int, _vp],
    "ogc_batch_norm_maxpool_bwd": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp, _vp],
    "ogc_conv1x1_gemm_gnstats": [_int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp],
    "ogc_group_norm_fwd_stats": [_int, _int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _int, _vp],
    "ogc_group_norm_maxpool_fwd_stats": [_int, _int, _int, _int, _int, _flt, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                         _vp, _int, _vp],
    "ogc_conv1x1_wgrad": [_int, _int, _int, _int, _vp, _vp, _vp, _vp],
    "ogc_group_norm_maxpool_bwd": [_int, _int, _int, _int, _int, _int, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                                   _vp, _vp],
}

# the 16-bit-activation forms (include/ogc_ops.h "16-bit activations"): the prototypes of the entry points they shadow
for _n in ("ogc_group_linear_fwd", "ogc_group_points_grad_rev", "ogc_conv1x1_gemm", "ogc_conv1x1_gemm_affine",
           "ogc_conv1x1_gemm_affine_pool", "ogc_conv1x1_wgrad_affine", "ogc_conv1x1_wgrad_affine_pooled",
           "ogc_conv1x1_dgrad_pooled", "ogc_conv1x1_wgrad_moments", "ogc_conv1x1_wgrad_moments_pooled",
           "ogc_conv1x1_dgrad_adjoint", "ogc_conv1x1_dgrad_adjoint_pooled", "ogc_group_norm_bwd",
           "ogc_group_norm_maxpool_bwd_sparse", "ogc_group_points_grad_rev_dwx"):
    SIGNATURES[_n + "_h"] = SIGNATURES[_n]
SIGNATURES["ogc_group_linear_fwd_direct_h"] = SIGNATURES["ogc_group_linear_fwd_direct"]
SIGNATURES["ogc_conv1x1_wgrad_xf_h"] = SIGNATURES["ogc_conv1x1_wgrad"]
SIGNATURES["ogc_group_linear_fwd_pt_h"] = SIGNATURES["ogc_group_linear_fwd"]

HEADER_VERSION = 205   # OGC_VERSION of include/ogc_ops.h the SIGNATURES table above was written against
_lib = None
_fns = {}  # entry point name -> bound ctypes function


class OgcOpsError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle.  Raises if the HIP library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise OgcOpsError(
                "libogc_ops.so not found at %s — build it with `python ogc_amd/csrc/build.py` "
                "(there is no CPU fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = _int
        L.ogc_version.restype = _int
        if L.ogc_version() != HEADER_VERSION:
            raise OgcOpsError("libogc_ops.so at %s is version %d, this package binds version %d of include/ogc_ops.h — rebuild it "
                              "with `python ogc_amd/csrc/build.py --force`" % (LIB_PATH, L.ogc_version(), HEADER_VERSION))
        L.ogc_distance_contracted.restype = _int
        if bool(L.ogc_distance_contracted()) != FMAD:
            raise OgcOpsError("%s evaluates distances %s, OGC_FMAD asked for the other form" %
                              (LIB_PATH, "contracted (fma)" if L.ogc_distance_contracted() else "un-contracted"))
        L.ogc_slot_masks_ws_floats.restype = ctypes.c_longlong
        L.ogc_cell_grid_bytes.restype = ctypes.c_longlong
        L.ogc_last_error.restype = ctypes.c_char_p
        L.ogc_set_deterministic(1 if DETERMINISTIC else 0)
        _lib = L
    return _lib


def set_deterministic(on):
    """Switch the deterministic-gradient mode of the library and of the host layers (fused.py reads this module's flag)."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    load().ogc_set_deterministic(1 if on else 0)


CALL_COUNTS = None   # a collections.Counter() here counts the entry points called (tools/det_probe.py)

# Tracing: OGC_ROCTX=1 brackets every entry point with a roctx range named after the REFERENCE kernel it stands
# for (K1 .. K10 of SURVEY §8: pointnet2/src/*_gpu.cu) or, for the fused extensions, after the entry point itself — what
# `rocprofv3 --marker-trace` shows next to the kernel rows.  Off by default: two library calls per launch.
ROCTX_NAMES = {
    "ogc_furthest_point_sampling": "K1 furthest_point_sampling", "ogc_furthest_point_sampling_chain": "K1 furthest_point_sampling (chain)",
    "ogc_gather_points": "K2 gather_points", "ogc_gather_points_grad": "K3 gather_points_grad",
    "ogc_knn": "K4 knn", "ogc_knn_clamped": "K4 knn (+ sqrt + radius clamp)", "ogc_knn_clamped_cells": "K4 knn (shared cell grid)",
    "ogc_three_nn": "K5 three_nn", "ogc_three_interpolate": "K6 three_interpolate",
    "ogc_three_interpolate_grad": "K7 three_interpolate_grad", "ogc_three_interpolate_grad_rev": "K7 three_interpolate_grad (gather form)",
    "ogc_three_interpolate_grad_rev_bs": "K7 three_interpolate_grad (gather form)",
    "ogc_group_points": "K8 group_points", "ogc_group_points_grad": "K9 group_points_grad",
    "ogc_group_points_grad_rev": "K9 group_points_grad (gather form)",
    "ogc_ball_query": "K10 ball_query", "ogc_ball_query_cells": "K10 ball_query (shared cell grid)",
}
_roctx = None


def _roctx_lib():
    global _roctx
    if _roctx is None:
        _roctx = False
        if os.environ.get("OGC_ROCTX", "0") == "1":
            for name in ("librocprofiler-sdk-roctx.so", "libroctx64.so"):
                try:
                    lib = ctypes.CDLL(os.path.join(os.environ.get("ROCM_PATH", "/opt/rocm"), "lib", name))
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    lib.roctxRangePushA.restype = ctypes.c_int
                    lib.roctxRangePop.restype = ctypes.c_int
                    _roctx = lib
                    break
                except (OSError, AttributeError):
                    continue
    return _roctx


def call(name, *args):
    """Invoke an entry point; non-zero status becomes a Python exception (the reference would
    have printed and exit(-1)'d, e.g. ball_query_gpu.cu:62-66)."""
    if CALL_COUNTS is not None:
        CALL_COUNTS[name] += 1
    fn = _fns.get(name)
    if fn is None:
        fn = _fns[name] = getattr(load(), name)
    tx = _roctx_lib()
    if tx:
        tx.roctxRangePushA(ROCTX_NAMES.get(name, name).encode())
        try:
            rc = fn(*args)
        finally:
            tx.roctxRangePop()
    else:
        rc = fn(*args)
    if rc != 0:
        msg = load().ogc_last_error().decode("utf-8", "replace")
        raise OgcOpsError("%s failed (status %d): %s" % (name, rc, msg))
    return rc
