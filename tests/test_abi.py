"""CPU tests of the boundary: the C-ABI library loads and exports every symbol include/ogc_ops.h declares
(no compute calls — there is no GPU here), and the product refuses CPU tensors instead of falling back."""
import ctypes
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "ogc_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ogc_[a-z0-9_]+)\s*\(", text)))


def test_header_declares_the_ten_reference_entry_points():
    syms = _declared_symbols()
    for name in ["ogc_ball_query", "ogc_group_points", "ogc_group_points_grad", "ogc_gather_points",
                 "ogc_gather_points_grad", "ogc_furthest_point_sampling", "ogc_knn", "ogc_three_nn",
                 "ogc_three_interpolate", "ogc_three_interpolate_grad"]:
        assert name in syms


def test_library_exports_every_declared_symbol():
    from ogc_amd.csrc import build as b
    lib_path = b.build()
    lib = ctypes.CDLL(lib_path)
    for name in _declared_symbols():
        assert hasattr(lib, name), "libogc_ops.so does not export %s" % name
    lib.ogc_version.restype = ctypes.c_int
    assert lib.ogc_version() >= 100
    header = open(os.path.join(ROOT, "include", "ogc_ops.h")).read()
    import re
    from ogc_amd import _lib
    assert lib.ogc_version() == int(re.search(r"#define OGC_VERSION (\d+)", header).group(1)) == _lib.HEADER_VERSION


def test_python_binding_covers_every_entry_point():
    from ogc_amd import _lib
    declared = set(_declared_symbols()) - {"ogc_version", "ogc_last_error", "ogc_distance_contracted"}
    assert declared == set(_lib.SIGNATURES)


def test_python_binding_matches_the_prototypes():
    """Argument count and kind (int / float / pointer-or-stream) of every ctypes signature against include/ogc_ops.h:
    a missing entry would make ctypes pass the stream as a 32-bit int."""
    from ogc_amd import _lib
    text = open(os.path.join(ROOT, "include", "ogc_ops.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"\s+", " ", text)
    protos = dict(re.findall(r"\b(ogc_[a-z0-9_]+)\s*\(([^()]*)\)\s*;", text))
    kinds = {ctypes.c_void_p: "ptr", ctypes.c_int: "int", ctypes.c_float: "float", ctypes.c_double: "double",
             ctypes.c_longlong: "ll"}
    for name, argtypes in _lib.SIGNATURES.items():
        params = [p.strip() for p in protos[name].split(",") if p.strip() not in ("", "void")]
        want = []
        for p in params:
            if "*" in p or "ogc_stream_t" in p:
                want.append("ptr")
            elif p.startswith("float"):
                want.append("float")
            elif p.startswith("double"):
                want.append("double")
            elif p.startswith("long long"):
                want.append("ll")
            else:
                assert p.startswith("int"), (name, p)
                want.append("int")
        assert [kinds[a] for a in argtypes] == want, (name, protos[name])


def test_drop_in_module_has_the_ten_pybind_names():
    # reference: pointnet2/src/pointnet2_api.cpp:10-25
    from ogc_amd import pointnet2_cuda as m
    for name in ["ball_query_wrapper", "group_points_wrapper", "group_points_grad_wrapper",
                 "gather_points_wrapper", "gather_points_grad_wrapper", "furthest_point_sampling_wrapper",
                 "knn_wrapper", "three_nn_wrapper", "three_interpolate_wrapper",
                 "three_interpolate_grad_wrapper"]:
        assert callable(getattr(m, name))


def test_no_cpu_fallback():
    from ogc_amd.pointnet2.pointnet2 import ball_query, furthest_point_sample, knn
    pc = torch.zeros(1, 8, 3)
    for fn in (lambda: knn(2, pc, pc), lambda: furthest_point_sample(pc, 2), lambda: ball_query(1.0, 2, pc, pc)):
        with pytest.raises(RuntimeError):
            fn()


def test_operator_api_surface():
    # names a `from pointnet2.pointnet2 import *` caller relies on (reference pointnet2.py:42,78,109,140,187,230,260)
    import ogc_amd.pointnet2.pointnet2 as api
    for name in ["furthest_point_sample", "gather_operation", "knn", "three_nn", "three_interpolate",
                 "grouping_operation", "ball_query", "gather_nd", "QueryAndGroup", "GroupAll"]:
        assert name in api.__all__ and hasattr(api, name)


def test_install_drop_in():
    import sys
    import ogc_amd
    saved = {k: sys.modules.get(k) for k in ("pointnet2_cuda", "pointnet2", "pointnet2.pointnet2")}
    try:
        for k in saved:
            sys.modules.pop(k, None)
        ogc_amd.install_drop_in()
        import pointnet2_cuda
        from pointnet2.pointnet2 import knn  # noqa: F401
        assert pointnet2_cuda is ogc_amd.pointnet2_cuda
    finally:
        for k, v in saved.items():
            if v is None:
                sys.modules.pop(k, None)
            else:
                sys.modules[k] = v


def test_roctx_names_cover_the_ten_reference_kernels():
    """with OGC_ROCTX=1 every C-ABI call is bracketed by a roctx range; the ten reference kernels K1..K10
     each have a named range, and every name maps to a declared entry point."""
    from ogc_amd import _lib
    ks = {v.split()[0] for v in _lib.ROCTX_NAMES.values()}
    assert ks == {"K%d" % i for i in range(1, 11)}
    assert set(_lib.ROCTX_NAMES) <= set(_lib.SIGNATURES)
