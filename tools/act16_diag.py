import sys, torch
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import ogc_amd
from ogc_amd.pointnet2 import pointnet2 as api
from test_act16_gpu import _sa_level, rel
nat = api._native
nat.set_matmul_precision("fp32")
o0, f0, g0, _ = _sa_level(False)
nat.set_matmul_precision("bf16")
o1, f1, g1, _ = _sa_level(False)
o2, f2, g2, m = _sa_level(True)
print("bf16 operands vs fp32: out %.3e featgrad %.3e paramgrad %.3e" % (rel(o1, o0), rel(f1, f0), rel(g1, g0)))
print("act16         vs fp32: out %.3e featgrad %.3e paramgrad %.3e" % (rel(o2, o0), rel(f2, f0), rel(g2, g0)))
print("act16 vs bf16 operands: out %.3e featgrad %.3e paramgrad %.3e" % (rel(o2, o1), rel(f2, f1), rel(g2, g1)), m)
