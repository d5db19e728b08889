"""Uninitialised reads on the FlowStep3D training path: every float tensor torch.empty / empty_like / new_empty hands out on the
device is filled with NaN first, then the flow trainer replay of tests/test_driver_golden.py runs in this process.  A kernel that
reads memory nobody wrote turns a monitored value or a gradient into NaN.  (development tool)
    python tools/flow_poison.py [pytest -k expression]"""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VALUE = float(os.environ.get("POISON", "nan"))
_empty, _empty_like, _new_empty = torch.empty, torch.empty_like, torch.Tensor.new_empty


INT_VALUE = os.environ.get("POISON_INT")   # e.g. 1: a valid index everywhere, so nothing faults — but results move if it is read


def _poison(t):
    if isinstance(t, torch.Tensor) and t.is_cuda and t.numel():
        if t.is_floating_point():
            t.fill_(VALUE)
        elif INT_VALUE is not None and t.dtype in (torch.int32, torch.int64, torch.int16):
            t.fill_(int(INT_VALUE))
    return t


torch.empty = lambda *a, **k: _poison(_empty(*a, **k))
torch.empty_like = lambda *a, **k: _poison(_empty_like(*a, **k))
torch.Tensor.new_empty = lambda self, *a, **k: _poison(_new_empty(self, *a, **k))
import pytest
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_driver_golden.py"), "-m", "gpu", "-x", "-q", "-s", "-k",
                      sys.argv[1] if len(sys.argv) > 1 else "train_flow_trainer_replays_the_reference_trainer_gpu"]))
