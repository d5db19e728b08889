"""Error behaviour of the C ABI, entry point by entry point (include/ogc_ops.h): an empty batch is a no-op, missing
buffers are refused with a status and a message — never a launch on null pointers, never a crash.  (The reference
validates only the ball query and exit(-1)s on launch errors; the replacement reports instead.)
The calls run in a child process so that a faulting entry point fails this test by name instead of killing pytest."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r'''
import ctypes, sys
import torch
from ogc_amd import _lib
L = _lib.load()
stream = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, argtypes in sorted(_lib.SIGNATURES.items()):
    if not argtypes or argtypes[-1] is not ctypes.c_void_p:
        continue  # queries and process-wide settings: nothing is launched
    def args(int_value, first_int):
        out, seen_int = [], False
        for t in argtypes[:-1]:
            if t is ctypes.c_int:
                out.append(first_int if not seen_int else int_value); seen_int = True
            elif t in (ctypes.c_float, ctypes.c_double):
                out.append(1.0)
            elif t is ctypes.c_longlong:
                out.append(int_value)   # (batch strides of the FlowStep3D gate kernels)
            else:
                out.append(None)
        return out + [stream]
    fn = getattr(L, name)
    rc_empty = fn(*args(4, 0))          # leading dimension (the batch) zero, nothing else valid: no-op or refusal
    torch.cuda.synchronize()
    rc_null = fn(*args(4, 4))           # plausible shapes, every buffer missing
    msg = L.ogc_last_error().decode("utf-8", "replace")
    torch.cuda.synchronize()
    print("RESULT %s %d %d %s" % (name, rc_empty, rc_null, msg.replace("\n", " ")), flush=True)
print("DONE", flush=True)
'''


def test_every_entry_point_refuses_null_buffers_and_accepts_empty_batches():
    out = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, capture_output=True, text=True, timeout=600)
    lines = [l.split(" ", 4) for l in out.stdout.splitlines() if l.startswith("RESULT ")]
    done = [l[1] for l in lines]
    assert out.returncode == 0 and "DONE" in out.stdout, "crashed after %s\n%s" % (done[-1:] or "start", out.stderr[-1500:])
    from ogc_amd import _lib
    import ctypes
    assert len(lines) == sum(1 for a in _lib.SIGNATURES.values() if a and a[-1] is ctypes.c_void_p)
    for _, name, rc_empty, rc_null, msg in (l + [""] * (5 - len(l)) for l in lines):
        assert int(rc_null) != 0, "%s accepted null buffers" % name
        assert msg.strip(), "%s refused without a message" % name
        # first integer zero (the batch for most entry points): a no-op or an argument refusal, never a failed launch
        assert int(rc_empty) in (0, -1, -3), "%s: status %s with a zero leading dimension" % (name, rc_empty)
        assert int(rc_null) in (-1, -3), "%s: status %s for null buffers" % (name, rc_null)
