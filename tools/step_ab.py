"""A/B of one module-level switch (NAME of ogc_amd.fused, or package.module:NAME) on the bench step, alternating in ONE process on one GPU:
    python tools/step_ab.py SPARSE_POOL_BACKWARD [rounds [value_a value_b]]
prints ms per step with the switch at value_a (True) / value_b (False) for every round (20 timed steps each, a fresh process
per measurement)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import sys, runpy
sys.argv = ['bench.py', '--steps', '20', '--warmup', '5']
import importlib
_mod, _, _name = %r.rpartition(":")
setattr(importlib.import_module(_mod or "ogc_amd.fused"), _name, %s)
runpy.run_path('bench.py', run_name='__main__')
"""


def once(name, value):
    out = subprocess.run([sys.executable, "-c", CODE % (name, value)], cwd=ROOT, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        raise SystemExit(out.stderr[-2000:])
    return json.loads(line[-1])["ms_per_step"]


if __name__ == "__main__":
    name = sys.argv[1]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    va, vb = (sys.argv[3], sys.argv[4]) if len(sys.argv) > 4 else ("True", "False")
    for r in range(rounds):
        on, off = once(name, va), once(name, vb)
        print("%s  %s: %.3f ms   %s: %.3f ms" % (name, va, on, vb, off), flush=True)
