"""Which gradient tensors of the float64-truth tests miss the ordinary bound (reference-fp32 error, stored conditioning, floor) on
this implementation, per fixture, over a few repetitions (atomic accumulation orders vary): the candidates for
tests/golden/gate_flip_tensors.json.   python tools/gate_flip_list.py [reps]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import ogc_amd  # noqa: F401
import golden_cases as gc
torch.backends.cuda.matmul.allow_tf32 = False
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
out = {}
for name, kw, N, B in gc.SEG_CASES:
    need, worst = set(), 0.0
    for _ in range(reps):
        saved = gc.GATE_FLIP_TENSORS
        gc.GATE_FLIP_TENSORS = {}
        try:
            b = gc.truth_segnet("cuda", name, kw, N, B)
        finally:
            gc.GATE_FLIP_TENSORS = saved
        for what, e_o, e_r, ok in b.rows:
            if not ok:
                need.add(what); worst = max(worst, e_o)
    out[name] = {"tensors": sorted(need), "worst": worst}
    print(name, len(need), "worst %.3e" % worst, file=sys.stderr)
print(json.dumps({k: v["tensors"] for k, v in out.items()}, indent=0))
