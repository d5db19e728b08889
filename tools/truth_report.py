"""Where does the HIP path's distance to the float64 truth come from?  (tests/golden/truth_f64.npz)

    python tools/truth_report.py [segnet_sapien ...]

Runs the model scenarios of tests/test_truth_f64_gpu.py with every fused kernel family on, all of them off (the
reference's op sequence on torch + the ten base operators), and each family off in turn, and prints the error of the
output and the median / maximum error of the parameter-gradient norms next to the reference's own fp32 errors.
Families are switched off by hiding their entry point from the gates in ogc_amd/fused.py (diagnosis only)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402
import ogc_amd.pointnet2.pointnet2 as api  # noqa: E402

FAMILIES = {
    "group_norm": ["group_norm_fwd_wrapper", "group_norm_maxpool_fwd_wrapper"],
    "conv": ["conv1x1_wgrad_wrapper"],
    "norm_act_conv": ["conv1x1_gemm_affine_wrapper"],
    "batch_norm": ["batch_norm_fwd_wrapper"],
    "slot_masks": ["slot_masks_fwd_wrapper"],
    "attention": ["attention_fwd_wrapper"],
    "small_linear": ["small_linear_fwd_wrapper"],
    "grouped_first": ["group_linear_fwd_wrapper"],
}


class Hiding:
    def __init__(self, real, hidden):
        object.__setattr__(self, "_real", real)
        object.__setattr__(self, "_hidden", set(hidden))

    def __getattr__(self, name):
        if name in self._hidden:
            raise AttributeError(name)
        return getattr(self._real, name)


def summary(budget):
    out = [r for r in budget.rows if not r[0].startswith(("gnorm/", "ghead/"))]
    gn = np.array([(r[1], r[2]) for r in budget.rows if r[0].startswith("gnorm/")])
    gh = np.array([(r[1], r[2]) for r in budget.rows if r[0].startswith("ghead/")])
    s = " ".join("%s %.1e(%.1e)" % (w.split(".")[-1], o, r) for w, o, r, _ in out)
    return "%s | gnorm med %.1e(%.1e) max %.1e(%.1e) | ghead med %.1e(%.1e) max %.1e(%.1e)" % (
        s, np.median(gn[:, 0]), np.median(gn[:, 1]), gn[:, 0].max(), gn[:, 1].max(),
        np.median(gh[:, 0]), np.median(gh[:, 1]), gh[:, 0].max(), gh[:, 1].max())


def main():
    real = api._native
    rows = "--rows" in sys.argv
    which = [a for a in sys.argv[1:] if not a.startswith("--")] or [c[0] for c in gc.SEG_CASES + gc.FLOW_CASES]
    for case in gc.SEG_CASES + gc.FLOW_CASES:
        if case[0] not in which:
            continue
        fn = gc.truth_segnet if case[0].startswith("segnet") else gc.truth_flownet
        print("== %s   ours(reference fp32), relative L2 vs float64" % case[0])
        variants = [("all fused", [])] + [("all off", sum(FAMILIES.values(), []))] + [("-" + k, v) for k, v in FAMILIES.items()]
        for tag, hidden in variants:
            api._native = Hiding(real, hidden) if hidden else real
            try:
                budget = fn("cuda", *case)
                print("%-16s %s" % (tag, summary(budget)), flush=True)
                if rows and tag in ("all fused", "-batch_norm"):
                    for w, o, r, _ in sorted(budget.rows, key=lambda r: -r[1]):
                        print("      %-64s ours %.2e ref32 %.2e" % (w, o, r))
            except Exception as e:  # noqa: BLE001
                print("%-16s failed: %r" % (tag, e))
        api._native = real


if __name__ == "__main__":
    main()
