"""Uninitialised-read detector: fill the caching allocator's free blocks with NaN, then run a model scenario.
A kernel that reads memory it (or a predecessor) never wrote shows up as NaN in outputs or gradients."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as gc  # noqa: E402
import tools.truth_report as tr  # noqa: E402


def poison(gib=6):
    blocks = [torch.full((1 << 28,), float("nan"), device="cuda") for _ in range(gib)]       # 1 GiB each
    small = [torch.full((n,), float("nan"), device="cuda") for n in (64, 256, 1024, 4096, 65536, 1 << 20) for _ in range(64)]
    del blocks, small
    torch.cuda.synchronize()


def main():
    names = [a for a in sys.argv[1:]] or ["flownet_sapien"]
    for case in gc.SEG_CASES + gc.FLOW_CASES:
        if case[0] not in names:
            continue
        fn = gc.truth_segnet if case[0].startswith("segnet") else gc.truth_flownet
        print("fresh   ", tr.summary(fn("cuda", *case)))
        poison()
        print("poisoned", tr.summary(fn("cuda", *case)))
        poison()
        print("poisoned", tr.summary(fn("cuda", *case)))


if __name__ == "__main__":
    main()
