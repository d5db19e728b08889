"""SHA-256 of the oracle's furthest-point-sampling indices on raw-scan sized clouds (N = 100000 and 180000 -> 8192 samples:
the callers of utils/data_util.py:8-20, data_prepare/kittisf/downsample_kittisf.py:25,49), written to
tests/golden/fps_large_sha.json.  The clouds are tests/golden/detgen.py's (integer hashes: the GPU test rebuilds them bit for
bit); the second one has a quarter of its points rounded onto a 1 m lattice, i.e. thousands of exact distance ties that the
reference's block reduction resolves by bit-reversed thread id (sampling_gpu.cu:86-91,136-137).

    python tests/golden/make_fps_large.py        (CPU only; ~1 min)"""
import hashlib
import json
import os
import sys
import time

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import detgen  # noqa: E402
from oracle import oracle as orc  # noqa: E402

CASES = [("n100000", 100000, 8192, 601, 0.0), ("n180000_lattice", 180000, 8192, 602, 0.25)]


def cloud(n, seed, lattice):
    pc = detgen.cloud(1, n, seed)
    k = int(n * lattice)
    pc[0, :k] = np.round(pc[0, :k])
    return np.ascontiguousarray(pc, np.float32)


if __name__ == "__main__":
    orc.build()
    out = {}
    for name, n, m, seed, lattice in CASES:
        t0 = time.time()
        idx = orc.fps(cloud(n, seed, lattice), m)
        out[name] = {"n": n, "m": m, "seed": seed, "lattice": lattice,
                     "sha256": hashlib.sha256(np.ascontiguousarray(idx, np.int32).tobytes()).hexdigest(),
                     "head": [int(v) for v in idx[0, :8]], "distinct": int(len(np.unique(idx)))}
        print(name, out[name], "%.1f s" % (time.time() - t0))
    json.dump(out, open(os.path.join(HERE, "fps_large_sha.json"), "w"), indent=1, sort_keys=True)
