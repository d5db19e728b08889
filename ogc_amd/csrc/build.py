"""Builds libogc_ops.so (HIP, gfx950 only) in-tree with plain hipcc — no cmake, no hipify.

    python ogc_amd/csrc/build.py [--force] [--verbose]

The .so stays next to the sources (git-ignored, but shipped to the GPU box by gpurun).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["api.hip", "det.hip", "knn.hip", "ball_query.hip", "fps.hip", "gather_group.hip", "interpolate.hip", "kabsch.hip", "small_solvers.hip", "neighbour_loss.hip", "soft_nn.hip", "seg_loss.hip", "group_norm.hip", "batch_norm.hip", "grid.hip", "conv1x1.hip", "conv1x1_h.hip", "conv1x1_wgrad.hip", "attention.hip", "slot_masks.hip", "small_linear.hip", "loss_glue.hip", "mlp_chain.hip", "gn_fused_bwd.hip", "adam.hip", "gemm_chunk.hip", "chamfer.hip", "flow_step.hip"]
HEADERS = ["ogc_common.h", "grid.h", "conv_stage.h", "conv1x1_shared.h", "act_io.h", "conv1x1_epilogue.h", os.path.join("..", "..", "include", "ogc_ops.h")]
LIB = os.path.join(HERE, "libogc_ops.so")
# The search kernels once more with the distance expression contracted as `nvcc --fmad=true` contracts it (ogc_common.h, OGC_FMAD):
# libogc_ops_fmad.so = those objects + all the others of libogc_ops.so.  Opt-in at load time (OGC_FMAD=1, ogc_amd/_lib.py).
FMAD_SOURCES = ["knn.hip", "ball_query.hip", "fps.hip", "grid.hip"]
LIB_FMAD = os.path.join(HERE, "libogc_ops_fmad.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC",
    "-ffp-contract=off",        # distance arithmetic is pinned to the un-fused fp32 sequence
    "-munsafe-fp-atomics",      # scatter-add gradients use hardware fp32 atomic add
    # MFMA accumulators may be allocated in ordinary VGPRs (gfx90a and later take either file).  The default forces them
    # into AGPRs, and kernels whose accumulators cross a loop back-edge or a branch then copy them AGPR -> VGPR -> AGPR every
    # round (conv1x1_wgrad_kernel<4,4>: 128 v_accvgpr moves per 128 MFMAs); with the VGPR form the copies disappear.
    "-mllvm", "-amdgpu-mfma-vgpr-form=1",
    "-Wall", "-Wno-unused-function",
]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    hdrs = [os.path.join(HERE, h) for h in HEADERS] + [os.path.abspath(__file__)]
    objs, objs_fmad, jobs = [], [], []
    for s in SOURCES:
        src = os.path.join(HERE, s)
        obj = os.path.join(HERE, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs):
            jobs.append([HIPCC] + FLAGS + ["-c", src, "-o", obj])
        if s in FMAD_SOURCES:
            obj_f = os.path.join(HERE, s.replace(".hip", ".fmad.o"))
            objs_fmad.append(obj_f)
            if force or _newer(obj_f, [src] + hdrs):
                jobs.append([HIPCC] + FLAGS + ["-DOGC_FMAD=1", "-c", src, "-o", obj_f])
        else:
            objs_fmad.append(obj)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _newer(LIB, objs):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    if force or jobs or _newer(LIB_FMAD, objs_fmad):
        run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_FMAD] + objs_fmad)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
