"""Builds libd3d_hip.so (gfx950) in-tree: python -m detection_3d_amd.build"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
LIB = os.path.join(LIBDIR, "libd3d_hip.so")
SOURCES = ["grid.hip", "conv.hip", "conv_ws.hip", "conv_bf16.hip", "bn.hip", "boxes.hip", "roi_align.hip", "backward.hip", "rpn_head.hip", "topk.hip"]
# -ffp-contract=off: the box geometry shares an arithmetic contract with the CPU oracle
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-std=c++17"]


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build_library(force=False, verbose=False):
    os.makedirs(LIBDIR, exist_ok=True)
    objdir = os.path.join(LIBDIR, "obj")
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, "d3d_internal.h"), os.path.join(HERE, "..", "include", "d3d_hip.h")]
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

    def compile_one(src):
        s = os.path.join(CSRC, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            cmd = [hipcc] + FLAGS + ["-c", s, "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
        return o

    with ThreadPoolExecutor(max_workers=4) as ex:
        objs = list(ex.map(compile_one, SOURCES))
    if force or _stale(LIB, objs):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build_library(force="--force" in sys.argv, verbose=True))
