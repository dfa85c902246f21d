"""Builds libdreamzs.so (the gfx950 engine) in-tree with hipcc.

    python -m pydream_amd.build [--force]

hipcc cross-compiles without a GPU.  The library is git-ignored but travels to the GPU box
with the working tree.  -ffp-contract=off is part of the numerical contract:
every fused multiply-add in the kernels is an explicit fma().
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SRC = os.path.join(HERE, "csrc", "dz_engine.hip")
DEPS = [SRC, os.path.join(HERE, "csrc", "dz_kernels.h"), os.path.join(HERE, "csrc", "dz_device.h"),
        os.path.join(HERE, "csrc", "dz_megakernel.h"),
        os.path.join(ROOT, "include", "dreamzs.h")]
LIB = os.path.join(HERE, "libdreamzs.so")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared",
         "-Wno-unused-value", "-Wno-unused-result"]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def build(force=False, verbose=False):
    if not force and os.path.exists(LIB) and all(os.path.getmtime(LIB) >= os.path.getmtime(d) for d in DEPS):
        return LIB
    cmd = [hipcc()] + FLAGS + ["-o", LIB, SRC, "-ldl"]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
