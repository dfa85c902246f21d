"""Builds libdreamzs.so (the gfx950 engine) in-tree with hipcc.

    python -m pydream_amd.build [--force] [-j N]

hipcc cross-compiles without a GPU.  The library is git-ignored but travels to the GPU box
with the working tree.  -ffp-contract=off is part of the numerical contract:
every fused multiply-add in the kernels is an explicit fma().

The persistent generation kernel has ~300 template instantiations; they are compiled as one
translation unit per row-tile count (csrc/dz_mega_tu.hip with -DDZ_TU_NRT=1..16; 9..16: k_generations_d2, 128 < d <= 256) next to the
engine's own (csrc/dz_engine.hip), in parallel, into pydream_amd/build/*.o, then linked.
Objects are rebuilt only when a file they include has changed.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libdreamzs.so")
HEADERS = [os.path.join(CSRC, h) for h in ("dz_kernels.h", "dz_device.h", "dz_megakernel.h", "dz_mega_launch.h", "dz_megakernel_w4.h", "dz_megakernel_d2.h")] + \
          [os.path.join(ROOT, "include", "dreamzs.h")]
CFLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]
# (object name, source, extra flags, headers it depends on)
UNITS = [("dz_engine.o", os.path.join(CSRC, "dz_engine.hip"), [], HEADERS)] + \
        [("dz_mega_nrt%d.o" % n, os.path.join(CSRC, "dz_mega_tu.hip"), ["-DDZ_TU_NRT=%d" % n], HEADERS[:6]) for n in range(1, 17)]
DEPS = sorted({u[1] for u in UNITS} | set(HEADERS))


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    return not os.path.exists(target) or any(os.path.getmtime(target) < os.path.getmtime(d) for d in deps)


def build(force=False, verbose=False, jobs=None):
    extra = os.environ.get("DREAMZS_CFLAGS", "").split()          # e.g. -DDZ_EXPERIMENTS for the instrumented builds of tools/
    lib = os.environ.get("DREAMZS_BUILD_LIB") or LIB
    objdir = os.environ.get("DREAMZS_BUILD_DIR") or OBJDIR
    if not force and not extra and not _stale(lib, DEPS):
        return lib
    os.makedirs(objdir, exist_ok=True)
    todo = []
    for name, src, flags, hdrs in UNITS:
        obj = os.path.join(objdir, name)
        if force or extra or _stale(obj, [src] + hdrs):
            todo.append([hipcc()] + CFLAGS + flags + extra + ["-c", src, "-o", obj])
    jobs = jobs or int(os.environ.get("DREAMZS_BUILD_JOBS", "0")) or min(len(UNITS), max(1, (os.cpu_count() or 2) - 1))

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
    with ThreadPoolExecutor(max_workers=jobs) as ex:
        list(ex.map(run, todo))
    run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + [os.path.join(objdir, u[0]) for u in UNITS] + ["-ldl"])
    return lib


if __name__ == "__main__":
    j = int(sys.argv[sys.argv.index("-j") + 1]) if "-j" in sys.argv else None
    print(build(force="--force" in sys.argv, verbose=True, jobs=j))
