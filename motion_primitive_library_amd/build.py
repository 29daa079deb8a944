"""Builds csrc/libmplx.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m motion_primitive_library_amd.build [--force]

hipcc cross-compiles without a GPU.  The flags matter for parity:
-ffp-contract=off keeps every a*b+c as two IEEE operations (the reference's CPU
build never fuses, CMakeLists.txt:5-8) and no fast-math flag is ever passed.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(CSRC, "libmplx.so")
SOURCES = ["expand_kernel.hip", "expand_tile_kernel.hip", "expand_grid_kernel.hip", "map_prep_kernel.hip", "map_prep_api.cpp", "post_kernel.hip", "post_api.cpp", "edge_kernel.hip", "edge_api.cpp", "mplx_api.cpp", "planner_capi.cpp", "pack_kernel.hip", "lists_copy_api.cpp"]
HEADERS = ["mplx_internal.h", "mplx_ctx.h", "mplx_device_common.h", "host_planner.hpp", os.path.join("..", "..", "include", "mplx.h")]
ARCH = "gfx950"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False):
    """Compile libmplx.so if missing or older than its sources; returns its path."""
    if not force and not is_stale():
        return LIB
    cmd = [hipcc_path(), "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off",
           "-fPIC", "-shared", "-Wall", "-o", LIB] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (proc.stdout, proc.stderr))
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
