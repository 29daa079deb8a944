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
SOURCES = ["expand_kernel.hip", "expand_tile_kernel.hip", "expand_grid_kernel.hip", "expand_lex_kernel.hip", "expand_pair_kernel.hip", "map_prep_kernel.hip", "map_prep_api.cpp", "post_kernel.hip", "identity_kernel.hip", "post_api.cpp", "edge_kernel.hip", "edge_api.cpp", "mplx_api.cpp", "planner_capi.cpp", "pack_kernel.hip", "store_model_kernel.hip", "lists_copy_api.cpp", "pack_api.cpp", "comm_api.cpp"]
HEADERS = ["mplx_internal.h", "mplx_ctx.h", "mplx_device_common.h", "mplx_grid_common.h", "host_planner.hpp", "host_lpastar.hpp", os.path.join("..", "..", "include", "mplx.h"), os.path.join("..", "..", "include", "mplx_debug.h")]
ARCH = "gfx950"


def hipcc_path():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (set HIPCC or install ROCm under /opt/rocm)")


def _obj_dir():
    d = os.path.join(CSRC, "build")
    os.makedirs(d, exist_ok=True)
    return d


def is_stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, s) for s in SOURCES + HEADERS]
    return any(os.path.getmtime(d) > t for d in deps)


FLAGS = ["--offload-arch=" + ARCH, "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall"]
EXTRA_FLAGS = {}  # per-source extra flags


def build_variant(out_path, defines, verbose=False):
    """A diagnostic build of the whole library with extra -D flags into `out_path` (never the in-tree libmplx.so):
    e.g. build_variant(".../libmplx_pt.so", ["MPLX_PHASE_TIMING"]) for profiles/micro/phase_times.py."""
    cmd = [hipcc_path()] + FLAGS + ["-D" + d for d in defines] + ["-shared", "-o", out_path] + SOURCES
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc failed:\n%s\n%s" % (proc.stdout, proc.stderr))
    return out_path


def build(force=False, verbose=False, jobs=None):
    """Compile libmplx.so if missing or older than its sources; returns its path.  One object per source
    (csrc/build/, git-ignored), compiled in parallel, only the stale ones; then one link."""
    if not force and not is_stale():
        return LIB
    from concurrent.futures import ThreadPoolExecutor

    hipcc = hipcc_path()
    odir = _obj_dir()
    hdr_t = max(os.path.getmtime(os.path.join(CSRC, h)) for h in HEADERS)
    todo, objs = [], []
    for src in SOURCES:
        obj = os.path.join(odir, src + ".o")
        objs.append(obj)
        st = max(os.path.getmtime(os.path.join(CSRC, src)), hdr_t)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < st:
            todo.append((src, obj))

    def cc(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(src, []) + ["-x", "hip", "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        return src, subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)

    with ThreadPoolExecutor(max_workers=jobs or min(8, os.cpu_count() or 1)) as ex:
        for src, proc in ex.map(cc, todo):
            if proc.returncode != 0:
                raise RuntimeError("hipcc failed on %s:\n%s\n%s" % (src, proc.stdout, proc.stderr))
            if verbose and proc.stderr.strip():
                print(proc.stderr, file=sys.stderr)
    cmd = [hipcc, "--offload-arch=" + ARCH, "-fPIC", "-shared", "-o", LIB] + objs
    if verbose:
        print(" ".join(cmd), file=sys.stderr)
    proc = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True)
    if proc.returncode != 0:
        raise RuntimeError("hipcc link failed:\n%s\n%s" % (proc.stdout, proc.stderr))
    return LIB


if __name__ == "__main__":
    if "--define" in sys.argv:  # --define A=1,B=2 (or repeated) --out path
        defs = [d for i, a in enumerate(sys.argv[:-1]) if a == "--define" for d in sys.argv[i + 1].split(",")]
        out = sys.argv[sys.argv.index("--out") + 1]
        print(build_variant(os.path.abspath(out), defs, verbose=True))
    else:
        print(build(force="--force" in sys.argv, verbose=True))
