"""Where a node's time goes inside expand_grid_kernel: shader-clock ticks between the kernel's phase markers, summed
over all waves, from a diagnostic build (-DMPLX_PHASE_TIMING; built here into /tmp, loaded through MPLX_LIB).

    python profiles/micro/phase_times.py [C2 C3 C5 C4]      (on the GPU box)
"""
import ctypes as C
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
LIB = os.path.join(ROOT, "profiles", "micro", "libmplx_pt.so")  # git-ignored; built here when missing
SRC = os.path.join(ROOT, "motion_primitive_library_amd", "csrc", "expand_grid_kernel.hip")
if not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
    subprocess.run([sys.executable, "-m", "motion_primitive_library_amd.build", "--define", "MPLX_PHASE_TIMING", "--out", LIB],
                   check=True, cwd=ROOT, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
os.environ["MPLX_LIB"] = LIB
import motion_primitive_library_amd as m  # noqa: E402

NAMES = ["node load / dead test", "T1 axis entries", "prefix tables, yaw masks", "node hash, free-box query issue",
         "A pairs", "pass set-up + sample times", "rows", "box min/max + staging", "D stores + sample loops",
         "loop overhead"]
L = m._abi.lib()
fn = C.CDLL(LIB).mplx_debug_phase_ticks
fn.argtypes = [C.POINTER(C.c_uint64), C.c_int]
out = {}
for name in (sys.argv[1:] or ["C2", "C3", "C5", "C4"]):
    wl = m.workloads.make(name, potential_fn=m.workloads.device_potential_fn(0) if name == "C5" else None)
    env = m.EnvMap(wl.dim, 0)
    wl.apply(env)
    fr = env.upload_frontier(wl.nodes)
    lists = env.alloc_lists(wl.n_nodes, want_state=True)
    env.expand_lists_resident(fr, lists)
    env.synchronize()
    buf = (C.c_uint64 * 16)()
    fn(buf, 1)
    reps = 5
    for _ in range(reps):
        env.expand_lists_resident(fr, lists)
    env.synchronize()
    fn(buf, 0)
    env.close()
    tot = sum(buf[i] for i in range(10))
    out[name] = {NAMES[i]: {"ticks_per_node": buf[i] / reps / wl.n_nodes, "share": buf[i] / tot} for i in range(10)}
    out[name]["ticks_per_node_total"] = tot / reps / wl.n_nodes
    print(name, "ticks per node: %.0f" % out[name]["ticks_per_node_total"])
    for i in range(10):
        print("   %-34s %8.0f  %5.1f %%" % (NAMES[i], buf[i] / reps / wl.n_nodes, 100.0 * buf[i] / tot))
print(json.dumps(out))
