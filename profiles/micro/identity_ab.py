"""identity pass A/B by environment (read per call): python profiles/micro/identity_ab.py VAR=val,VAR=val ... (C4 random + wavefront)"""
import ctypes as C, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import motion_primitive_library_amd as m
from motion_primitive_library_amd import _abi
variants = [a for a in sys.argv[1:] if "=" in a or a == "-"]
wl = m.workloads.make("C4")
env = m.EnvMap(3); wl.apply(env)
L = _abi.lib()
lists = env.alloc_lists(wl.n_nodes, want_state=True)
ns = lists.n_slots
heur = m.env.DeviceArray(env, ns * 8); flags = m.env.DeviceArray(env, ns); canon = m.env.DeviceArray(env, ns * 4)
goal = wl.nodes[:, 0].copy()
g = _abi.GoalSpec()
g.goal, g.control, g.w, g.v_max = goal.ctypes.data, wl.control, 10.0, 2.0
g.tol_pos, g.tol_vel, g.tol_acc, g.tol_yaw = 0.5, -1.0, -1.0, -1.0
s = lists.c_struct()
os.environ["MPLX_POST_PARTITION_MIN"] = "0"
def run(want_canon, reps=10):
    o = _abi.Post(); o.heur, o.flags, o.canon = heur.ptr, flags.ptr, canon.ptr if want_canon else None
    for _ in range(2): _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    env.synchronize(); env.timer_begin()
    for _ in range(reps): _abi.check(env._ctx, L.mplx_post_lists_device(env._ctx, C.byref(s), wl.n_nodes, C.byref(g), C.byref(o)))
    return env.timer_end() / reps
out = {}
for label in ("random", "wavefront"):
    nodes = wl.nodes if label == "random" else m.workloads.wavefront_frontier(wl, wl.n_nodes, 0)
    fr = env.upload_frontier(nodes); env.expand_lists_resident(fr, lists); env.synchronize()
    base = run(False); ref = None; rec = {"heur_flags_only_ms": round(base, 4)}
    for rnd in range(2):
        for v in variants:
            kv = {} if v == "-" else dict(x.split("=") for x in v.split(","))
            os.environ.update(kv)
            ms = run(True)
            c = canon.download(np.int32, (ns,))
            cnt = lists.count.download(np.int32, (wl.n_nodes,))
            valid = (np.arange(lists.stride)[None, :] < cnt[:, None]).ravel()
            cv = c[valid]
            if ref is None: ref = cv
            rec.setdefault(v, []).append({"identity_ms": round(ms - base, 4), "form": env.last_identity_form(), "same": bool(np.array_equal(cv, ref))})
            for k in kv: os.environ.pop(k)
    out[label] = rec; fr.free()
print(json.dumps(out, indent=1))
