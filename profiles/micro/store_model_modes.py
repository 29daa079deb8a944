"""Round 4: the store pattern of a C4 launch on its own (mplx_debug_store_model), varied: order inside a node (step-major as
the kernels write, or row-major: a row's whole 2.5-KB segment at a time), nodes per chunk, workgroups per CU -- on the SAME
allocation, next to the kernel's time.  Is there a pattern that writes the same bytes faster?"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import motion_primitive_library_amd as m

wl = m.workloads.make("C4")
env = m.EnvMap(wl.dim, 0)
wl.apply(env)
fr = env.upload_frontier(wl.nodes)


def timed(fn, k=20):
    for _ in range(5):
        fn()
    env.synchronize()
    env.timer_begin()
    for _ in range(k):
        fn()
    return env.timer_end() / k


for a in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    lists = env.alloc_lists(wl.n_nodes, want_state=True, want_iters=False)
    for _ in range(60):
        env.expand_lists_resident(fr, lists)
    rec = {"alloc": a, "kernel_ms": round(timed(lambda: env.expand_lists_resident(fr, lists)), 4)}
    modes = ((0, 5), (1, 5), (16, 5), (64, 5), (256 + 4, 5), (256 + 1, 5), (256 + 16, 5), (0, 2), (0, 3), (0, 8), (256 + 4, 2), (256 + 4, 8))
    if len(sys.argv) > 2 and sys.argv[2] == "hybrid":
        modes = ((1, 5), (256 + 1, 5), (512 + 1, 5), (1, 4), (512 + 1, 4), (4, 5), (512 + 4, 5))
    for mode, wgs in modes:
        os.environ["MPLX_STORE_MODEL_MODE"] = str(mode)
        os.environ["MPLX_STORE_MODEL_WGS"] = str(wgs)
        rec["%s chunk %d, %d wg/cu" % ("row-major" if mode & 256 else ("hybrid" if mode & 512 else "step-major"), (mode & 255) or 4, wgs)] = round(
            timed(lambda: env.debug_store_model(lists)), 4)
    print(json.dumps(rec), flush=True)
    lists.free()
