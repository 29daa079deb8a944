"""ctypes binding of the C ABI in include/mplx.h (csrc/libmplx.so).

There is no fallback: if the shared library is missing or cannot be loaded the
import of the engine fails with an explicit error, and every compute call
needs a gfx950 device (mplx_create fails without one).
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MPLX_LIB") or os.path.join(HERE, "csrc", "libmplx.so")  # MPLX_LIB: a diagnostic build

OK = 0
ERR_ARG, ERR_HIP, ERR_NO_DEVICE, ERR_STATE = -1, -2, -3, -4

# every symbol include/mplx.h declares, in declaration order
SYMBOLS = [
    "mplx_create", "mplx_destroy", "mplx_last_error", "mplx_abi_version",
    "mplx_set_map", "mplx_edit_map", "mplx_map_upload_bytes", "mplx_read_cells", "mplx_set_potential", "mplx_set_region", "mplx_set_params", "mplx_set_controls",
    "mplx_update_potential_map", "mplx_set_search_region_path",
    "mplx_expand_device", "mplx_expand", "mplx_expand_lists_device", "mplx_expand_lists", "mplx_get_succ",
    "mplx_set_goal", "mplx_post_lists_device", "mplx_post_packed_device",
    "mplx_pack_lists_device", "mplx_comm_unique_id", "mplx_comm_init", "mplx_comm_destroy", "mplx_comm_broadcast_map",
    "mplx_comm_allgather_lists", "mplx_comm_schedule",
    "mplx_check_edges",
    "mplx_device_alloc", "mplx_device_free", "mplx_memcpy_h2d", "mplx_memcpy_d2h", "mplx_memset",
    "mplx_synchronize", "mplx_timer_begin", "mplx_timer_end",
    "mplx_planner_create", "mplx_planner_destroy", "mplx_planner_attach_ctx", "mplx_planner_set_provider",
    "mplx_planner_set_map", "mplx_planner_edit_map", "mplx_planner_set_controls", "mplx_planner_configure", "mplx_planner_plan",
    "mplx_planner_trajectory", "mplx_planner_trajectory_end", "mplx_planner_closed_set", "mplx_planner_open_set", "mplx_planner_last_error", "mplx_planner_timing", "mplx_planner_set_prior_trajectory", "mplx_planner_set_prior_trajectory_potential", "mplx_planner_use_device_heuristic", "mplx_planner_set_lpastar", "mplx_planner_reset", "mplx_planner_linked_nodes", "mplx_planner_update_blocked_nodes", "mplx_planner_update_cleared_nodes", "mplx_planner_sub_state_space", "mplx_planner_set_edge_provider",
    "mplx_selftest_math", "mplx_selftest_forward_state", "mplx_set_lists_route", "mplx_last_lists_route", "mplx_last_grid_kernel", "mplx_last_identity_form", "mplx_debug_store_model", "mplx_yaw_pin_stats", "mplx_service", "mplx_device_info",
]

ROUTE_AUTO, ROUTE_DENSE, ROUTE_TILE, ROUTE_GRID = 0, 1, 2, 3


class Params(C.Structure):
    _fields_ = [
        ("control", C.c_int32), ("reserved", C.c_int32),
        ("dt", C.c_double), ("w", C.c_double), ("wyaw", C.c_double),
        ("v_max", C.c_double), ("a_max", C.c_double), ("j_max", C.c_double), ("yaw_max", C.c_double),
        ("potential_weight", C.c_double), ("gradient_weight", C.c_double),
    ]


class Succ(C.Structure):
    _fields_ = [
        ("status", C.c_void_p), ("cost", C.c_void_p), ("hash", C.c_void_p),
        ("state", C.c_void_p), ("state_stride", C.c_int64), ("iters", C.c_void_p),
    ]


class SuccLists(C.Structure):
    _fields_ = [
        ("count", C.c_void_p), ("action", C.c_void_p), ("cost", C.c_void_p), ("hash", C.c_void_p),
        ("state", C.c_void_p), ("state_stride", C.c_int64), ("iters", C.c_void_p), ("node_stride", C.c_int64),
        ("heur", C.c_void_p), ("flags", C.c_void_p),  # ABI v8: written by the expansion launch (mplx_set_goal)
    ]


class PackedLists(C.Structure):
    _fields_ = [
        ("count", C.c_void_p), ("offs", C.c_void_p), ("action", C.c_void_p), ("cost", C.c_void_p), ("hash", C.c_void_p),
        ("state", C.c_void_p), ("state_stride", C.c_int64), ("capacity", C.c_int64),
    ]


COMM_ID_BYTES = 128
COMM_META = 8
ROWBIT_ACTION, ROWBIT_COST, ROWBIT_HASH, ROWBIT_STATE = 1, 2, 4, 8
COMM_COPY, COMM_SEND, COMM_RECV = 0, 1, 2
ROW_COUNT, ROW_ACTION, ROW_COST, ROW_HASH, ROW_STATE0 = 0, 1, 2, 3, 4


class CommOp(C.Structure):
    _fields_ = [("kind", C.c_int32), ("peer", C.c_int32), ("row", C.c_int32), ("elem", C.c_int32),
                ("src_off", C.c_int64), ("dst_off", C.c_int64), ("bytes", C.c_int64)]


class GoalSpec(C.Structure):
    _fields_ = [("goal", C.c_void_p), ("control", C.c_int32), ("goal_control", C.c_int32), ("w", C.c_double),
                ("v_max", C.c_double), ("tol_pos", C.c_double), ("tol_vel", C.c_double), ("tol_acc", C.c_double),
                ("tol_yaw", C.c_double)]


class EdgesOut(C.Structure):
    _fields_ = [("free_flag", C.c_void_p), ("cost", C.c_void_p), ("cells", C.c_void_p), ("cell_count", C.c_void_p),
                ("cell_cap", C.c_int32), ("outside", C.c_void_p)]


class Post(C.Structure):
    _fields_ = [("heur", C.c_void_p), ("flags", C.c_void_p), ("canon", C.c_void_p)]


class PlannerConfig(C.Structure):
    _fields_ = [
        ("control", C.c_int32), ("max_expand", C.c_int32), ("batch", C.c_int32), ("goal_control", C.c_int32),
        ("dt", C.c_double), ("w", C.c_double), ("v_max", C.c_double), ("epsilon", C.c_double),
        ("tol_pos", C.c_double), ("tol_vel", C.c_double), ("tol_acc", C.c_double), ("tol_yaw", C.c_double),
    ]


class PlanSummary(C.Structure):
    _fields_ = [
        ("ok", C.c_int32), ("expansions", C.c_int32), ("closed", C.c_int32), ("opened", C.c_int32),
        ("nodes", C.c_int32), ("device_launches", C.c_int32), ("pairs", C.c_int64),
        ("cost", C.c_double), ("total_time", C.c_double), ("J", C.c_double * 4),
        ("segments", C.c_int32), ("state_mismatches", C.c_int32),
    ]


class PlanTiming(C.Structure):
    _fields_ = [(k, C.c_double) for k in ("total_ms", "provider_ms", "fill_ms", "pick_ms", "relax_ms", "recover_ms")] + \
               [(k, C.c_int64) for k in ("relaxed", "improved", "pushes", "materialised", "heur_from_device")]


class MplxError(RuntimeError):
    def __init__(self, code, text):
        super().__init__("mplx error %d: %s" % (code, text))
        self.code = code


_lib = None


def lib():
    """Load libmplx.so (once).  Fails loudly when the HIP extension is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "motion_primitive_library_amd: %s is missing. Build it with "
            "`python -m motion_primitive_library_amd.build` (needs hipcc); there is no CPU fallback."
            % LIB_PATH)
    try:
        L = C.CDLL(LIB_PATH)
    except OSError as e:  # e.g. libamdhip64.so not found
        raise ImportError("motion_primitive_library_amd: cannot load %s: %s" % (LIB_PATH, e))
    vp, i32, i64, dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
    sig = {
        "mplx_create": (C.c_int, [C.c_int, C.c_int, C.POINTER(vp)]),
        "mplx_destroy": (None, [vp]),
        "mplx_last_error": (C.c_char_p, [vp]),
        "mplx_abi_version": (C.c_int, []),
        "mplx_set_map": (C.c_int, [vp, vp, vp, vp, dbl]),
        "mplx_set_potential": (C.c_int, [vp, vp]),
        "mplx_set_region": (C.c_int, [vp, vp]),
        "mplx_set_params": (C.c_int, [vp, C.POINTER(Params)]),
        "mplx_set_controls": (C.c_int, [vp, vp, i32, i32]),
        "mplx_edit_map": (C.c_int, [vp, vp, vp, i64]),
        "mplx_map_upload_bytes": (C.c_int, [vp, C.POINTER(C.c_uint64)]),
        "mplx_read_cells": (C.c_int, [vp, C.c_int, vp, i64, vp]),
        "mplx_set_goal": (C.c_int, [vp, C.POINTER(GoalSpec)]),
        "mplx_update_potential_map": (C.c_int, [vp, vp, vp, vp, dbl, vp]),
        "mplx_set_search_region_path": (C.c_int, [vp, vp, i32, i32, vp, vp]),
        "mplx_expand_device": (C.c_int, [vp, vp, i64, i64, C.POINTER(Succ)]),
        "mplx_expand": (C.c_int, [vp, vp, i64, i64, C.POINTER(Succ)]),
        "mplx_expand_lists_device": (C.c_int, [vp, vp, i64, i64, C.POINTER(SuccLists)]),
        "mplx_expand_lists": (C.c_int, [vp, vp, i64, i64, C.POINTER(SuccLists)]),
        "mplx_get_succ": (C.c_int, [vp, vp, vp, vp, vp, C.POINTER(i32)]),
        "mplx_post_lists_device": (C.c_int, [vp, C.POINTER(SuccLists), i64, C.POINTER(GoalSpec), C.POINTER(Post)]),
        "mplx_check_edges": (C.c_int, [vp, vp, vp, i64, i64, C.POINTER(EdgesOut)]),
        "mplx_pack_lists_device": (C.c_int, [vp, C.POINTER(SuccLists), i64, C.POINTER(PackedLists), C.POINTER(i64)]),
        "mplx_post_packed_device": (C.c_int, [vp, C.POINTER(PackedLists), i64, C.POINTER(GoalSpec), C.POINTER(Post)]),
        "mplx_comm_unique_id": (C.c_int, [vp]),
        "mplx_comm_init": (C.c_int, [vp, vp, i32, i32]),
        "mplx_comm_destroy": (C.c_int, [vp]),
        "mplx_comm_broadcast_map": (C.c_int, [vp, i32]),
        "mplx_comm_allgather_lists": (C.c_int, [vp, C.POINTER(PackedLists), i64, C.POINTER(PackedLists), vp, vp]),
        "mplx_comm_schedule": (i64, [i32, i32, vp, i32, C.POINTER(CommOp), i64, vp, vp]),
        "mplx_device_alloc": (C.c_int, [vp, C.c_size_t, C.POINTER(vp)]),
        "mplx_device_free": (C.c_int, [vp, vp]),
        "mplx_memcpy_h2d": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "mplx_memcpy_d2h": (C.c_int, [vp, vp, vp, C.c_size_t]),
        "mplx_memset": (C.c_int, [vp, vp, C.c_int, C.c_size_t]),
        "mplx_synchronize": (C.c_int, [vp]),
        "mplx_timer_begin": (C.c_int, [vp]),
        "mplx_timer_end": (C.c_int, [vp, C.POINTER(C.c_float)]),
        "mplx_planner_create": (C.c_int, [C.c_int, C.POINTER(vp)]),
        "mplx_planner_destroy": (None, [vp]),
        "mplx_planner_attach_ctx": (C.c_int, [vp, vp]),
        "mplx_planner_set_provider": (C.c_int, [vp, vp, vp, vp]),
        "mplx_planner_set_map": (C.c_int, [vp, vp, vp, vp, dbl]),
        "mplx_planner_edit_map": (C.c_int, [vp, vp, vp, i64]),
        "mplx_planner_set_controls": (C.c_int, [vp, vp, i32, i32]),
        "mplx_planner_configure": (C.c_int, [vp, C.POINTER(PlannerConfig)]),
        "mplx_planner_plan": (C.c_int, [vp, vp, vp, C.POINTER(PlanSummary)]),
        "mplx_planner_trajectory": (C.c_int, [vp, vp, vp, i32]),
        "mplx_planner_trajectory_end": (C.c_int, [vp, vp]),
        "mplx_planner_closed_set": (C.c_int, [vp, vp, i32, C.POINTER(i32)]),
        "mplx_planner_open_set": (C.c_int, [vp, vp, i32, C.POINTER(i32)]),
        "mplx_planner_last_error": (C.c_char_p, [vp]),
        "mplx_planner_timing": (C.c_int, [vp, C.POINTER(PlanTiming)]),
        "mplx_planner_use_device_heuristic": (C.c_int, [vp, C.c_int]),
        "mplx_planner_set_prior_trajectory": (C.c_int, [vp, vp]),
        "mplx_planner_set_prior_trajectory_potential": (C.c_int, [vp, vp, vp, dbl, dbl]),
        "mplx_planner_set_lpastar": (C.c_int, [vp, C.c_int]),
        "mplx_planner_reset": (C.c_int, [vp]),
        "mplx_planner_linked_nodes": (C.c_int, [vp, vp, i64, C.POINTER(i64), C.POINTER(i64), C.POINTER(i64)]),
        "mplx_planner_update_blocked_nodes": (C.c_int, [vp, vp, i64]),
        "mplx_planner_update_cleared_nodes": (C.c_int, [vp, vp, i64]),
        "mplx_planner_sub_state_space": (C.c_int, [vp, i32]),
        "mplx_planner_set_edge_provider": (C.c_int, [vp, vp, vp]),
        "mplx_selftest_math": (C.c_int, [vp, C.c_int, vp, vp, vp, i64]),
        "mplx_selftest_forward_state": (C.c_int, [i32, i32, vp, vp, C.c_double, vp]),
        "mplx_set_lists_route": (C.c_int, [vp, C.c_int]),
        "mplx_last_lists_route": (C.c_int, [vp]),
        "mplx_last_grid_kernel": (C.c_int, [vp]),
        "mplx_last_identity_form": (C.c_int, [vp]),
        "mplx_debug_store_model": (C.c_int, [vp, C.POINTER(SuccLists), i64]),
        "mplx_yaw_pin_stats": (C.c_int, [vp, C.POINTER(i64), C.POINTER(i64)]),
        "mplx_service": (C.c_int, [vp, C.c_int, C.POINTER(i64)]),
        "mplx_device_info": (C.c_int, [vp, C.c_char_p, C.c_size_t, C.POINTER(i32)]),
    }
    for name in SYMBOLS:
        fn = getattr(L, name)  # AttributeError if the library does not export it
        fn.restype, fn.argtypes = sig[name]
    _lib = L
    return L


def check(ctx, rc):
    if rc != OK:
        msg = lib().mplx_last_error(ctx)
        raise MplxError(rc, msg.decode() if msg else "?")
