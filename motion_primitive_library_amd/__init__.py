"""motion_primitive_library_amd -- MI355X-native successor expansion for
search-based motion-primitive planning (the env_map<Dim>::get_succ path of
sikang/motion_primitive_library), behind the C ABI of include/mplx.h.

The compute path is csrc/libmplx.so (hand-written HIP for gfx950).  Importing
the package does not need a GPU; creating an EnvMap does, and there is no CPU
fallback of any kind.
"""
from . import _abi, shard, workloads
from .env import (ACC, ACCxYAW, JRK, JRKxYAW, SNP, SNPxYAW, VEL, VELxYAW, SLOT_BLOCKED, SLOT_FINITE,
                  SLOT_SKIP_DYN, SLOT_SKIP_SAME, DeviceArray, EnvMap, Lists, PackedLists, Slots, Waypoint, lists_from_dense,
                  pack_host_lists)

from .planner import MapPlanner, MapUtil, Trajectory

__all__ = ["MapPlanner", "MapUtil", "Trajectory", "EnvMap", "Waypoint", "Slots", "Lists", "lists_from_dense", "PackedLists", "pack_host_lists", "DeviceArray", "workloads", "VEL", "ACC", "JRK", "SNP", "VELxYAW",
           "ACCxYAW", "JRKxYAW", "SNPxYAW", "SLOT_SKIP_SAME", "SLOT_FINITE", "SLOT_BLOCKED", "SLOT_SKIP_DYN"]
