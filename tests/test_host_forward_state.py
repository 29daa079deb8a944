"""CPU: the host search's evaluation of successor states (csrc/host_planner.hpp::forward_state, exported for tests as
mplx_selftest_forward_state) against the oracle -- and, where built, against the reference's own headers -- bit for
bit, for every control flag, odd durations and non-lattice states.  The engine's search builds the 112-byte state of
every new node with it instead of moving the states across PCIe (the device's states are compared with it in
tests/test_gpu_plan.py::test_host_evaluated_states_equal_the_devices)."""
import ctypes as C
import os

import numpy as np
import pytest

import motion_primitive_library_amd as m
from motion_primitive_library_amd import _abi
from oracle import oracle as O


@pytest.mark.parametrize("dim", [2, 3])
@pytest.mark.parametrize("control", [0x01, 0x03, 0x07, 0x0F, 0x11, 0x13, 0x17, 0x1F])
@pytest.mark.parametrize("dt", [1.0, 0.5, 0.3, 2.0 / 3])
def test_forward_state_equals_the_oracle_and_the_reference(dim, control, dt):
    rng = np.random.default_rng(dim * 100 + control + int(dt * 1000))
    W = m.workloads
    edge = 40
    grid = np.zeros([edge] * dim, dtype=np.int8)  # free map: every valid primitive is emitted with its state
    vals = [-1.0, -1.0 / 3, 0.0, 0.4, 1.1] if dim == 2 else [-0.9, 0.0, 0.7]
    U = W.grid_controls(vals, dim, yaw_rates=[-0.37, 0.0, 0.52] if control & 0x10 else None)
    n = 24
    F = 4 * dim + 2
    nodes = np.zeros((F, n))
    nodes[:dim] = rng.uniform(1.0, 3.0, size=(dim, n))
    nodes[dim:2 * dim] = rng.uniform(-0.6, 0.6, size=(dim, n))
    nodes[2 * dim:3 * dim] = rng.uniform(-0.5, 0.5, size=(dim, n))
    nodes[3 * dim:4 * dim] = rng.uniform(-0.5, 0.5, size=(dim, n))
    nodes[dim, :3] = -0.0  # signed zeros survive `0.0 + x` differently from x
    nodes[4 * dim] = rng.uniform(-3.1, 3.1, size=n)
    nodes[4 * dim + 1] = rng.uniform(0.0, 5.0, size=n)
    env = O.Env(dim, control, U, grid, [edge] * dim, [0.0] * dim, 0.1, dt=dt)  # no limits: everything valid
    lib = _abi.lib()
    out = np.zeros(F)
    for ref in ([False, True] if os.path.exists(O.REF_SO) else [False]):
        r = O.expand(env, nodes, threads=1, ref=ref)
        st = r["status"].reshape(n, U.shape[0])
        checked = 0
        for k in range(n):
            node = np.ascontiguousarray(nodes[:, k])
            for ci in np.nonzero((st[k] == 1) | (st[k] == 2))[0]:
                u = np.ascontiguousarray(U[ci])
                assert lib.mplx_selftest_forward_state(dim, control, node.ctypes.data, u.ctypes.data, C.c_double(dt),
                                                       out.ctypes.data) == 0
                want = r["state"][:, k * U.shape[0] + ci]
                assert np.array_equal(out.view(np.uint64), np.ascontiguousarray(want).view(np.uint64)), (
                    ref, k, ci, out.tolist(), want.tolist())
                checked += 1
        assert checked > n * U.shape[0] // 2
