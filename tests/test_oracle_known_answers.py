"""CPU: hand-derivable known answers for the oracle.
The plan-level pins of README.md:199-202 are in test_plan_known_answer.py."""
import numpy as np

from oracle import oracle as O


def _empty(dim, edge):
    return np.zeros(edge ** dim, np.int8), [edge] * dim, [0.0] * dim


def test_3d_acc_micro_case():
    U = np.array([[1, -1, 0.5], [2, 0, 0], [1.5, -0.5, 0], [1, -0.5, 0]], dtype=float)
    cells, md, org = _empty(3, 40)
    env = O.Env(3, O.ACC, U, cells, md, org, 0.1, v_max=2)
    nodes = O.make_nodes(3, [[1, 2, 2]], vel=[[0.5, 0, -0.5]], t=[3])
    assert O.lattice_hash(3, O.ACC, nodes[:, 0]) == 0x29b5530c22821406
    r = O.expand(env, nodes)
    assert r["status"].tolist() == [1, 3, 1, 1]  # u=(2,0,0): |v| = 2.5 > 2; 2.0 is not > 2.0
    assert r["cost"][[0, 2, 3]].tolist() == [12.25, 12.5, 11.25]
    assert [int(h) for h in r["hash"][[0, 2, 3]]] == [0x29b55209a71350a4, 0x29b5520a2bcc5303, 0x29b55209a7654187]
    assert r["iters"].tolist() == [16, 0, 20, 16]  # n = 15 runs 16 times, n = 20 exactly 20
    assert r["state"][:, 0].tolist() == [2.0, 1.5, 1.75, 1.5, -1.0, 0.0, 1.0, -1.0, 0.5, 0, 0, 0, 0, 4.0]


def test_3d_jrk_interior_velocity_root():
    U = np.array([[-2, 0, 0], [2, 0, 0], [-1, 1, -1]], dtype=float)
    cells, md, org = _empty(3, 40)
    env = O.Env(3, O.JRK, U, cells, md, org, 0.1, v_max=1, a_max=2)
    r = O.expand(env, O.make_nodes(3, [[2, 2, 2]], vel=[[-0.9, 0, 0]], acc=[[1, 0, 0.5]]))
    assert r["status"].tolist() == [1, 3, 1]
    assert r["cost"][[0, 2]].tolist() == [14.0, 13.0]
    assert [int(h) for h in r["hash"][[0, 2]]] == [0xd8416834e706b2ff, 0xd86aea900f79281b]
    assert r["iters"].tolist() == [9, 0, 9]
    assert r["state"][:3, 0].tolist() == [1.2666666666666666, 2.0, 2.25]


def test_2d_yaw_micro_case_with_exact_tie():
    U = np.array([[0, 0, 0], [0, 0.5, 0], [0, 1, 0], [0, 1, 0.5], [0.5, 0, -0.5]], dtype=float)
    cells, md, org = _empty(2, 100)
    env = O.Env(2, O.ACCxYAW, U, cells, md, org, 0.1, v_max=2, yaw_max=0.5)
    r = O.expand(env, O.make_nodes(2, [[5, 5]], vel=[[1, 0]], yaw=[0]))
    assert r["status"].tolist() == [1, 1, 3, 1, 1]  # (0,1,0) leaves the field of view; last one is an exact tie
    assert r["cost"][[0, 1, 3, 4]].tolist() == [10.0, 10.293004091356016, 11.023518376463223, 10.295318288213227]
    assert [int(h) for h in r["hash"][[0, 1, 3, 4]]] == [0x00a3b72e604e08d0, 0x00a3b72e6092531f, 0x00a3b72e609787e8,
                                                         0x00a3b72e63f001a3]
    assert r["iters"].tolist() == [11, 11, 0, 11, 16]


def test_2d_potential_cells_at_voxel_boundaries():
    pot = np.zeros((100, 100), np.int8)
    pot[:, 52:55] = 40
    U = np.array([[0, 0], [0.5, 0], [0, 0.5]], dtype=float)
    env = O.Env(2, O.ACC, U, np.zeros(100 * 100, np.int8), [100, 100], [0, 0], 0.1, v_max=2, potential=pot,
                potential_weight=0.5)
    nodes = O.make_nodes(2, [[5, 5]], vel=[[0.5, 0]])
    r = O.expand(env, nodes)
    assert r["cost"].tolist() == [22.0, 18.25, 22.25] and r["iters"].tolist() == [5, 11, 5]
    pot = np.zeros((100, 100), np.int8)
    pot[:, 55] = 40
    pot[:, 56] = 100
    U = np.array([[0, 0], [0.5, 0], [1, 0]], dtype=float)
    env = O.Env(2, O.ACC, U, np.zeros(100 * 100, np.int8), [100, 100], [0, 0], 0.1, v_max=2, potential=pot,
                potential_weight=0.5)
    r = O.expand(env, nodes)
    assert r["status"].tolist() == [1, 2, 2] and r["cost"][0] == 10.0 and r["iters"].tolist() == [5, 10, 12]


def test_corridor_start_first_expansion_hashes():
    U = np.array([[x, y] for x in (-0.5, 0, 0.5) for y in (-0.5, 0, 0.5)], dtype=float)
    env = O.Env(2, O.ACC, U, np.zeros(799 * 199, np.int8), [799, 199], [0, -5], 0.05, v_max=1, a_max=1)
    nodes = O.make_nodes(2, [[2.5, -3.5]])
    assert O.lattice_hash(2, O.ACC, nodes[:, 0]) == 0x00028253a2ec9d98
    r = O.expand(env, nodes)
    assert r["status"].tolist() == [1, 1, 1, 1, 0, 1, 1, 1, 1]
    assert r["cost"][[0, 1]].tolist() == [10.5, 10.25]
    assert [int(h) for h in r["hash"]] == [0x00028253dc64246a, 0x00028253dc6422e8, 0x00028253dc641beb,
                                           0x00028253a2ec94b1, 0x00028253a2ec9d98, 0x00028253a2ec9c5e,
                                           0x00028253a3f54b0b, 0x00028253a3f46151, 0x00028253a3f47b3b]


def test_quantisation_rounds_half_away_from_zero():
    wp = np.zeros(10)
    wp[0], wp[1], wp[2], wp[3] = -3.75, 0.0, 0.25, 0.75
    seed = 0
    for v in (-375, 3, 0, 8):  # pos_x, vel_x, pos_y, vel_y (waypoint.h:95-112 interleaves per axis)
        seed ^= ((v + 2 ** 64) % 2 ** 64 + 0x9e3779b9 + ((seed << 6) % 2 ** 64) + (seed >> 2)) % 2 ** 64
    assert O.lattice_hash(2, O.ACC, wp) == seed


def test_sample_loop_runs_n_or_n_plus_one_times():
    plus_one = [n for n in range(5, 40) if O.loop_count(1.0, n) == n + 1]
    assert plus_one == [6, 7, 10, 13, 14, 15, 19, 22, 23, 24, 26, 27, 28, 29, 30, 31, 33, 37, 38]
    assert all(O.loop_count(1.0, n) in (n, n + 1) for n in range(5, 201))
    assert sum(O.loop_count(1.0, n) == n + 1 for n in range(5, 201)) == 100


def test_default_heuristic():
    wp = np.zeros(10)
    goal = np.zeros(10)
    goal[0], goal[1] = 3.0, -4.5
    assert O.heur(2, O.ACC, 10.0, 1.0, wp, goal) == 45.0
    assert O.heur(2, O.ACC, 10.0, 2.0, wp, goal) == 22.5
    assert O.heur(2, O.ACC, 10.0, -1.0, wp, goal) == 45.0
    assert O.heur(2, O.ACC, 10.0, 1.0, goal, goal) == 0.0


def test_three_vector_norm_summation_order_is_within_two_ulps():
    """env_map.h:116's vel.norm() on a 3-vector is the one expression of the path whose value depends on Eigen's redux
    tree, which the reference does not pin (oracle/stub_include/Eigen/mini_dense.h): index order
    (x^2 + y^2) + z^2 -- Eigen 3.3 vectorised, the oracle, the stand-in headers and the kernels -- against the scalar
    unrolled x^2 + (y^2 + z^2).  The two norms differ by at most two units in the last place; the potential-map COST that
    multiplies them by gradient_weight * dt is therefore identical to ~5e-16 relative, nine orders inside north_star's 1e-6."""
    rng = np.random.default_rng(5)
    v = np.round(rng.uniform(-3, 3, size=(200000, 3)) / 0.125) * 0.125 + rng.uniform(-1e-3, 1e-3, size=(200000, 3))
    a = np.sqrt((v[:, 0] * v[:, 0] + v[:, 1] * v[:, 1]) + v[:, 2] * v[:, 2])
    b = np.sqrt(v[:, 0] * v[:, 0] + (v[:, 1] * v[:, 1] + v[:, 2] * v[:, 2]))
    differ = a != b
    assert 0 < differ.mean() < 0.5          # the order does matter on some inputs ...
    assert np.all(np.abs(a - b) <= 2 * np.spacing(np.maximum(a, b)))   # ... by two units in the last place at most
    assert np.all(np.abs(a - b) <= 4.5e-16 * b)
