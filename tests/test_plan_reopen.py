"""CPU: the batched host search when a closed node is re-opened.

The reference's A* pushes a node again when a cheaper path reaches it after it
was closed (graph_search.h:108-141), and then expands it again.  That happens
with an inconsistent heuristic: the planner's own default v_max <= 0 (unbounded
velocity, h = w * distance, env_base.h:58-64), with or without eps > 1
(PlannerBase::setEpsilon) -- on the corridor map 895 expansions for 836 closed
nodes.  The
successor cache of the batched search must hand a node's lists out once and
forget them, so that the second expansion launches again: batch = 1 (the
reference's loop) and batch = 8 / 64 must give the same search, and the same
search as the reference's own MapPlanner where that is built.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
from test_plan_known_answer import corridor, provider_from_oracle


def run(m, batch, eps, v_max_env, v_max_heur, w=10.0, max_num=-1):
    c = corridor()
    U = m.workloads.grid_controls([-0.5, 0.0, 0.5], 2)
    oenv = O.Env(2, O.ACC, U, c["cells"], c["dim"], c["origin"], c["res"], v_max=v_max_env, a_max=1.0, dt=1.0, w=w)
    prov, keep = provider_from_oracle(oenv)
    pl = m.MapPlanner(2, provider=prov)
    mu = m.MapUtil(2)
    mu.setMap(c["origin"], c["dim"], c["cells"], c["res"])
    pl.setMapUtil(mu)
    pl.setVmax(v_max_heur)
    pl.setDt(1.0)
    pl.setW(w)
    pl.setU(U)
    pl.setEpsilon(eps)
    pl.setBatch(batch)
    pl.setMaxNum(max_num)
    ok = pl.plan(m.Waypoint(2, m.ACC, pos=c["start"]), m.Waypoint(2, m.ACC, pos=c["goal"]))
    s = pl.summary()
    tr = pl.getTraj()
    pl.close()
    del keep
    return ok, s, tr, oenv


@pytest.mark.parametrize("eps,v_env,v_heur", [(1.0, -1.0, -1.0), (2.0, -1.0, -1.0), (1.0, -1.0, 2.0)])
def test_reopened_nodes_batch_sizes_agree(engine, eps, v_env, v_heur):
    ok1, s1, t1, _ = run(engine, 1, eps, v_env, v_heur)
    assert ok1
    # the premise: this search re-expands closed nodes (more expansions than closed nodes)
    assert s1["expansions"] > s1["closed"], s1
    for batch in (8, 64):
        ok, s, t, _ = run(engine, batch, eps, v_env, v_heur)
        assert ok
        for k in ("closed", "expansions", "cost", "total_time", "segments", "nodes", "opened"):
            assert s[k] == s1[k], (batch, k, s[k], s1[k])
        assert np.array_equal(t.actions, t1.actions) and np.array_equal(t.nodes, t1.nodes)
        assert s["device_launches"] < s1["device_launches"]


@pytest.mark.skipif(not os.path.exists(O.REF_PLANNER_SO), reason="oracle/_ref/libmpl_ref_planner.so not built")
@pytest.mark.parametrize("eps", [1.0, 2.0])
def test_reopened_nodes_match_the_reference_planner(engine, eps):
    ok, s, t, oenv = run(engine, 64, eps, -1.0, -1.0)
    assert s["expansions"] > s["closed"]
    c = corridor()
    start = engine.Waypoint(2, engine.ACC, pos=c["start"]).to_row()
    goal = engine.Waypoint(2, engine.ACC, pos=c["goal"]).to_row()
    r = O.ref_plan(oenv, start, goal, use_gpu=False, epsilon=eps)
    assert ok and r["ok"]
    assert s["expansions"] == r["expansions"] and s["closed"] == r["closed"] and s["opened"] == r["opened"]
    assert s["cost"] == r["cost"] and s["total_time"] == r["total_time"] and s["segments"] == r["segments"]
