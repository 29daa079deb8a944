"""The ONE JSON line bench.py prints: from a long-form result (a committed bench output of round 5) the compact line
must keep the contract's keys, `roofline`, `cpu_baseline`, the per-configuration numbers, and stay under 8 KB -- a record
that keeps only the tail of stdout must keep all of it (round-5 review: the 30 KB line was cut)."""
import importlib.util
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _long_form():
    path = os.path.join(ROOT, "profiles", "r05_bench_c4_final.json")
    return json.loads([l for l in open(path) if l.startswith("{")][-1])


def test_compact_line_is_small_and_complete():
    b = _bench()
    long_form = _long_form()
    assert len(json.dumps(long_form)) > 16000  # the problem being solved (the driver keeps 8 KB)
    line = b.compact_line(long_form, "bench_detail.json")
    text = json.dumps(line, separators=(",", ":"))
    assert len(text) < 8000, len(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline", "detail"):
        assert k in line, k
    assert line["config"]["workload"].startswith("C4") and "model" not in line["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in line["roofline"], k
    assert abs(line["roofline"]["frac"] - long_form["roofline"]["frac"]) < 1e-5
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in line["cpu_baseline"], k
    for name in ("C2", "C3", "C5"):
        e = line["other_configs"][name]
        assert e["kernel_ms"] > 0 and "cpu_baseline" in e and "bound" in e and "issue_frac" in e
    assert [c["gpus"] for c in line["strong_scaling_compute_bound"]] == [1, 2, 4, 8]
    assert set(line["plan"]["3D_160"]["engine_host_search"]["timing_split"]) == {"provider_ms", "fill_ms", "pick_ms", "relax_ms", "recover_ms"}
    # numbers only: no prose legs on the line
    def walk(d, path=""):
        for k, v in d.items():
            assert k not in ("what", "timing", "speedup_note"), path + k
            if isinstance(v, dict):
                walk(v, path + k + ".")
    walk(line)


def test_compact_line_survives_failed_legs():
    b = _bench()
    long_form = _long_form()
    long_form["wavefront"] = {"error": "RuntimeError: boom"}
    long_form["other_configs"] = {"error": "x"}
    long_form["plan"]["3D"] = {"error": "y"}
    line = b.compact_line(long_form, "d.json")
    assert line["wavefront"] == {"error": "RuntimeError: boom"} and line["plan"]["3D"] == {"error": "y"}


def test_cpu_baseline_protocol_on_a_small_workload():
    """bench.py::cpu_baseline end to end on the CPU (the reference build or the restatement, whichever is present): the
    repetition is sized by measurement, one warm-up + `reps` timed repetitions, the median and its spread reported, for all
    host threads and for one."""
    import sys
    sys.path.insert(0, ROOT)
    import motion_primitive_library_amd as m
    b = _bench()
    wl = m.workloads.make("C2", scale=0.125, n_nodes=96)
    cb, oenv, n = b.cpu_baseline(wl, rep_seconds=0.08, rep_seconds_1thread=0.05, reps=5)
    assert cb["unit"] == "pairs/s" and cb["value"] > 0 and cb["value_1thread"] > 0
    assert cb["kind"] in ("reference", "port") and "median" in cb["sample"] and cb["cores"] >= 1
    for leg in ("all_cores", "one_thread"):
        p = cb["protocol"][leg]
        assert p["reps"] == 5 and p["min"] <= p["value"] <= p["max"]
        assert p["rep_seconds"] > 0.004, p                         # sized up from the 96-node frontier ...
        assert p["frontier_passes_per_rep"] > 1 and p["frontier_nodes_per_pass"] == 96   # ... by walking it many times
        assert p["nodes_per_rep"] == p["frontier_passes_per_rep"] * 96
    assert n == cb["protocol"]["all_cores"]["nodes_per_rep"] and oenv is not None
