"""CPU: the engine's host search (csrc/host_planner.hpp: A*, prior trajectory; csrc/host_lpastar.hpp: LPA* with map edits and
re-rooting) compiled with AddressSanitizer + UBSan + _GLIBCXX_ASSERTIONS and run with the CPU oracle as successor provider
(tests/sanitize/host_search_harness.cpp).  GPU sanitizers are not available on the pool; the host search is where the
pointer-heavy code lives -- the round-5 advisor found its LPA* corrupting memory after updateBlockedNodes + getSubStateSpace
+ plan with exactly such a harness.  The scenario is part of the harness now."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(shutil.which("g++") is None, reason="g++ not found")
def test_host_search_is_clean_under_asan_ubsan(tmp_path):
    exe = str(tmp_path / "host_search_harness")
    cmd = ["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D_GLIBCXX_ASSERTIONS",
           "-ffp-contract=off", "-o", exe, os.path.join(ROOT, "tests", "sanitize", "host_search_harness.cpp"),
           os.path.join(ROOT, "oracle", "mpl_oracle.cpp"), "-pthread"]
    build = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert build.returncode == 0, build.stderr[-3000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    run = subprocess.run([exe], capture_output=True, text=True, timeout=600, env=env)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    assert "harness: ok" in run.stdout
    assert "ERROR: AddressSanitizer" not in run.stderr and "runtime error" not in run.stderr
