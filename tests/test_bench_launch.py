"""CPU: `python bench.py --gpus N` launched plainly (no torch.distributed.run around it) reaches its own launcher
logic; on a box with fewer GPUs than ranks it stops there with a clear message instead of a usage error."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_plain_invocation_with_gpus_2_prepares_the_launch_and_reports_missing_gpus():
    import torch
    if torch.cuda.device_count() >= 2:
        import pytest
        pytest.skip("2 GPUs present: covered by the GPU tests")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.pop("MPLX_BENCH_BACKEND", None)
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                          cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert proc.returncode != 0
    msg = proc.stderr + proc.stdout
    assert "needs 2 GPUs, %d visible" % torch.cuda.device_count() in msg
    assert "torch.distributed.run --nnodes=1 --nproc-per-node 2" in msg  # the launch it had prepared
