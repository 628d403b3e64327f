"""bench.py's launcher side, on the CPU: `python bench.py --gpus N` without a launcher starts its own ranks."""
import importlib.util
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_self_launch_command_is_torch_distributed_run_as_a_child(monkeypatch):
    b = _bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env

        class R:
            returncode = 7
        return R()
    monkeypatch.setattr(subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert b.self_launch(4) == 7                      # the launcher's exit code is bench.py's
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-5:] == [os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "3"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_gpus_2_without_world_size_goes_through_the_launcher_not_exit_2():
    """On a box without a GPU the two ranks start, find no device and fail with THEIR message (exit 1 through the launcher):
    bench.py no longer refuses with 'must be launched with torch.distributed.run' (exit 2)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    import torch
    if torch.cuda.is_available():
        return   # the GPU suite covers the run itself (tests/test_gpu_bench.py)
    assert "must be launched with" not in p.stderr
    assert "needs a GPU" in p.stderr and p.returncode not in (0, 2), (p.returncode, p.stderr[-1500:])
