"""RCCL's first contact with the sharded proof flow (`backend="nccl"` had never been initialised on
the GPU box).  No multi-GPU box exists for the tests, so the collectives are forced at world size 1 in a fresh child
process (RCCL must be initialised before anything else in the process touches the GPU)."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_sharded_flow_with_forced_collectives_over_rccl_at_world_1(lib_built):
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", RANK="0", WORLD_SIZE="1")
    r = subprocess.run([sys.executable, os.path.join(HERE, "rccl_world1_child.py")], env=env, capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_OK")]
    assert len(ok) == 2 and "h_eval=1" in ok[0] and "h_eval=0" in ok[1], r.stdout
    print("\n".join(ok + [ln for ln in r.stdout.splitlines() if ln.startswith("RCCL_VERSION")]))
