"""The prover's stream design (main + up to 8 side streams per ctx, 27 with three provers) on 2, 4 and 8 hardware
queues: GPU_MAX_HW_QUEUES only counts before a process's first HIP call, so every setting runs in a fresh child
(tools/queue_stress.py), 3 concurrent provers x 300 proofs each, and a hang is a failure (120 s limit per child).
Plus: torch initialising HIP before zg_init (the runtime's default queue count), and the h multiexp in 3 / 4 parts."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(env_extra, *flags, timeout=120):
    env = dict(os.environ)
    env.pop("GPU_MAX_HW_QUEUES", None)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "queue_stress.py")] + list(flags), env=env, cwd=ROOT,
                       capture_output=True, text=True, timeout=timeout)   # TimeoutExpired = a hang = a failure
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-2000:])
    line = json.loads(p.stdout.strip().splitlines()[-1])
    assert line["mismatches"] == 0 and not line["errors"], line
    return line


@pytest.mark.parametrize("queues", [2, 4, 8])
def test_three_provers_on_few_hardware_queues(ctx, queues):
    line = _run({"GPU_MAX_HW_QUEUES": str(queues)}, "--provers", "3", "--proofs", "300", "--log-m", "16")
    assert line["GPU_MAX_HW_QUEUES"] == str(queues)


def test_torch_initialises_hip_first(ctx):
    _run({}, "--provers", "3", "--proofs", "300", "--log-m", "16", "--torch-first")


@pytest.mark.parametrize("parts", [3, 4])
def test_h_multiexp_in_more_parts(ctx, parts):
    """ZG_OPT_H_PARTS = 3 / 4 (main + 4 witness + 3-4 h streams) on the default queue count; one prover, 2^20."""
    _run({}, "--provers", "1", "--proofs", "40", "--log-m", "20", "--h-parts", str(parts))
