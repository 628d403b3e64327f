"""bench.py's contract with the driver, on a small instance: ONE JSON line with the agreed keys, the roofline and
cpu_baseline objects; and the N > 1 code path (two ranks sharing this GPU over gloo) printing the same line from rank 0."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--log-m", "16", "--steps", "3", "--warmup", "1", "--no-msm", "--no-dense", "--no-sha", "--shard-model", "0", "--provers", "1"]


def _line(out):
    lines = [l for l in out.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out[-2000:]
    return json.loads(lines[0])


def _check_contract(d, n_gpus):
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == n_gpus and d["steps"] == 3 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "constraints/s" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["config"]["n_constraints"] / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    r = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in r, k
    assert r["bound"] == "hbm" and r["unit"] == "GB/s" and r["peak"] == 8000.0
    if r["achieved"] is not None:
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-9
    assert r["traffic"] is None        # PMC figures are only ever attached to the workload they were measured on


def test_bench_line_contract_single_gpu():
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL + ["--cpu-log-m", "16"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-2000:]
    d = _line(p.stdout)
    _check_contract(d, 1)
    c = d["cpu_baseline"]
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in c, k
    assert c["kind"] == "port" and c["cores"] >= 1 and "proof bytes equal to the GPU's: True" in c["sample"]
    assert 20 < d["launches_per_proof"] < 400 and "b_g1_behind_b_g2" in d["digit_sort_sharing"]
    # the reference's own per-proof flow (read the key checked, prove once): key_load, one_shot, outer_prove
    kl, osh, op = d["key_load"], d["one_shot"], d["outer_prove"]
    assert kl["bytes"] > 0 and 0 < kl["checked_s"] < 5 and 0 < kl["unchecked_s"] < 5
    assert osh["ms"] > 0 and op.get("proof_verifies") is True and op["total_s"] > 0 and op["pk_read_s"] > 0, op
    assert d["per_key_preparation"]["prepare_s"] >= 0
    hm = d["h_msm"]   # the h multiexp's phases from HIP events
    assert hm["sort_ms"] > 0 and hm["accumulate_ms"] > 0 and hm["reduce_ms"] > 0 and hm["accumulate_launches_per_proof"] >= 1


def test_bench_two_ranks_on_one_gpu_over_gloo():
    env = dict(os.environ, ZA_BENCH_DEVICE="0", MASTER_ADDR="127.0.0.1")
    p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo",
                        "--no-cpu-baseline"] + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-2000:])
    d = _line(p.stdout)
    _check_contract(d, 2)
    assert d["scaling"] == "strong" and "sharded_phases" in d and d["sharded_phases"]["msm_phase_ms"] > 0


def test_bench_gpus_2_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher on the command line and no WORLD_SIZE in the environment: bench.py starts
    the ranks itself (as child processes, before it touches the GPU) and leaves with their exit code (missing 2)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["ZA_BENCH_DEVICE"] = "0"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dist-backend", "gloo", "--no-cpu-baseline"]
                       + SMALL, capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert p.returncode == 0, (p.stdout[-1000:], p.stderr[-2000:])
    d = _line(p.stdout)
    _check_contract(d, 2)
    assert d["scaling"] == "strong" and d["sharded_phases"]["msm_phase_ms"] > 0

