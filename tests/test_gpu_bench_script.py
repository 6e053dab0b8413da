"""bench.py end to end on the GPU with a tiny workload: the script the driver runs must keep
printing exactly one JSON line with the contract's fields, in plain mode and with the RCCL
process group forced (1 rank: the only size a 1-GPU box offers)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REQUIRED = {"metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
            "vs_baseline", "dtype", "data", "config", "roofline"}


def _run(extra, env=None):
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "1",
           "--pairs-per-gpu", "24", "--n-orb", "192", "--n-lbd", "40"] + extra
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **(env or {}))
    res = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=e)
    assert res.returncode == 0, res.stderr[-3000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_single_gpu_line():
    d = _run(["--cpu-budget-s", "1"])
    assert REQUIRED <= set(d) and "cpu_baseline" in d
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["roofline"]["bound"] == "hbm" and 0 < d["roofline"]["frac"] < 1
    assert set(d["roofline"]) >= {"bound", "achieved", "peak", "unit", "frac", "traffic"}
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["cores"] >= 1
    assert "workload" in d["config"]


def test_bench_forced_rccl_group_one_rank():
    d = _run(["--no-cpu-baseline", "--force-dist"], env={"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29611"})
    assert REQUIRED <= set(d) and d["value"] > 0
    assert "RCCL gather" in d["config"]["parallelism"]
