"""bench.py's launch contract on a box without a GPU: an N > 1 request must never produce an N = 1 line."""
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def run(args, env_extra=None):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], capture_output=True, text=True, env=env, timeout=300)


def result_lines(out):
    lines = []
    for ln in out.splitlines():
        try:
            rec = json.loads(ln)
        except ValueError:
            continue
        if isinstance(rec, dict) and "metric" in rec:
            lines.append(rec)
    return lines


def test_gpus_2_without_launcher_spawns_ranks_and_fails_loudly_without_gpus():
    """`python bench.py --gpus 2` (how the driver's N = 1 command line is written, BENCH_r01.json.cmd) starts two ranks
    itself; here they have no GPU, so the run must end non-zero and print no result line (round 1 printed n_gpus: 1)."""
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("CPU-side contract test")
    r = run(["--gpus", "2", "--steps", "1", "--warmup", "0", "--nx", "16", "--no-roofline", "--no-cpu-baseline"])
    assert r.returncode != 0
    assert result_lines(r.stdout) == []
    assert "needs a GPU" in r.stderr or "failed" in r.stderr


def test_world_size_mismatch_is_refused():
    import torch

    if torch.cuda.is_available():
        import pytest

        pytest.skip("CPU-side contract test")
    r = run(["--gpus", "4"], {"RANK": "0", "WORLD_SIZE": "1", "LOCAL_RANK": "0"})
    assert r.returncode != 0 and result_lines(r.stdout) == []
    assert "refusing" in r.stderr
