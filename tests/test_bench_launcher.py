"""`python bench.py --gpus N` must start its own N ranks (the driver's command form), aggregate rank 0's JSON line and
fail loudly when a rank dies.  CPU: --dry-run keeps the launcher, rendezvous (gloo, 127.0.0.1) and MAX-over-ranks
reduction and skips the GPU work."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run(*args, env=None):
    e = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH, *args], env=e, capture_output=True, text=True, timeout=240)


def test_gpus2_self_launch_prints_one_json_line():
    r = _run("--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["ranks"] == [[0, 0], [1, 1]]          # one process per GPU: LOCAL_RANK r for rank r
    assert out["elapsed_max_s"] >= 0.02              # MAX over ranks (rank 1 sleeps longer)


def test_failed_rank_fails_the_bench():
    r = _run("--gpus", "2", "--dry-run", env={"VTD_BENCH_FAIL_RANK": "1"})
    assert r.returncode != 0
    assert "ranks failed" in r.stderr


def test_world_size_mismatch_is_an_error():
    r = _run("--gpus", "2", "--dry-run", env={"WORLD_SIZE": "3", "RANK": "0"})
    assert r.returncode != 0 and "does not match" in r.stderr
