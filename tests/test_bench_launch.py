"""`python bench.py --gpus N` with no launcher around it starts its N ranks itself (the reference's runner spawns one process per GPU,
speechain/runner.py:2014) -- checked here on CPU with `--launch-check`: the two child ranks come up with the environment a launcher would
have given them, form a process group (gloo without a GPU), all-reduce one number, rank 0's single JSON line reaches the parent's stdout."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(args, env=None):
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, cwd=ROOT, capture_output=True, text=True, timeout=300,
                          env=env or {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})


def test_bench_starts_its_own_ranks():
    r = _run(["--gpus", "2", "--launch-check"])
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d == {"launch_check": "ok", "world": 2, "backend": "gloo", "sum": 3.0}


def test_a_failing_rank_ends_the_launch_with_an_error():
    # WORLD_SIZE of a launcher that does not match --gpus: every rank refuses, the exit code says so
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1")
    r = _run(["--gpus", "2", "--launch-check"], env=env)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
