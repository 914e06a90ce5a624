"""The multi-rank code path of bench.py, rehearsed on one GPU: one rank under torch.distributed.run with a real RCCL process group
and the multi-rank control flow forced on (BENCH_FORCE_DIST=1) -- process-group start-up, parameter broadcast, hook-launched
bucket all-reduces in the eager steps, hipGraph capture with a process group alive, the all-reduces between the two graph
replays, barrier + MAX over ranks, and the one-JSON-line contract on stdout.

Guards three defects this rehearsal found (none visible to the gloo CPU test): collectives launched from gradient hooks while
backward was being captured ("capturing stream has unjoined work"), RCCL's version banner on stdout, and the RCCL watchdog
thread's event queries being rejected by a global-mode capture."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_multirank_path_on_one_rank():
    env = dict(os.environ, BENCH_FORCE_DIST="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29533", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "5", "--no-cpu-baseline"]
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=420)  # a child process, never an exec
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[:2000]  # exactly one line on stdout: the JSON result
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["value"] > 0 and d["unit"] == "utterances/s"
    assert d["roofline"]["launches_timed"] > 0
