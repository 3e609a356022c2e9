"""CPU: `python bench.py --gpus N` outside torch.distributed.run starts the N rank processes itself (the driver invokes it that
way) — the parent only spawns, waits and relays rank 0's JSON line; a failing rank makes the parent fail."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=e, capture_output=True, text=True, timeout=300)


def test_self_launch_two_ranks_gloo():
    p = _run(["--gpus", "2", "--selftest-launcher"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["selftest"] == "launcher" and d["n_gpus"] == 2 and d["gather_ok"] is True


def test_self_launch_reports_failing_rank():
    # MASTER settings come from the launcher itself; a rank that cannot run (bad flag) must make the parent exit non-zero
    p = _run(["--gpus", "2", "--selftest-launcher", "--config", "notanumber"])
    assert p.returncode != 0
