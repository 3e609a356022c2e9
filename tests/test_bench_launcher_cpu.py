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


def test_config4_plumbing_two_ranks_gloo():
    """`python bench.py --config 4 --gpus 2 --selftest-launcher`: BASELINE.json configs[4]'s plumbing without a GPU — the 10 x 10 work
    items target-major over two self-launched ranks, fabricated records, the ragged gather, reassembly in item order."""
    p = _run(["--gpus", "2", "--selftest-launcher", "--config", "4"])
    assert p.returncode == 0, p.stderr[-2000:]
    d = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][0])
    assert d["selftest"] == "config4" and d["n_gpus"] == 2 and d["gather_ok"] is True
    assert d["items_per_rank"] == [50, 50] and d["targets_per_rank"] == [5, 5]
