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


def check_multi_gpu_block(mg, world):
    """what the first 8-GPU run will be checked with: did the communicator see `world` ranks, which device did each hold, how long
    did its chains and its gather take"""
    assert mg["rccl_ranks"] == world and mg["backend"] == "gloo"
    assert [e["rank"] for e in mg["ranks"]] == list(range(world))
    assert len({e["gpu"] for e in mg["ranks"]}) == world
    for e in mg["ranks"]:
        assert {"rank", "local_rank", "device", "gpu", "name", "chain_ms", "log_gather_ms"} <= set(e)


def test_self_launch_two_ranks_gloo():
    p = _run(["--gpus", "2", "--selftest-launcher"])
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["selftest"] == "launcher" and d["n_gpus"] == 2 and d["gather_ok"] is True
    check_multi_gpu_block(d["multi_gpu"], 2)


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
    check_multi_gpu_block(d["multi_gpu"], 2)


def test_two_ranks_on_one_device_fail_the_run():
    """The line's `multi_gpu` block comes from the communicator; a job whose ranks share a physical device must not produce a line
    at all (bench.rank_report).  Exercised on the function itself with a fake communicator."""
    sys.path.insert(0, ROOT)
    import bench

    class FakeDist:
        def get_world_size(self): return 2
        def get_backend(self): return "fake"
        def all_gather_object(self, out, mine):
            out[0] = dict(mine, rank=0)
            out[1] = dict(mine, rank=1)  # the same device identity twice
    import pytest
    with pytest.raises(SystemExit) as ei:
        bench.rank_report(FakeDist(), None, 0, 2, 0, 1.0, 0.1, cpu=True)
    assert "same device" in str(ei.value)
