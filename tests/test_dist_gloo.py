"""CPU, world_size 2, gloo: the multi-GPU path of bench.py — work-item assignment and the single log gather."""
import os

import numpy as np
import torch.multiprocessing as mp


def _worker(rank, world, port, q):
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import torch.distributed as dist
    import __graft_entry__ as graft
    sharding = __import__("importlib").import_module("icp_proposal_amd.sharding") if False else None
    pkg_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "icp-proposal_amd")
    import importlib.util
    spec = importlib.util.spec_from_file_location("icp_sharding", os.path.join(pkg_dir, "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    items = sharding.assign_work_items(5, world)[rank]
    rec = np.zeros((7, 4 + 10 + 3))
    rec[:, 0] = np.arange(7)
    rec[:, 2] = rank
    rec[:, 3] = -100.0 + rank * 10 + np.arange(7)  # rank 1, step 6 is the best sample
    allrec = sharding.gather_records(rec, dist)
    rk, st, best = sharding.best_sample(allrec)
    # ragged gather: rank 0 holds 3 blocks, rank 1 holds 2 (5 work items over 2 ranks)
    blocks = [np.full((4, 6), 100.0 * rank + i) for i in range(len(items))]
    per_rank = sharding.gather_ragged(blocks, dist)
    ragged_ok = [b.shape for b in per_rank] == [(3, 4, 6), (2, 4, 6)] and per_rank[1][1][0, 0] == 101.0 and per_rank[0][2][3, 5] == 2.0
    q.put((rank, items, allrec.shape, float(allrec[1, 3, 2]), rk, st, ragged_ok))
    dist.barrier()
    dist.destroy_process_group()


def test_gather_and_assignment_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 200
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    for r in res:
        assert tuple(r[2]) == (2, 7, 17) and r[3] == 1.0 and (r[4], r[5]) == (1, 6)
        assert r[6], "ragged gather returned wrong blocks"
