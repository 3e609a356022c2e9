"""Multi-GPU plumbing for independent chains: static assignment of (target, chain) work items to
ranks and the single gather of fixed-size per-step log records at log-write time.

Chains never exchange data while sampling (reference: apps/femur/RunMHRandomInitComparison.scala:66-87 runs them as
independent JVM threads), so there is no data-path collective; `gather_records` is the only communication and uses
whatever backend the process group was created with (RCCL = "nccl" on the GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def assign_work_items(n_items: int, world_size: int):
    """Round-robin: item i -> rank i % world_size.  Returns a list (per rank) of item indices; sizes differ by <= 1.
    (Independent chains on ONE target: RunMHRandomInitComparison-style jobs.  Batch registration uses assign_target_major.)"""
    return [list(range(rk, n_items, world_size)) for rk in range(world_size)]


def assign_target_major(n_targets: int, n_chains: int, world_size: int):
    """(target, chain) work items of a batch registration (item k = target k // n_chains, chain k % n_chains) over the ranks,
    TARGET-MAJOR: loads differ by at most one item, and a rank meets as few targets as that allows — every target a rank meets costs
    it a context (the target's upload, its sphere hierarchy, cold first searches of 0.6-3.4 ms), and chains of one target on one
    rank can share launches.  Round-robin made every rank meet every target (10 targets x 10 chains over 8 ranks: 10 contexts per
    rank instead of 2).

    1. quotas: n_items // world (+1 for the first n_items % world ranks);
    2. whole targets go to the rank with the largest remaining quota while one fits (longest-processing-time order);
    3. a target that fits nowhere whole is cut into pieces that FILL ranks' remaining quotas exactly where a subset of them adds up to
       n_chains (subset sum over <= world numbers), else dealt to the largest remaining quotas in turn.
    Returns a list (per rank) of item indices in target-major order."""
    n_items = n_targets * n_chains
    quota = [n_items // world_size + (1 if rk < n_items % world_size else 0) for rk in range(world_size)]
    out = [[] for _ in range(world_size)]
    split = []
    for t in range(n_targets):
        rk = max(range(world_size), key=lambda k: (quota[k], -k))
        if quota[rk] >= n_chains:
            out[rk] += list(range(t * n_chains, (t + 1) * n_chains))
            quota[rk] -= n_chains
        else:
            split.append(t)
    for t in split:
        chains = list(range(t * n_chains, (t + 1) * n_chains))
        # subset of ranks whose remaining quotas add up to exactly n_chains (prefer few, large pieces)
        order = sorted((k for k in range(world_size) if quota[k] > 0), key=lambda k: (-quota[k], k))
        reach = {0: []}
        for k in order:
            for tot, used in sorted(reach.items(), reverse=True):
                nt = tot + quota[k]
                if nt <= n_chains and nt not in reach:
                    reach[nt] = used + [k]
        ranks = reach.get(n_chains)
        if ranks is None:  # no exact fill: largest remaining quotas in turn
            ranks = order
        for k in ranks:
            take = min(quota[k], len(chains))
            out[k] += chains[:take]
            chains = chains[take:]
            quota[k] -= take
            if not chains:
                break
        assert not chains, "internal: work items left unassigned"
    for o in out:
        o.sort()
    return out


def gather_records(records: np.ndarray, dist=None, device=None):
    """all_gather of a [n_steps, record_len] float64 array; every rank must pass the same shape.
    Returns an array [world, n_steps, record_len] (or records[None] without a process group)."""
    if dist is None or not dist.is_initialized():
        return records[None]
    import torch
    t = torch.from_numpy(np.ascontiguousarray(records))
    if device is not None:
        t = t.to(device)
    out = [torch.empty_like(t) for _ in range(dist.get_world_size())]
    dist.all_gather(out, t)
    return torch.stack(out).cpu().numpy()


def best_sample(all_records: np.ndarray):
    """BestSampleLogger over every chain (api/sampling/SamplingRegistration.scala:56-58,91-92): the record with the
    highest log value.  Returns (rank, step, record)."""
    flat = all_records[..., 3]
    rk, st = np.unravel_index(np.argmax(flat), flat.shape)
    return int(rk), int(st), all_records[rk, st]


def gather_ragged(blocks, dist=None, device=None):
    """One all_gather for ranks that hold DIFFERENT numbers of equally shaped record blocks (100 work items over 8 ranks:
    13/13/13/13/12/12/12/12): every rank pads its stack to the common maximum with NaN blocks, ships its true count in a
    header row, and the padding is dropped after the gather.  Returns a list (per rank) of [n_blocks_r, steps, len] arrays."""
    blocks = [np.asarray(b, dtype=np.float64) for b in blocks]
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    shape = blocks[0].shape if blocks else (0, 0)
    if world == 1:
        return [np.stack(blocks) if blocks else np.zeros((0,) + shape)]
    import torch
    # block shapes agree across ranks by construction; the count is agreed on with one small all_reduce(max)
    n_local = torch.tensor([len(blocks), shape[0], shape[1]], dtype=torch.int64)
    if device is not None:
        n_local = n_local.to(device)
    n_max = n_local.clone()
    dist.all_reduce(n_max, op=dist.ReduceOp.MAX)
    nb, steps, ln = (int(v) for v in n_max.cpu())
    buf = np.full((nb + 1, steps, ln), np.nan)
    buf[0, 0, 0] = len(blocks)
    for i, b in enumerate(blocks):
        buf[1 + i] = b
    gathered = gather_records(buf.reshape(nb + 1, steps * ln), dist, device).reshape(world, nb + 1, steps, ln)
    return [g[1:1 + int(g[0, 0, 0])] for g in gathered]


def run_batch(pkg, model, targets, n_chains: int, n_steps: int, make_setup, dist=None, device_index: int = 0, base_seed: int = 1024,
              chains_per_launch: int = 0, return_stats: bool = False):
    """Batch registration (BASELINE.json configs[4]; reference: the 10-way target pool × per-target chain loop of
    apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:106-163): work items = (target, chain) pairs, dealt target-major
    over the ranks (assign_target_major); a rank keeps ONE context per target it meets; chains never communicate; the per-step
    records of all items are exchanged with a single all_gather at the end.  Returns (items, records): items[k] = (target index,
    chain index) and records[k] = [n_steps, 14 + rank] for every item of the whole job, in item order, on every rank.
    chains_per_launch = B > 1: the rank steps B of its chains OF ONE TARGET side by side through icp_chain_step_batched (one context
    per chain — model and target are shared between them on the device; SURVEY.md §8e "within a GPU, batch B chains per launch") —
    same records, chain by chain.  0 (default): three targets' worth of chains, at most 32 (the one-workgroup factorisations and
    decompositions of B chains then run on B CUs instead of one after the other; chains of DIFFERENT targets share a submission as
    well: the step only needs them to share the model); 1: one context, the chains one by one.
    return_stats: a third value, this rank's {items, contexts_built, targets_met, chain_ms, gather_ms}."""
    import time
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    items = [(t, c) for t in range(len(targets)) for c in range(n_chains)]
    mine = assign_target_major(len(targets), n_chains, world)[rank]
    blocks = []
    contexts_built = 0
    pool = []  # this rank's contexts, kept from target to target
    if chains_per_launch <= 0:
        # (measured on one MI355X, face configuration: 10 chains per submission 6.7k it/s, 20: 9.1k, 30: 10.5k, 40: 10.3k — a round costs
        # its launches and the slowest chain's decomposition, whatever it carries: tools/r4_many.sh)
        # Short chains: round 4 took fewer — a context cost ≈ 14 ms to make the first time (four streams; 10 x 10 x 50 steps: 4,390 it/s
        # with 10 per submission, 4,765 with 20, 3,560 with 30).  Round 5: side streams on first use, a keyed model, the main streams made
        # ahead by a helper thread — a further context is 1.1-2 ms —, and from 24 chains on a submission runs inside the on-device loop:
        # 6,960 it/s with 20 per submission, 7,300-7,540 with 25.
        per_target = 3 if n_steps >= 30 else 1   # (round 5: a further context costs 1.1 ms, not 14: also short chains in submissions of ~25)
        chains_per_launch = max(1, min(32, per_target * n_chains))
        # … in submissions of EQUAL size (100 items: 4 x 25, not 3 x 30 + 10): from 24 chains on a submission of wide-step chains runs
        # as the on-device loop (icp_chains_run_on_device: 1.7 ms per step of 30 face chains against 2.0 host-stepped), a remainder of 10
        # would be stepped by the host at 6.8k it/s
        if len(mine) > chains_per_launch:
            n_sub = -(-len(mine) // chains_per_launch)
            chains_per_launch = -(-len(mine) // n_sub)
    t_start = time.perf_counter()
    phase = dict(contexts=0.0, set_target=0.0, setups=0.0, chains=0.0, steps=0.0, close=0.0)  # where a rank's wall time goes (seconds)
    my_targets = sorted(set(items[k][0] for k in mine))
    theta0 = lambda k: pkg.random_initial_parameters(model, items[k][1], base_seed)
    seed = lambda k: base_seed + 1000 * items[k][0] + items[k][1]
    setups = {}
    def setup_of(t):
        if t not in setups:
            setups[t] = make_setup(model, targets[t])
        return setups[t]
    if chains_per_launch > 1:
        # `chains_per_launch` work items side by side, in item order (target-major): the chains of one target and — the wide step only
        # needs its chains to share the MODEL — of the targets behind it.  The rank's contexts (per-chain scratch, streams, pinned
        # buffers; the model's device data is shared) are made ONCE and handed from target to target (icp_ctx_set_target): a context
        # costs 20+ ms to make and as much to destroy.
        order = sorted(mine)
        # (the streams of the contexts to come are made by a helper thread while the first context does the model's one-off host work)
        pkg.expect_contexts(device_index, min(len(order), chains_per_launch))
        for g0 in range(0, len(order), chains_per_launch):
            group = order[g0:g0 + chains_per_launch]
            tp = time.perf_counter()
            while len(pool) < len(group):
                pool.append(pkg.IcpContext(model, targets[items[group[len(pool)]][0]], device=device_index))
                contexts_built += 1
            ctxs = pool[:len(group)]
            phase["contexts"] += time.perf_counter() - tp; tp = time.perf_counter()
            for cx, k in zip(ctxs, group):
                if cx.target is not targets[items[k][0]]:
                    cx.setTarget(targets[items[k][0]])
            phase["set_target"] += time.perf_counter() - tp; tp = time.perf_counter()
            sets = [setup_of(items[k][0]) for k in group]
            phase["setups"] += time.perf_counter() - tp; tp = time.perf_counter()
            chains = [pkg.SamplingRegistration(cx, st, theta0(k), seed=seed(k)) for cx, st, k in zip(ctxs, sets, group)]
            phase["chains"] += time.perf_counter() - tp; tp = time.perf_counter()
            for k, rec in zip(group, pkg.run_chains_batched(chains, n_steps)):
                rec[:, 0] = k                              # the record's index field carries the item id across the gather
                blocks.append(rec)
            phase["steps"] += time.perf_counter() - tp; tp = time.perf_counter()
            for ch in chains:
                ch.close()
            phase["close"] += time.perf_counter() - tp
    else:                                                  # one context for the rank, its chains one after the other
        for t in my_targets:
            ks = [k for k in mine if items[k][0] == t]
            if not pool:
                pool.append(pkg.IcpContext(model, targets[t], device=device_index))
                contexts_built += 1
            ctx = pool[0]
            if ctx.target is not targets[t]:
                ctx.setTarget(targets[t])
            for k in ks:
                chain = pkg.SamplingRegistration(ctx, setup_of(t), theta0(k), seed=seed(k))
                rec = chain.run(n_steps)
                rec[:, 0] = k
                blocks.append(rec)
                chain.close()
    tp = time.perf_counter()
    for cx in pool:
        cx.close()
    phase["close"] += time.perf_counter() - tp
    t_chains = time.perf_counter()
    import torch
    dev = torch.device("cuda", device_index) if (dist is not None and dist.is_initialized() and dist.get_backend() == "nccl") else None
    per_rank = gather_ragged(blocks, dist, dev)
    t_gather = time.perf_counter()
    out = [None] * len(items)
    for blocks_r in per_rank:
        for b in blocks_r:
            out[int(b[0, 0])] = b
    if return_stats:
        return items, out, dict(items=len(mine), contexts_built=contexts_built, targets_met=len(my_targets),
                                chain_ms=1e3 * (t_chains - t_start), gather_ms=1e3 * (t_gather - t_chains),
                                phase_ms={k: round(1e3 * v, 1) for k, v in phase.items()})
    return items, out
