"""Multi-GPU plumbing for independent chains: static assignment of (target, chain) work items to
ranks and the single gather of fixed-size per-step log records at log-write time.

Chains never exchange data while sampling (reference: apps/femur/RunMHRandomInitComparison.scala:66-87 runs them as
independent JVM threads), so there is no data-path collective; `gather_records` is the only communication and uses
whatever backend the process group was created with (RCCL = "nccl" on the GPUs, gloo in the CPU tests)."""
from __future__ import annotations

import numpy as np


def assign_work_items(n_items: int, world_size: int):
    """Round-robin: item i -> rank i % world_size.  Returns a list (per rank) of item indices; sizes differ by <= 1."""
    return [list(range(rk, n_items, world_size)) for rk in range(world_size)]


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
              chains_per_launch: int = 1):
    """Batch registration (BASELINE.json configs[4]; reference: the 10-way target pool × per-target chain loop of
    apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:106-163): work items = (target, chain) pairs, dealt round-robin
    over the ranks; a rank keeps ONE context per target it meets; chains never communicate; the per-step records of all items
    are exchanged with a single all_gather at the end.  Returns (items, records): items[k] = (target index, chain index) and
    records[k] = [n_steps, 14 + rank] for every item of the whole job, in item order, on every rank.
    chains_per_launch > 1: the rank steps that many of its chains in lockstep through icp_chain_step_batched (one context per
    chain; SURVEY.md §8e "within a GPU, batch B chains per launch") — same records, chain by chain."""
    rank = dist.get_rank() if (dist is not None and dist.is_initialized()) else 0
    world = dist.get_world_size() if (dist is not None and dist.is_initialized()) else 1
    items = [(t, c) for t in range(len(targets)) for c in range(n_chains)]
    mine = assign_work_items(len(items), world)[rank]
    blocks, ctx, ctx_target = [], None, -1
    order = sorted(mine, key=lambda k: items[k][0])
    if chains_per_launch > 1:
        for g0 in range(0, len(order), chains_per_launch):
            group = order[g0:g0 + chains_per_launch]
            ctxs = [pkg.IcpContext(model, targets[items[k][0]], device=device_index) for k in group]
            chains = [pkg.SamplingRegistration(cx, make_setup(model, targets[items[k][0]]),
                                               pkg.random_initial_parameters(model, items[k][1], base_seed),
                                               seed=base_seed + 1000 * items[k][0] + items[k][1]) for cx, k in zip(ctxs, group)]
            for k, rec in zip(group, pkg.run_chains_batched(chains, n_steps)):
                rec[:, 0] = k
                blocks.append(rec)
            for ch in chains:
                ch.close()
            for cx in ctxs:
                cx.close()
        order = []
    for k in order:                                            # target-major: one context per target
        t, c = items[k]
        if t != ctx_target:
            if ctx is not None:
                ctx.close()
            ctx, ctx_target = pkg.IcpContext(model, targets[t], device=device_index), t
        chain = pkg.SamplingRegistration(ctx, make_setup(model, targets[t]), pkg.random_initial_parameters(model, c, base_seed),
                                         seed=base_seed + 1000 * t + c)
        rec = chain.run(n_steps)
        rec[:, 0] = k                                          # the record's index field carries the item id across the gather
        blocks.append(rec)
        chain.close()
    if ctx is not None:
        ctx.close()
    import torch
    dev = torch.device("cuda", device_index) if (dist is not None and dist.is_initialized() and dist.get_backend() == "nccl") else None
    per_rank = gather_ragged(blocks, dist, dev)
    out = [None] * len(items)
    for blocks_r in per_rank:
        for b in blocks_r:
            out[int(b[0, 0])] = b
    return items, out

