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
