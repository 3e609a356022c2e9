#!/usr/bin/env python3
"""bench.py — MH iterations/s of the closest-point-proposal path on MI355X (BASELINE.json metric).

One "step" = one Metropolis–Hastings step of the femur configuration of the reference's
apps/femur/IcpProposalRegistration.scala:59-85 (0.9 ICP mixture [TargetSampling + ModelSampling, K = 2·rank,
σt = 10, σn = 5, step 0.1] + 0.1 random walk; prior × independent Gaussian(0, 2) likelihood on 4·rank points)
against the synthetic ~50k-vertex target of BASELINE.json configs[1] (the bundled femur target
subdivided 6-way per edge, 58,322 vertices / 116,640 triangles, seeded 0.05 mm jitter).  Model, target and all
chain state are resident in HBM before the timed region starts; the per-step host<->device traffic is the
(10 + r)-double state vector in and a handful of doubles out.

Multi-GPU (--gpus N, launched by torch.distributed.run): every rank runs an independent chain on its own GPU
(weak scaling, no data-path collective); the fixed-size per-step log records are gathered ONCE with an RCCL
all_gather at log-write time, inside the timed region.

Prints ONE JSON line (rank 0).  Extra legs after the timed region (rank 0, N = 1 only): the roofline of the
dominant kernel measured with HIP events on the library's stream, and the CPU baseline (oracle/, "port", 1 core)
on a bounded sample of the same workload.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
DOMINANT = "k_step_filter"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--profile-steps", type=int, default=300, help="steps of the HIP-event roofline leg (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=16, help="steps of the CPU-oracle baseline leg (0 = skip)")
    ap.add_argument("--subdiv", type=int, default=6, help="edge subdivision of the synthetic target (6 -> 58,322 vertices)")
    ap.add_argument("--chains-per-gpu", type=int, default=1,
                    help="independent chains per GPU, stepped in lockstep through icp_chain_step_batched (default 1 = the BASELINE.json "
                         "configuration; more is the RunMHRandomInitComparison-style many-chains job on fewer GPUs)")
    ap.add_argument("--many-chains", type=int, default=32,
                    help="extra leg after the timed region (1 GPU, 1 chain per GPU only): aggregate rate of this many chains on the GPU "
                         "stepped through icp_chain_step_batched, reported as `many_chains` (0 = skip)")
    ap.add_argument("--fused", type=int, default=2, choices=[0, 1, 2],
                    help="host<->device call pattern per step: 0 per-method calls, 1 propose + icp_chain_eval_step, 2 one icp_chain_step")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch multi-GPU runs with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...")
        args.gpus = world

    dist = torch = None
    if world > 1 or "RANK" in os.environ:  # under torch.distributed.run the RCCL path is taken even with one rank
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as graft
    pkg = graft.load_package()

    # ---- workload (identical on every rank; synthetic target built from the bundled femur data)
    model, target = pkg.data.synthetic_femur_target(n_subdiv=args.subdiv)
    r = model.rank
    B = max(1, args.chains_per_gpu)
    ctxs = [pkg.IcpContext(model, target, device=local_rank) for _ in range(B)]  # (a context holds one chain's scratch)
    ctx = ctxs[0]
    setup = pkg.femur_icp_proposal_registration(model, target, fused=args.fused)
    theta0 = pkg.initial_parameters(model)
    chains = []
    for i in range(B):
        gid = rank * B + i  # chain id within the job
        th = theta0.copy()
        if gid > 0:  # apps/femur/RandomSamplesFromModel.scala:28-35: chain i > 0 starts from c ~ N(0, 0.1·I)
            th[10:] = np.random.default_rng(1024 + gid).normal(size=r) * np.sqrt(0.1)
        chains.append(pkg.SamplingRegistration(ctxs[i], setup, th, seed=1024 + gid))
    chain = chains[0]
    rec_len = 4 + 10 + r

    def run_chains(n):
        """n steps of every chain of this rank -> records [B * n, rec_len]"""
        if B == 1:
            return chain.run(n)
        return np.concatenate(pkg.run_chains_batched(chains, n))

    def barrier():
        if dist is not None:
            torch.cuda.synchronize()
            dist.barrier()
            torch.cuda.synchronize()

    def gather_logs(rec):
        """single RCCL gather of the fixed-size per-step records"""
        if dist is None:
            return rec
        t = torch.from_numpy(rec).cuda()
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t)
        torch.cuda.synchronize()
        return out

    # ---- warmup (also builds the RCCL communicator)
    w = run_chains(max(args.warmup, 1))
    gather_logs(w)

    # ---- timed region: exactly K steps per rank + the log gather
    barrier()
    t0 = time.perf_counter()
    rec = run_chains(args.steps)
    t_chain = time.perf_counter() - t0
    gather_logs(rec)
    t_gather = time.perf_counter() - t0 - t_chain
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    n_acc = int(rec[:, 1].sum())
    n_icp = int((rec[:, 2] < 2).sum())

    line = {
        "metric": "ICP-proposal MH iterations/sec (femur GPMM r=50, ~50k-vtx target)",
        "value": world * B * args.steps / dt,
        "unit": "iterations/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": "BASELINE.json configs[1]: femur 50-basis GPMM (N=%d, rank %d) vs synthetic target M=%d vertices / %d triangles; "
                        "%d chain%s per GPU; 0.9 ICP(Target+Model sampling, K=%d) + 0.1 random walk; prior x independent Gaussian(0,2) on %d points"
                        % (model.n_points, r, target.n_points, target.n_cells, B, "" if B == 1 else "s (lockstep, icp_chain_step_batched)", 2 * r, 4 * r),
            "chains_per_gpu": B,
            "calls_per_step": {0: "per-method", 1: "propose + icp_chain_eval_step", 2: "icp_chain_step"}[args.fused] if B == 1 else "icp_chain_step_batched",
            "chain_ms": 1e3 * t_chain,
            "log_gather_ms": 1e3 * t_gather,
            "accepted": n_acc,
            "icp_proposals": n_icp,
        },
        "roofline": None,
        "cpu_baseline": None,
    }

    if rank == 0 and world == 1:
        # ---- roofline of the dominant kernel: HIP events on the stream the kernel runs on (the library's stream)
        if args.profile_steps > 0:
            ctx.profile_start(max_launches=64 * args.profile_steps + 1024)   # (a batch's launches go out on the first chain's context)
            if B == 1:
                chain.run(args.profile_steps, want_records=False)
            else:
                pkg.run_chains_batched(chains, args.profile_steps, want_records=False)
            stats = ctx.profile_stop()
            if DOMINANT in stats:
                k = stats[DOMINANT]
                n_queries = setup.eval["n_model_ids"]  # ids 0..4r-1 (the proposal's 0..2r-1 are a subset, shared)
                # algorithmic bytes per launch: target vertices 3·M·8 + target triangles 3·Tt·4 + queries K·3·8
                alg_bytes = 3 * target.n_points * 8 + 3 * target.n_cells * 4 + n_queries * 24
                if DOMINANT == "k_step_filter":  # the merged launch also holds the TargetSampling search: model vertices + its queries
                    alg_bytes += 3 * model.n_points * 8 + 2 * r * 24
                chains_per_launch = 1.0
                if B > 1:  # one launch holds the searches of a group of chains: units per launch = chain steps / launches
                    chains_per_launch = B * args.profile_steps / max(k["calls"], 1)
                    alg_bytes = int(alg_bytes * chains_per_launch)
                avg_s = k["avg_us"] * 1e-6
                achieved = alg_bytes / avg_s / 1e9
                traffic = None
                tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
                if os.path.exists(tfile) and B == 1:
                    traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
                line["roofline"] = {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                    "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "kernel": DOMINANT,
                                    "avg_launch_us": k["avg_us"], "launches": k["calls"], "algorithmic_bytes": alg_bytes,
                                    "chains_per_launch": chains_per_launch,
                                    "note": "one chain per launch: the filter is a chain of latencies (spheres + queries in, patch test, sphere tests, hits out), "
                                            "not a stream; HBM fraction reported as north_star asks (more chains per launch: --chains-per-gpu)"}
                line["kernel_us_per_step"] = {name: round(s["total_ms"] * 1e3 / args.profile_steps, 2) for name, s in stats.items()}
                line["kernel_us_per_step_note"] = ("HIP events around each launch (these add ~3 us per launch); a step's launches alternate between two "
                                                   "streams and overlap the previous step's finish launch; k_step_begin includes the time it waits ON THE "
                                                   "DEVICE for that launch to start or for the eigen-decomposition it draws from (k_posterior_eigen, side stream)")
        # ---- CPU baseline: the oracle's chain (same math, brute force, 1 thread) on a bounded sample
        if args.cpu_steps > 0 and B == 1:
            from oracle import oracle as O
            om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
            icp = [O.proposal_params(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], p.get("boundary_aware", True),
                                     n_model_ids=p.get("n_model_ids", 0), target_pts=p.get("target_pts")) for p in setup.icp]
            e = setup.eval
            ep = O.evaluator_params(e["kind"], e["mode"], n_model_ids=e["n_model_ids"], target_pts=e["target_pts"],
                                    p0=e["gauss_mean"], p1=e["gauss_sigma"], p2=e["exp_rate"])
            cfg = O.chain_config(icp, [p.get("weight", 0.5) for p in setup.icp], setup.w_icp, setup.w_rw, setup.rw_sigma, ep)
            t1 = time.perf_counter()
            acc_o, comp_o, _, states_o = O.run_chain(om, ot, cfg, theta0, 1024, args.cpu_steps)
            cdt = time.perf_counter() - t1
            # the first cpu_steps records of a fresh GPU chain must reproduce the oracle's decisions
            chk = pkg.SamplingRegistration(ctx, setup, theta0, seed=1024)
            crec = chk.run(args.cpu_steps)
            same = bool(np.array_equal(crec[:, 1].astype(np.uint8), acc_o)) and \
                float(np.abs(crec[:, 14:] - states_o[:, 10:]).max()) <= 1e-5 * max(float(np.abs(states_o[:, 10:]).max()), 1e-30)
            chk.close()
            line["cpu_baseline"] = {"value": args.cpu_steps / cdt, "unit": "iterations/s", "cores": 1, "kind": "port",
                                    "sample": "%d MH steps of the same workload, oracle/icp_oracle.c (brute-force f64, single thread, "
                                              "posterior/likelihood of the current state carried over like the reference's Memoize); "
                                              "host has %d logical cores" % (args.cpu_steps, os.cpu_count()),
                                    "gpu_matches_oracle_on_sample": same}
    for ch in chains:
        ch.close()
    for cx in ctxs:
        cx.close()
    if rank == 0 and world == 1 and B == 1 and args.many_chains > 1:
        # ---- not the headline: the same workload with many independent chains on the one GPU (SURVEY.md §8e "within a GPU,
        # batch B chains per launch"; RunMHRandomInitComparison-style jobs), one context per chain, lockstep submissions
        try:
            nB = args.many_chains
            mctx = [pkg.IcpContext(model, target, device=local_rank) for _ in range(nB)]
            mch = []
            for i in range(nB):
                th = theta0.copy()
                if i > 0:
                    th[10:] = np.random.default_rng(1024 + i).normal(size=r) * np.sqrt(0.1)
                mch.append(pkg.SamplingRegistration(mctx[i], setup, th, seed=1024 + i))
            pkg.run_chains_batched(mch, 40, want_records=False)
            n_m = 300
            t1 = time.perf_counter()
            pkg.run_chains_batched(mch, n_m, want_records=False)
            mdt = time.perf_counter() - t1
            line["many_chains"] = {"chains_per_gpu": nB, "value": nB * n_m / mdt, "unit": "iterations/s", "steps_per_chain": n_m,
                                   "entry_point": "icp_chain_step_batched"}
            for ch in mch:
                ch.close()
            for cx in mctx:
                cx.close()
        except Exception as e:  # the headline line is printed whatever happens in this extra leg
            line["many_chains"] = {"error": str(e)[:200]}
    if rank == 0:
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
