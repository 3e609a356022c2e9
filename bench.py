#!/usr/bin/env python3
"""bench.py — MH iterations/s of the closest-point-proposal path on MI355X (BASELINE.json metric).

One "step" = one Metropolis–Hastings step of the reference's experiment configuration named by --config:

  1 (default, the configuration BASELINE.json's metric is quoted on)  apps/femur/IcpProposalRegistration.scala:59-85:
      femur 50-basis GPMM, 0.9 ICP mixture [TargetSampling + ModelSampling, K = 2·rank, σt = 10, σn = 5, step 0.1] + 0.1 random
      walk; prior × independent Gaussian(0, 2) likelihood on 4·rank points; synthetic ~50k-vertex target (the
      bundled femur target subdivided 6-way per edge, 58,322 vertices / 116,640 triangles, seeded 0.05 mm jitter)
  2  apps/femur/RunMHRandomInitComparison.scala:54-61: femur 100-basis GPMM, every model point a sample point (K = N = 1622),
      ICP(ModelSampling), SymmetricEvaluation; same 58k target; chain i > 0 starts from c ~ N(0, 0.1·I)
  3  apps/bfm/BfmFittingPartial.scala:62-83 on the BFM-sized synthetic stand-in (N = 28,561, rank 200, partial target with a
      boundary): 0.4 pose + 0.55 ICP(ModelSampling, K = 400) + 0.05 random walk, full-mesh Hausdorff evaluator

  4  batch registration (apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:106-163 on the configs[3] problem): --targets x
      --chains work items (target, random initial shape) dealt target-major over the ranks (sharding.assign_target_major), every
      chain --steps long, chains of one target on one rank stepped through icp_chain_step_batched; ONE ragged RCCL gather of all
      records at the end.  value = MH iterations of the whole job per second.

Model, target and all chain state are resident in HBM before the timed region starts; the per-step host<->device traffic is
the (10 + r)-double state vector in and a handful of doubles out.

Multi-GPU (--gpus N): every rank runs an independent chain on its own GPU (weak scaling, no data-path collective); the
fixed-size per-step log records are gathered ONCE with an RCCL all_gather at log-write time, inside the timed region.  Launched
by `python -m torch.distributed.run` the ranks are taken from the environment; launched plainly (`python bench.py --gpus N`) this
process — which never touches a GPU — starts the N rank processes itself and relays rank 0's line.

Prints ONE JSON line (rank 0).  Extra legs after the timed region (rank 0, N = 1 only): per-kernel time with HIP events on the
library's streams -> `roofline` of the time-dominant kernel (+ the whole step and the distance kernel against both rooflines),
and the CPU baselines B1 / B2 of BASELINE.md §3 (oracle/, bounded samples).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (6.29 TB/s measured copy)
F32_VECTOR_TFLOPS = 157.3  # same guide: peak FP32 vector (the filters run on packed f32; the exact resolves in f64)
F64_VECTOR_TFLOPS = 78.6   # AMD's public MI355X figure for vector FP64 (not in the local guide; SURVEY.md §8d uses it)
F64_MATRIX_TFLOPS = 78.6   # … and for matrix FP64 (v_mfma_f64_16x16x4: 1,024 multiply-adds in 32 cycles per SIMD = the vector rate)
# what the instruction was MEASURED to sustain on this pool (tools/src/mfma_f64_bench.hip, round 6: four independent accumulators per wave,
# back to back): stated beside the data-sheet peak the fractions are priced against
F64_MATRIX_MEASURED = {"tool": "tools/src/mfma_f64_bench.hip", "TFLOPs_by_waves_per_simd": {"1": 32.2, "2": 42.5, "4": 46.9},
                       "ns_per_instruction_one_wave": 60.2}
N_SIMD = 256 * 4           # 256 CUs x 4 SIMDs (MI355X_MICROARCH.md)
N_XCD = 8
METRIC = "ICP-proposal MH iterations/sec (femur GPMM r=50, ~50k-vtx target)"


# ---------------------------------------------------------------------------------------------- launcher
def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: N child processes, one per GPU, over 127.0.0.1.
    This parent makes no GPU call (it only spawns, waits and relays)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for rk in range(args.gpus):
        env = dict(os.environ, RANK=str(rk), LOCAL_RANK=str(rk), WORLD_SIZE=str(args.gpus), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if rk == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(rk, rc) for rk, rc in enumerate(rcs) if rc != 0]
    if bad:
        raise SystemExit("rank processes failed: " + ", ".join(f"rank {rk} -> exit {rc}" for rk, rc in bad))


def rank_report(dist, torch, rank, world, local_rank, chain_ms, gather_ms, cpu=False):
    """What every rank of the job held and how long it took, gathered over the job's own communicator: `rccl_ranks` is the size the
    COMMUNICATOR reports (not the launcher's environment), `ranks` one entry per rank with its device ordinal, an identity of the
    physical GPU (uuid, or PCI address where the build has none), its chain and gather times.  Two ranks on one physical device
    fail the run: a scaling curve measured that way is not one."""
    if dist is None:
        ident = device_identity(torch, local_rank, cpu)
        return {"rccl_ranks": 1, "backend": None, "ranks": [dict(rank=0, local_rank=local_rank, chain_ms=chain_ms, log_gather_ms=gather_ms, **ident)]}
    mine = dict(rank=rank, local_rank=local_rank, chain_ms=chain_ms, log_gather_ms=gather_ms, **device_identity(torch, local_rank, cpu))
    everyone = [None] * dist.get_world_size()
    dist.all_gather_object(everyone, mine)
    gpus = [e["gpu"] for e in everyone]
    if len(set(gpus)) != len(gpus):
        raise SystemExit("two ranks hold the same device: " + json.dumps(everyone))
    if dist.get_world_size() != world:
        raise SystemExit("the communicator has %d ranks, the launcher announced %d" % (dist.get_world_size(), world))
    return {"rccl_ranks": dist.get_world_size(), "backend": dist.get_backend(), "ranks": everyone}


def device_identity(torch, local_rank, cpu=False):
    if cpu:
        return {"device": "cpu", "gpu": "cpu-process-%d" % os.getpid(), "name": "host"}
    if torch is None:  # (a single process outside a launcher never imports torch: the library's own HIP ordinal)
        return {"device": "hip:%d" % local_rank, "gpu": "hip-ordinal-%d-of-process-%d" % (local_rank, os.getpid()), "name": "MI355X (not queried: no torch in this process)"}
    pr = torch.cuda.get_device_properties(local_rank)
    ident = getattr(pr, "uuid", None)
    if ident is None or not str(ident).strip("0-"):  # (builds without a uuid, or an all-zero one: the PCI address)
        ident = "pci-%s:%s:%s" % (getattr(pr, "pci_domain_id", "?"), getattr(pr, "pci_bus_id", "?"), getattr(pr, "pci_device_id", "?"))
    return {"device": "cuda:%d" % torch.cuda.current_device(), "gpu": str(ident), "name": pr.name}


def selftest_launcher_rank():
    """Body of a rank under --selftest-launcher (CPU, gloo): exercises the launcher's environment, the barrier-bracketed timing
    and the gather; prints the same kind of line from rank 0.  Used by tests/test_bench_launcher_cpu.py (no GPU there)."""
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if "--config=4" in sys.argv or ("--config" in sys.argv and sys.argv[sys.argv.index("--config") + 1] == "4"):
        selftest_config4(dist, rank, world)
        return
    rec = np.full((5, 8), float(rank))
    dist.barrier()
    t0 = time.perf_counter()
    t = torch.from_numpy(rec)
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ok = all(float(o[0, 0]) == float(k) for k, o in enumerate(out))
    rep = rank_report(dist, None, rank, world, int(os.environ.get("LOCAL_RANK", "0")), 1e3 * float(dt.item()), 0.0, cpu=True)
    if rank == 0:
        print(json.dumps({"selftest": "launcher", "n_gpus": world, "gather_ok": ok, "max_s": float(dt.item()), "multi_gpu": rep}))
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


def selftest_config4(dist, rank, world):
    """--config 4 --selftest-launcher (CPU, gloo): the batch job's plumbing without a GPU — target-major assignment of the
    10 x 10 work items, a fabricated record block per item, the ragged gather, reassembly in item order."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("icp_sharding", os.path.join(ROOT, "icp-proposal_amd", "sharding.py"))
    sharding = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(sharding)
    n_targets, n_chains, n_steps, ln = 10, 10, 4, 14 + 5
    assign = sharding.assign_target_major(n_targets, n_chains, world)
    blocks = []
    for k in assign[rank]:
        b = np.full((n_steps, ln), float(k))
        b[:, 1] = rank
        blocks.append(b)
    t0 = time.perf_counter()
    per_rank = sharding.gather_ragged(blocks, dist)
    gather_ms = 1e3 * (time.perf_counter() - t0)
    got = {}
    for rk, bl in enumerate(per_rank):
        for b in bl:
            got[int(b[0, 0])] = (rk, b)
    ok = sorted(got) == list(range(n_targets * n_chains)) and all(got[k][0] == next(r for r in range(world) if k in assign[r]) for k in got)
    rep = rank_report(dist, None, rank, world, int(os.environ.get("LOCAL_RANK", "0")), 0.0, gather_ms, cpu=True)
    if rank == 0:
        print(json.dumps({"selftest": "config4", "n_gpus": world, "gather_ok": bool(ok), "items_per_rank": [len(a) for a in assign],
                          "targets_per_rank": [len(set(k // n_chains for k in a)) for a in assign], "gather_ms": gather_ms, "multi_gpu": rep}))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit(3)


# ---------------------------------------------------------------------------------------------- workloads
def face_model(pkg, args):
    kw = {}
    if args is not None and args.face_grid:
        kw["grid"] = args.face_grid
    if args is not None and args.face_rank:
        kw["rank"] = args.face_rank
    return pkg.data.synthetic_face_model(**kw)


def build_workload(pkg, config, subdiv, fused, args=None):
    """-> dict(model, target, setup, name, init(gid))"""
    if config == 1:
        model, target = pkg.data.synthetic_femur_target(n_subdiv=subdiv)
        setup = pkg.femur_icp_proposal_registration(model, target, fused=fused)
        name = ("BASELINE.json configs[1]: femur 50-basis GPMM (N=%d, rank %d) vs synthetic target M=%d vertices / %d triangles; "
                "0.9 ICP(Target+Model sampling, K=%d) + 0.1 random walk; prior x independent Gaussian(0,2) on %d points; target-side sample "
                "points = a deterministic VERTEX SUBSET of the target (data.decimated_point_subset), a stand-in for the positions of "
                "Scalismo's quadric decimation (NonRigidIcpProposal.scala:46), which stays on the JVM side of the boundary"
                % (model.n_points, model.rank, target.n_points, target.n_cells, 2 * model.rank, 4 * model.rank))
    elif config == 2:
        model, target = pkg.data.synthetic_femur_target(n_subdiv=subdiv, n_components=100)
        setup = pkg.femur_random_init_comparison(model, target, fused=fused)
        name = ("BASELINE.json configs[2] (one chain per GPU): femur 100-basis GPMM (N=%d, rank %d) vs synthetic target M=%d vertices / %d "
                "triangles; ICP(ModelSampling, K = N = %d); prior x independent Gaussian(0,2), SymmetricEvaluation on all %d points"
                % (model.n_points, model.rank, target.n_points, target.n_cells, model.n_points, model.n_points))
    elif config == 3:
        model = face_model(pkg, args)
        target = pkg.data.synthetic_partial_target(model, seed=7)
        setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff", fused=fused)
        name = ("BASELINE.json configs[3]: BFM-face12-sized synthetic stand-in (N=%d, T=%d, rank %d) vs partial target M=%d vertices / %d "
                "triangles (with boundary); 0.4 pose + 0.55 ICP(ModelSampling, K=%d) + 0.05 random walk; prior x full-mesh Hausdorff "
                "evaluator" % (model.n_points, model.n_cells, model.rank, target.n_points, target.n_cells, 2 * model.rank))
    else:
        raise SystemExit("--config must be 1, 2, 3 or 4")

    def init(gid):
        if config == 2:  # every chain of the random-init comparison starts from a random shape (chain 0 from the mean)
            return pkg.random_initial_parameters(model, gid)
        th = pkg.initial_parameters(model)
        if gid > 0:  # apps/femur/RandomSamplesFromModel.scala:28-35: chain i > 0 starts from c ~ N(0, 0.1·I)
            th[10:] = np.random.default_rng(1024 + gid).normal(size=model.rank) * np.sqrt(0.1)
        return th
    return dict(model=model, target=target, setup=setup, name=name, init=init, config=config)


def algorithmic_step(model, target, setup, target_has_boundary):
    """ALGORITHMIC bytes and flops of one MH step whose proposal came from the ICP mixture (brute-force
    definition: every operand read once per step; w = 8)."""
    N, T, r = model.n_points, model.n_cells, model.rank
    M, Tt = target.n_points, target.n_cells
    dirs = len(setup.icp)
    K = max([p.get("n_model_ids", 0) or np.asarray(p.get("target_pts", np.zeros((0, 3)))).reshape(-1, 3).shape[0] for p in setup.icp] + [0])
    e = setup.eval
    b = 3 * N * r * 8 + 3 * N * 8 * 2 + 3 * T * 4 + 3 * Tt * 4 + 3 * M * 8 + dirs * K * 3 * r * 8 + 3 * r * r * 8
    f = 2 * 3 * N * r + 30 * T + 12 * N
    for p in setup.icp:
        if p["direction"] == 1:
            f += 8 * np.asarray(p["target_pts"]).reshape(-1, 3).shape[0] * N
        else:
            f += 60 * p["n_model_ids"] * Tt + (8 * p["n_model_ids"] * M if target_has_boundary else 0)
    f += dirs * (2 * 3 * K * r * r + 18 * K * r) + dirs * (r ** 3 / 3 + 10 * r ** 3)
    if e["kind"] == 1:
        f += 60 * (N * Tt + M * T)
    else:
        if e["mode"] in (0, 2):
            f += 60 * e["n_model_ids"] * Tt
        if e["mode"] in (1, 2):
            f += 60 * np.asarray(e["target_pts"]).reshape(-1, 3).shape[0] * T
    return float(b), float(f)


def kernel_algorithmic_bytes(name, model, target, setup):
    """Algorithmic bytes of ONE launch of the named kernel (what it has to read and write at least once)."""
    N, T, r = model.n_points, model.n_cells, model.rank
    M, Tt = target.n_points, target.n_cells
    dirs = len(setup.icp)
    Km = max([p.get("n_model_ids", 0) for p in setup.icp] + [0])
    Kt = max([np.asarray(p.get("target_pts", np.zeros((0, 3)))).reshape(-1, 3).shape[0] for p in setup.icp if p["direction"] == 1] + [0])
    e = setup.eval
    Ke = e["n_model_ids"]
    Ksurf = max(Km, Ke)
    splits = lambda k: min(64, max(1, (k + 7) // 8))
    n1 = (r + 1) * (r + 1)
    if name == "k_step_filter":      # target vertices + triangle ids + surface queries; the TargetSampling search: model vertices + its queries
        return 3 * M * 8 + 3 * Tt * 4 + Ksurf * 24 + (3 * N * 8 + Kt * 24 if Kt else 0)
    if name == "k_step_begin":       # scaled basis + reference + mean in, instance out, V and P of the proposal, query records
        return 3 * N * r * 8 + 3 * N * 8 * 3 + 2 * r * r * 8 + Ksurf * 24
    if name == "k_step_resolve":     # query points, winners' triangles (3 corners), correspondence records out
        return (Ksurf + Kt) * (24 + 72) + (Km + Kt) * 100
    if name == "k_step_regression":  # gathered basis rows + correspondence records in, split-K partials out (lower-triangle tiles)
        return sum((Km if p["direction"] == 0 else Kt) * (3 * r * 8 + 56) + splits(Km if p["direction"] == 0 else Kt) * n1 * 8 * 0.625 for p in setup.icp)
    if name == "k_step_finish":      # partials in (lower triangle), M out, M + G^-1 for the two tails of every posterior
        return sum(splits(Km if p["direction"] == 0 else Kt) * n1 * 8 * 0.5 + r * r * 8 * 2 for p in setup.icp) + 2 * r * r * 8
    if name.startswith("k_posterior_eigen"):
        if r > 64:  # the tridiagonal route, one posterior per launch sequence: M in, reflectors out and in, V and Vt out (+ S)
            return r * r * 8 * 5 + r * 8
        # ranks <= 64, per launch (both directions): M, warm basis in, V, Vt, S out, rotation log out and in
        return dirs * (r * r * 8 * 4 + 3 * 51 * (r + 1) * 8 * 2)
    hd = e["kind"] == 1
    if name == "k_surface_filter":   # per-stage launch, or the wide step's evaluator sequence (Hausdorff: both full-mesh directions)
        if hd:
            return 3 * M * 8 + 3 * Tt * 4 + 3 * N * 8 + 3 * T * 4 + (N + M) * 24
        return 3 * M * 8 + 3 * Tt * 4 + Ksurf * 24
    if name == "k_surface_resolve":
        return (N + M) * (24 + 72 + 40) if hd else Ksurf * (24 + 72 + 40)
    if name == "k_vertex_filter":
        return 3 * M * 8 + Ksurf * 24
    if name == "k_instance":         # scaled basis + reference + mean in; instance and kept deformations out
        return 3 * N * r * 8 + 3 * N * 8 * 4
    if name == "k_posterior_factor": # summed partial in, M out, the scaled factor out and back in (ranks whose factor lives in global scratch)
        return n1 * 8 + r * r * 8 + (2 * (r + 1) * r * 8 if r > 116 else 0)
    if name == "k_transition_tail":  # M and G^-1 per tail (two tails per posterior)
        return 2 * dirs * 2 * r * r * 8
    if name == "k_propose":
        return 2 * r * r * 8
    return None



# ---------------------------------------------------------------------------------------------- CPU baseline: the reference's own parallelism
def oracle_chain_setup(O, wl):
    """the oracle's model / target / chain configuration of a workload (cpu_baseline legs only)"""
    model, target, setup = wl["model"], wl["target"], wl["setup"]
    om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
    icp = [O.proposal_params(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], p.get("boundary_aware", True),
                             n_model_ids=p.get("n_model_ids", 0), target_pts=p.get("target_pts")) for p in setup.icp]
    e = setup.eval
    ep = O.evaluator_params(e["kind"], e["mode"], n_model_ids=e["n_model_ids"], target_pts=e["target_pts"],
                            p0=e["gauss_mean"] if e["kind"] != 1 else e["exp_rate"], p1=e["gauss_sigma"], p2=e["exp_rate"])
    cfg = O.chain_config(icp, [p.get("weight", 0.5) for p in setup.icp], setup.w_icp, setup.w_rw, setup.rw_sigma, ep, w_pose=setup.w_pose,
                         pose_rot_sigma=setup.pose_rot_sigma, pose_trans_sigma=setup.pose_trans_sigma)
    return om, ot, cfg


def b1_child(args):
    """One of the processes of `b1_parallel` (never touches a GPU): a B1 chain of its own — oracle/, KD-tree + BVH, one thread — of the
    named configuration's workload.  Protocol over stdin/stdout: prints {"ready": seconds per step of a two-step probe} once its data and
    the static target's structures exist, reads "go <n>", runs n steps, prints {"steps": n, "seconds": ...}."""
    import __graft_entry__ as graft
    pkg = graft.load_package()  # (numpy side only: the HIP library is loaded on first use, and nothing here uses it)
    from oracle import oracle as O
    cfg_i, gid = args.b1_child
    if cfg_i == 4:    # one (target, chain) item of the batch registration: target gid // 10, random start gid % 10
        model = face_model(pkg, args)
        target = pkg.data.synthetic_partial_target(model, seed=100 + gid // 10)
        wl = dict(model=model, target=target, setup=pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=args.fused),
                  init=lambda g: pkg.random_initial_parameters(model, g % 10))
    else:
        wl = build_workload(pkg, cfg_i, args.subdiv, args.fused, args)
    om, ot, cfg = oracle_chain_setup(O, wl)
    theta0 = wl["init"](gid)
    O.set_search_backend(O.SEARCH_TREES)
    O.run_chain(om, ot, cfg, theta0, 1024 + gid, 1)  # (builds the static target's structures, as the reference does once)
    t = time.perf_counter()
    O.run_chain(om, ot, cfg, theta0, 1024 + gid, 2)
    print(json.dumps({"ready": (time.perf_counter() - t) / 2}), flush=True)
    word = sys.stdin.readline().split()
    n = int(word[1]) if len(word) == 2 and word[0] == "go" else 0
    if n <= 0:
        return
    t = time.perf_counter()
    O.run_chain(om, ot, cfg, theta0, 1024 + gid, n)
    print(json.dumps({"steps": n, "seconds": time.perf_counter() - t}), flush=True)


def b1_parallel(args, cfg_i, n_chains, budget_s, max_procs=64):
    """The reference's own CPU parallelism (RunMHRandomInitComparison.scala:66 `.par` over chains,
    StdIcpVsChainICPrandomInitComparisonAll.scala:106-108 over targets): P = min(chains, cores this process may use, max_procs)
    INDEPENDENT B1 chains — one single-threaded process each — running at the same time; value = steps of all of them / the longest
    one's wall time.  The chains start together (a "go" after every process has its data) and take about `budget_s` seconds."""
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    P = max(1, min(n_chains, avail, max_procs))
    base = [sys.executable, os.path.abspath(__file__), "--subdiv", str(args.subdiv), "--fused", str(args.fused),
            "--face-grid", str(args.face_grid), "--face-rank", str(args.face_rank)]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    procs = [subprocess.Popen(base + ["--b1-child", str(cfg_i), str(g)], stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True, env=env) for g in range(P)]
    try:
        probes = [json.loads(p.stdout.readline())["ready"] for p in procs]
        probe = sorted(probes)[len(probes) // 2]
        n = int(max(1, min(400, budget_s / max(probe, 1e-6))))
        t0 = time.perf_counter()
        for p in procs:
            p.stdin.write("go %d\n" % n)
            p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        wall = time.perf_counter() - t0
    finally:
        for p in procs:
            try:
                p.stdin.close()
            except Exception:
                pass
        for p in procs:
            p.wait()
    longest = max(r["seconds"] for r in res)
    return {"value": P * n / max(longest, 1e-9), "unit": "iterations/s", "cores": P, "chains": P, "steps_per_chain": n,
            "longest_chain_s": longest, "wall_s": wall, "per_chain_it_s": [round(r["steps"] / r["seconds"], 2) for r in res][:8],
            "host_logical_cores": os.cpu_count() or 0, "cores_available_to_this_process": avail,
            "what": "%d independent B1 chains (oracle/, KD-tree + BVH, one thread each, one process each) at the same time — the reference's own "
                    "parallelism over chains / targets (RunMHRandomInitComparison.scala:66, StdIcpVsChainICPrandomInitComparisonAll.scala:106-108); "
                    "value = all their steps / the longest chain's time" % P,
            "sample": "%d chains x %d MH steps of BASELINE.json configs[%d]" % (P, n, cfg_i)}


def mfma_block(config_key, kernels, live_us, flops):
    """MFMA utilisation of the projection kernels (north_star: "MFMA utilisation on the projection … against gfx950 peak"), two ways:
      counter: SQ_VALU_MFMA_BUSY_CYCLES / (N_SIMD x GRBM_GUI_ACTIVE / N_XCD) per launch, both counters from ONE rocprofv3 --pmc pass
               (profiles/r06_pmc_mfma.json; SQ_VALU_MFMA_BUSY_CYCLES counts cycles per SIMD, summed over the chip's 1,024 SIMDs;
               rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs — MI355X_MICROARCH.md "DVFS give-back" —, so /8 = the launch's
               cycles);
      flops:   algorithmic f64 multiply-add flops of one launch / the launch's duration measured in THIS run (HIP events) / the dense
               matrix-f64 peak."""
    out = {"peak_TFLOPs": F64_MATRIX_TFLOPS, "measured_instruction_rate": F64_MATRIX_MEASURED,
           "formula": "busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (%d SIMDs x GRBM_GUI_ACTIVE / %d XCDs); flops_frac = algorithmic f64 flops per launch / "
                      "avg launch duration (HIP events, this run) / %.1f TFLOP/s" % (N_SIMD, N_XCD, F64_MATRIX_TFLOPS), "kernels": {}}
    pmc = {}
    for tname in ("r06_pmc_mfma.json", "r05_pmc_mfma.json", "r04_pmc_mfma.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile):
            pmc = json.load(open(tfile)).get(config_key, {})
            out["counter_source"] = "profiles/" + tname + " (a tracked file from a rocprofv3 --pmc pass of the same command, NOT measured in this run)"
            if pmc:
                break
    for k in kernels:
        row = {}
        c = pmc.get(k, {})
        busy, act = c.get("SQ_VALU_MFMA_BUSY_CYCLES", {}).get("median"), c.get("GRBM_GUI_ACTIVE", {}).get("median")
        if busy is not None and act:
            row.update(SQ_VALU_MFMA_BUSY_CYCLES=busy, GRBM_GUI_ACTIVE=act,
                       busy_frac=c.get("busy_frac_median") if c.get("busy_frac_median") is not None else busy / (N_SIMD * act / N_XCD))
            if c.get("median_us") is not None and flops.get(k) and k not in live_us:  # (no event id of its own: the counter pass's dispatch timestamps)
                row.update(avg_launch_us=c["median_us"], duration_source="the counter pass", algorithmic_f64_flops=flops[k],
                           flops_frac=flops[k] / (c["median_us"] * 1e-6) / 1e12 / F64_MATRIX_TFLOPS)
        if k in live_us and flops.get(k):
            row.update(avg_launch_us=live_us[k], algorithmic_f64_flops=flops[k], achieved_TFLOPs=flops[k] / (live_us[k] * 1e-6) / 1e12,
                       flops_frac=flops[k] / (live_us[k] * 1e-6) / 1e12 / F64_MATRIX_TFLOPS)
        if row:
            out["kernels"][k] = row
    return out


def projection_flops(model, setup):
    """algorithmic f64 flops of ONE regression launch (all of the step's posteriors): per correspondence four rank-1 terms (the three
    rows of Q_i and Q_i^T n_i) on the lower triangle incl. diagonal of the (r+1) x (r+1) normal equations, 2 flops per multiply-add"""
    r = model.rank
    tri = (r + 1) * (r + 2) // 2
    tot = 0
    for p in setup.icp:
        K = p.get("n_model_ids", 0) if p["direction"] == 0 else np.asarray(p.get("target_pts", np.zeros((0, 3)))).reshape(-1, 3).shape[0]
        tot += 2 * 4 * K * tri
    return float(tot)


# ---------------------------------------------------------------------------------------------- main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--config", type=int, default=1, help="BASELINE.json configs[i]: 1 (metric configuration), 2, 3, 4 (batch registration)")
    ap.add_argument("--targets", type=int, default=10, help="--config 4: number of targets")
    ap.add_argument("--chains", type=int, default=10, help="--config 4: random-init chains per target")
    ap.add_argument("--face-grid", type=int, default=0, help="--config 3/4: side of the synthetic face grid (0 = 169: N = 28,561)")
    ap.add_argument("--face-rank", type=int, default=0, help="--config 3/4: rank of the synthetic face model (0 = 200)")
    ap.add_argument("--extra-configs", type=str, default="2,3,4",
                    help="default run (N = 1, config 1): short legs of these other configurations after the timed region, reported as "
                         "`extra_configs` (not the headline; '' = none)")
    ap.add_argument("--profile-steps", type=int, default=300, help="steps of the HIP-event roofline leg (0 = skip)")
    ap.add_argument("--cpu-steps", type=int, default=96, help="steps of the B2 leg of the CPU baseline (B1 runs 4x as many; 0 = skip)")
    ap.add_argument("--subdiv", type=int, default=6, help="edge subdivision of the synthetic femur target (6 -> 58,322 vertices)")
    ap.add_argument("--chains-per-gpu", type=int, default=0,
                    help="independent chains per GPU, stepped in lockstep through icp_chain_step_batched (0 = default: 1 for --config 1-3, the "
                         "BASELINE.json configuration — more is the RunMHRandomInitComparison-style many-chains job on fewer GPUs; --config 4: "
                         "all chains of a target side by side, 1 = one after the other)")
    ap.add_argument("--many-chains", type=int, default=64,
                    help="extra leg after the timed region (1 GPU, 1 chain per GPU, config 1 only): aggregate rate of this many chains on the GPU "
                         "stepped through icp_chain_step_batched, reported as `many_chains` (0 = skip)")
    ap.add_argument("--fused", type=int, default=2, choices=[0, 1, 2, 3],
                    help="host<->device call pattern per step: 0 per-method calls, 1 propose + icp_chain_eval_step, 2 one icp_chain_step, "
                         "3 per-method calls over a chain bound with icp_chain_bind (the drop-in path)")
    ap.add_argument("--dropin-leg", type=int, default=1,
                    help="default run (N = 1, config 1): extra legs `dropin_per_method` — configs[1] and configs[3] stepped method by method "
                         "(Scalismo's call pattern), unbound and bound with icp_chain_bind, beside icp_chain_step (0 = skip)")
    ap.add_argument("--root-sampler-leg", type=int, default=1,
                    help="default run (N = 1, config 1): extra leg with the opt-in Cholesky-root sampler, reported as `cholesky_root_sampler` (0 = skip)")
    ap.add_argument("--sampler", type=str, default="eigen", choices=["eigen", "cholesky-root"],
                    help="posterior.sample() of the timed chain: the reference's KL basis (default, the parity path) or the opt-in Cholesky-root "
                         "sampler (NOT the reference's arithmetic: DESIGN.md §5.7; the line then carries \"sampler\": \"cholesky-root\")")
    ap.add_argument("--parallel-cpu-budget", type=float, default=5.0,
                    help="seconds of the B1_parallel leg of the CPU baseline (min(chains, cores) independent one-thread oracle chains at once; 0 = skip)")
    ap.add_argument("--events-out", type=str, default="",
                    help="also write the per-kernel HIP-event table of the roofline leg (durations WITHOUT device-side waits, the waits beside them) to this json file")
    ap.add_argument("--selftest-launcher", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--b1-child", type=int, nargs=2, default=None, help=argparse.SUPPRESS)
    ap.add_argument("--leg", type=str, default="", help=argparse.SUPPRESS)  # "config3", "config3:cholesky-root", "config4": one extra leg, in a process of its own
    args = ap.parse_args()
    if args.b1_child is not None:
        b1_child(args)
        return
    if args.leg:
        import __graft_entry__ as graft
        pkg = graft.load_package()
        name, _, sampler = args.leg.partition(":")
        if name == "config4":
            out = config4_leg(pkg, args, 0)
        elif name.startswith("dropin"):
            out = dropin_leg(pkg, args, int(name[len("dropin"):]), 0)
        elif name == "femur200":
            out = femur200_leg(pkg, args, 0)
        else:
            out = extra_config_leg(pkg, args, int(name[len("config"):]), 0, sampler=sampler or "eigen")
        print(json.dumps(out))
        return

    under_launcher = "RANK" in os.environ and "WORLD_SIZE" in os.environ
    if not under_launcher and args.gpus > 1:
        launch_ranks(args, sys.argv[1:])
        return
    if args.selftest_launcher:
        selftest_launcher_rank()
        return

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    args.gpus = world

    dist = torch = None
    if under_launcher:  # under a launcher the RCCL path is taken even with one rank
        import torch  # noqa: F811
        import torch.distributed as dist  # noqa: F811
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    import __graft_entry__ as graft
    pkg = graft.load_package()

    if args.config == 4:
        run_config4(pkg, args, dist, torch, rank, world, local_rank)
        if dist is not None:
            dist.destroy_process_group()
        return

    # ---- workload (identical on every rank; synthetic, built from the bundled femur data or procedurally)
    wl = build_workload(pkg, args.config, args.subdiv, args.fused, args)
    model, target, setup = wl["model"], wl["target"], wl["setup"]
    setup.sampler = args.sampler
    r = model.rank
    B = max(1, args.chains_per_gpu)
    ctxs = [pkg.IcpContext(model, target, device=local_rank) for _ in range(B)]  # (a context holds one chain's scratch)
    ctx = ctxs[0]
    chains = []
    for i in range(B):
        gid = rank * B + i  # chain id within the job
        chains.append(pkg.SamplingRegistration(ctxs[i], setup, wl["init"](gid), seed=1024 + gid))
    chain = chains[0]

    def run_chains(n, want_records=True):
        """n steps of every chain of this rank -> records [B * n, rec_len]"""
        if B == 1:
            return chain.run(n, want_records=want_records)
        rec = pkg.run_chains_batched(chains, n, want_records=want_records)
        return np.concatenate(rec) if want_records else None

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
    rate = world * B * args.steps / dt
    multi_gpu = rank_report(dist, torch, rank, world, local_rank, 1e3 * t_chain, 1e3 * t_gather)

    line = {
        "metric": METRIC,
        "value": rate,
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
            "workload": wl["name"] + "; %d chain%s per GPU" % (B, "" if B == 1 else "s (lockstep, icp_chain_step_batched)"),
            "baseline_config_index": args.config,
            "chains_per_gpu": B,
            "calls_per_step": {0: "per-method", 1: "propose + icp_chain_eval_step", 2: "icp_chain_step", 3: "per-method over icp_chain_bind"}[args.fused] if B == 1 else "icp_chain_step_batched",
            "chain_ms": 1e3 * t_chain,
            "log_gather_ms": 1e3 * t_gather,
            "accepted": n_acc,
            "icp_proposals": n_icp,
            "sampler": args.sampler,
        },
        "roofline": None,
        "cpu_baseline": None,
        "multi_gpu": multi_gpu,
    }

    if rank == 0 and world == 1:
        # ---- per-kernel time: HIP events on the streams the kernels run on (the library's streams) -> roofline block
        if args.profile_steps > 0:
            try:
                line["roofline"] = roofline_leg(pkg, args, wl, ctx, chains, B, rate, line)
            except Exception as e:  # the headline line is printed whatever happens in the extra legs
                line["roofline_error"] = str(e)[:300]
        # ---- CPU baselines B1 / B2 on bounded samples of the same workload
        if args.cpu_steps > 0 and B == 1:
            try:
                line["cpu_baseline"] = cpu_baseline_leg(pkg, args, wl, ctx)
            except Exception as e:
                line["cpu_baseline_error"] = str(e)[:300]
    for ch in chains:
        ch.close()
    for cx in ctxs:
        cx.close()
    if rank == 0 and world == 1 and B == 1 and args.config == 1 and args.extra_configs.strip():
        # ---- not the headline: short legs of the other single-GPU configurations, so that the driver's default run times them too
        # Each leg in a PROCESS OF ITS OWN — what `bench.py --config N` measures.  (Round 5: inside this process, behind the femur-50
        # context of the headline, the configs[3] chain ran 10 % slower than alone — 1,620 against 1,810 it/s, reproduced with any
        # rank <= 64 context made, and closed, before a rank-200 one; not the stream, pinned or device-buffer pools, not the kept models,
        # not the allocations' layout: NOTES round 5.  The legs are reports about other configurations, not about that interaction.)
        line["extra_configs"] = {"how": "every leg a child process of its own (python bench.py --leg configN), started after this process' own legs"}
        def child_leg(spec):
            cmd = [sys.executable, os.path.abspath(__file__), "--leg", spec, "--subdiv", str(args.subdiv), "--fused", str(args.fused),
                   "--face-grid", str(args.face_grid), "--face-rank", str(args.face_rank), "--cpu-steps", str(args.cpu_steps),
                   "--parallel-cpu-budget", str(args.parallel_cpu_budget)]
            done = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
            if done.returncode != 0:
                return {"error": (done.stderr or "")[-300:]}
            return json.loads(done.stdout.strip().splitlines()[-1])
        for cfg_i in [int(x) for x in args.extra_configs.split(",") if x.strip()]:
            try:
                line["extra_configs"]["config%d" % cfg_i] = child_leg("config%d" % cfg_i)
            except Exception as e:
                line["extra_configs"]["config%d" % cfg_i] = {"error": str(e)[:200]}
            if cfg_i == 3 and args.root_sampler_leg:  # … and with the opt-in Cholesky-root sampler (not the reference's arithmetic, §5.7)
                try:
                    line["extra_configs"]["config3_cholesky_root"] = child_leg("config3:cholesky-root")
                except Exception as e:
                    line["extra_configs"]["config3_cholesky_root"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and B == 1 and args.config == 1 and args.extra_configs.strip():
        try:  # the reference's largest model (rank 201): wide step + on-device loop since round 6
            line["extra_configs"]["femur200"] = child_leg("femur200")
        except Exception as e:
            line["extra_configs"]["femur200"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and B == 1 and args.config == 1 and args.dropin_leg and args.extra_configs.strip():
        # ---- not the headline: the north star's drop-in contract measured — the chain stepped method by method as Scalismo steps it
        # (unbound, and bound once with icp_chain_bind) beside icp_chain_step, configs[1] and configs[3], a child process each
        line["dropin_per_method"] = {"how": "python bench.py --leg dropinN (a process of its own per configuration); harness mode fused = 0 / 3 / 2"}
        for cfg_i in (1, 3):
            try:
                line["dropin_per_method"]["config%d" % cfg_i] = child_leg("dropin%d" % cfg_i)
            except Exception as e:
                line["dropin_per_method"]["config%d" % cfg_i] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and B == 1 and args.config == 1 and args.root_sampler_leg:
        # ---- not the headline and NOT the reference's arithmetic: the same chain with the opt-in Cholesky-root sampler
        # (icp_proposal_set_sampler) — what the accepted path costs when the reference's SVD convention for posterior.sample() is not required
        try:
            import copy
            rs = copy.copy(setup)
            rs.sampler = "cholesky-root"
            out = {"sampler": "cholesky-root", "unit": "iterations/s",
                   "note": "opt-in: z multiplies W = D L^-T (M = L L^T) instead of the KL basis V sqrt(S): same proposal distribution, same transition "
                           "density, no eigen-decomposition; a different realisation of the same Markov kernel, so NOT comparable decision for decision "
                           "with the reference (tests/test_gpu_parity.py::test_cholesky_root_sampler_*)"}
            for key, n_w, n in (("first_20_steps", 5, 20), ("value", 200, 3000)):
                rctx = pkg.IcpContext(model, target, device=local_rank)
                rch = pkg.SamplingRegistration(rctx, rs, wl["init"](0), seed=1024)
                rch.run(n_w, want_records=False)
                t1 = time.perf_counter()
                rrec = rch.run(n)
                out[key] = n / (time.perf_counter() - t1)
                out[key + "_accepted"] = int(rrec[:, 1].sum())
                out["runtime_stats"] = rctx.runtime_stats()
                rch.close(); rctx.close()
            line["cholesky_root_sampler"] = out
        except Exception as e:
            line["cholesky_root_sampler"] = {"error": str(e)[:200]}
    if rank == 0 and world == 1 and B == 1 and args.many_chains > 1 and args.config == 1:
        # ---- not the headline: the same workload with many independent chains on the one GPU (SURVEY.md §8e "within a GPU,
        # batch B chains per launch"; RunMHRandomInitComparison-style jobs), one context per chain, lockstep submissions
        try:
            nB = args.many_chains
            pkg.expect_contexts(local_rank, nB)
            mctx = [pkg.IcpContext(model, target, device=local_rank) for _ in range(nB)]
            mch = [pkg.SamplingRegistration(mctx[i], setup, wl["init"](i), seed=1024 + i) for i in range(nB)]
            pkg.run_chains_batched(mch, 40, want_records=False)
            n_m = 1000
            t1 = time.perf_counter()
            pkg.run_chains_batched(mch, n_m, want_records=False)
            mdt = time.perf_counter() - t1
            # the distance kernel with many chains per launch (north_star: "HBM GB/s on the distance kernel"): HIP events around the
            # launches of a short extra run (the first chain's context carries the group's launches)
            dk = None
            try:
                mctx[0].profile_start(max_launches=40000)
                n_p = 100
                pkg.run_chains_batched(mch, n_p, want_records=False)
                pst = mctx[0].profile_stop()
                if "k_step_filter" in pst:
                    f = pst["k_step_filter"]
                    per_launch = nB * n_p / max(f["calls"], 1)
                    dalg = kernel_algorithmic_bytes("k_step_filter", model, target, setup) * per_launch
                    dk = {"kernel": "k_step_filter (batched)", "chains_per_launch": per_launch, "avg_launch_us": f["avg_us"], "launches": f["calls"],
                          "algorithmic_bytes": dalg, "algorithmic_GBs": dalg / (f["avg_us"] * 1e-6) / 1e9,
                          "frac_hbm_algorithmic": dalg / (f["avg_us"] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                          "kernel_us_per_launch": {k: round(v["avg_us"], 2) for k, v in pst.items()},
                          "note": "algorithmic bytes per chain x chains per launch / launch duration; the chains of a launch search the SAME target, so the "
                                  "bytes that actually leave HBM are fewer (the target's spheres stay in L2 between the chains' workgroups): "
                                  "`counter_*` = FETCH_SIZE x 2 + WRITE_SIZE of the same kernel in the same regime (the newest profiles/r0N_pmc_traffic.json)"}
                    for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json"):
                        tfile = os.path.join(ROOT, "profiles", tname)
                        if not os.path.exists(tfile):
                            continue
                        ctr = json.load(open(tfile)).get("many_chains", {}).get("k_step_filter", {}).get("hbm_bytes_per_launch")
                        if ctr is not None:
                            dk["counter_source"] = "profiles/" + tname
                            dk["counter_bytes_per_launch"] = ctr
                            dk["counter_GBs"] = ctr / (f["avg_us"] * 1e-6) / 1e9
                            dk["frac_hbm_counter"] = dk["counter_GBs"] / HBM_PEAK_GBS
                            break
            except Exception as e:
                dk = {"error": str(e)[:200]}
            line["many_chains"] = {"chains_per_gpu": nB, "value": nB * n_m / mdt, "unit": "iterations/s", "steps_per_chain": n_m, "distance_kernel": dk,
                                   "entry_point": "icp_chains_run_on_device (the whole MH loop on the device: DESIGN §5.1c) from 24 chains on, "
                                                  "icp_chain_step_batched below"}
            for ch in mch:
                ch.close()
            for cx in mctx:
                cx.close()
        except Exception as e:
            line["many_chains"] = {"error": str(e)[:200]}
    if rank == 0:
        # fall-back counters of the whole process (icp_ctx_runtime_stats): every device-side wait that timed out, every step done twice
        line["runtime_stats"] = pkg._native.runtime_stats()
        print(json.dumps(line))
    if dist is not None:
        dist.destroy_process_group()


def _tracked_json(names):
    """the newest tracked profile file of a family (profiles/r06_…, r05_…): (path relative to the repository, content) or (None, {})"""
    for n in names:
        f = os.path.join(ROOT, "profiles", n)
        if os.path.exists(f):
            return "profiles/" + n, json.load(open(f))
    return None, {}


def config4_roofline(pkg, args, model, target, make_setup, device, n_chains=25, n_steps=240):
    """The roofline / MFMA block of configs[4]'s steady state: ONE submission of `n_chains` chains of one target stepped inside the
    on-device loop (what sharding.run_batch submits: 4 x 25 for the 10 x 10 job), its launches under HIP events on the launch streams.
    Event ids are the step's stages: k_posterior_eigen = one launch sequence of the tridiagonal route for up to 16 posteriors
    (k_assemble_many -> k_tridiag_many -> k_tri_solve_many -> k_tri_back_many -> 3 x k_tri_gemm_many + correction), k_instance =
    k_wide_instance<G>, k_step_regression = k_wide_xrows + k_wide_regression_fold, k_step_filter / k_step_resolve = the proposal's searches,
    k_surface_* / k_vertex_* = the evaluator's.  Counter figures (HBM bytes, MFMA-busy, waves per SIMD) are those of the SAME regime's
    rocprofv3 --pmc passes (profiles/r06_pmc_*.json, regime wide_loop25: tools/r6_profiles.sh) — tracked files, not measured in this run."""
    setup = make_setup(model, target)
    r, N = model.rank, model.n_points
    pkg.expect_contexts(device, n_chains)
    ctxs = [pkg.IcpContext(model, target, device=device) for _ in range(n_chains)]
    chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(n_chains)]
    try:
        pkg.run_chains_batched(chains, 30, want_records=False)
        ctxs[0].profile_start(max_launches=200 * n_steps + 4096)
        t1 = time.perf_counter()
        pkg.run_chains_batched(chains, n_steps, want_records=False)
        pdt = time.perf_counter() - t1
        raw = ctxs[0].profile_stop()
        paths = ctxs[0].step_paths()
    finally:
        for c in chains:
            c.close()
        for c in ctxs:
            c.close()
    pst = {k: v for k, v in raw.items() if not k.startswith("count.") and not k.endswith(".device_wait")}
    if not pst:
        return None, None
    dominant = max(pst, key=lambda k: pst[k]["total_ms"])
    k = pst[dominant]
    per_launch = n_chains * n_steps / max(k["calls"], 1)   # posteriors (chains) a launch of the dominant stage carries
    alg1 = kernel_algorithmic_bytes(dominant, model, target, setup)
    has_boundary = bool(pkg.data.boundary_vertex_flags(target).any())
    bytes_step, _ = algorithmic_step(model, target, setup, has_boundary)
    lat = latency_floor_model(model, setup, 0.55, 0.7, bytes_step, 0.0)
    lat["measured_us_per_round"] = 1e6 * pdt / n_steps
    lat["note"] = "floor of ONE chain's step; a round steps %d chains side by side in one launch sequence" % n_chains
    lat["frac"] = lat["floor_us_per_step"] / lat["measured_us_per_round"]
    tsrc, traffic = _tracked_json(("r06_pmc_traffic.json",))
    ssrc, sq = _tracked_json(("r06_pmc_sq.json",))
    traffic, sq = traffic.get("wide_loop25", {}), sq.get("wide_loop25", {})
    roof = {"bound": "latency" if dominant.startswith(SINGLE_WORKGROUP) else "hbm", "kernel": dominant, "avg_launch_us": k["avg_us"], "launches": k["calls"],
            "chains_per_launch": per_launch, "algorithmic_bytes": None if alg1 is None else alg1 * per_launch, "achieved": None, "peak": HBM_PEAK_GBS,
            "unit": "GB/s", "frac": None, "traffic": None, "latency": lat,
            "sample": "%d chains of one target in one submission per step inside icp_chains_run_on_device (step_paths.device_loop = %d), ONE call of %d "
                      "steps incl. its set-up (claim, capture, uploads: ~10 ms): %.0f it/s, %.2f ms per round"
                      % (n_chains, paths.get("device_loop", 0), n_steps, n_chains * n_steps / pdt, 1e3 * pdt / n_steps),
            "value_in_this_stretch": n_chains * n_steps / pdt,
            "whole_step": {"algorithmic_bytes_per_chain_step": bytes_step, "hbm_frac": bytes_step * (n_chains * n_steps / pdt) / (HBM_PEAK_GBS * 1e9)},
            "kernel_us_per_round": {name: round(v["total_ms"] * 1e3 / n_steps, 2) for name, v in pst.items()}}
    if alg1 is not None:
        roof["achieved"] = alg1 * per_launch / (k["avg_us"] * 1e-6) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    if dominant.startswith("k_posterior_eigen") and traffic:
        seq = ("k_assemble_many", "k_tridiag_many", "k_tri_solve_many", "k_tri_back_many", "k_tri_correction_many")
        t = sum(traffic[q]["hbm_bytes_per_launch"] for q in seq if q in traffic) + 3 * traffic.get("k_tri_gemm_many", {}).get("hbm_bytes_per_launch", 0)
        roof["traffic"] = t
        roof["traffic_source"] = tsrc + " (wide_loop25: the sequence's launches summed, 16 posteriors a launch)"
    # the step's HBM-bound kernels against the roofline (durations of THIS run, counter bytes of the tracked pass)
    groups = -(-n_chains // 8)
    gsize = -(-n_chains // groups)
    hbm = {}
    if "k_instance" in pst:
        alg = 3 * N * r * 8 + n_chains * 3 * N * 8 * 4   # the basis ONCE (its groups of chains share an XCD's L2); reference, mean in, instance + kept deformation out
        us = pst["k_instance"]["avg_us"]
        inst_name = "k_wide_instance<%d>" % min(max(gsize, 5), 8)
        row = {"kernel": inst_name, "avg_launch_us": us, "algorithmic_bytes": alg, "achieved_GBs": alg / (us * 1e-6) / 1e9,
               "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
               "note": "%d groups of <= %d chains, each a pass over the 137 MB basis — the groups of one point block on ONE XCD (round 6), so the passes "
                       "behind the first are L2 hits; the launch is then bound by its waves' unfused f64 multiply-adds (6 per chain and basis column)" % (groups, gsize)}
        c = traffic.get(inst_name, {}).get("hbm_bytes_per_launch")
        if c:
            row.update(counter_bytes_per_launch=c, counter_GBs=c / (us * 1e-6) / 1e9, counter_frac=c / (us * 1e-6) / 1e9 / HBM_PEAK_GBS)
        hbm["k_wide_instance"] = row
    if "k_step_regression" in pst:
        Km = max([p.get("n_model_ids", 0) for p in setup.icp] + [0])
        alg = n_chains * (Km * 4 * 16 * ((r + 1 + 15) // 16) * 8 + (r + 1) * (r + 1) * 8 * 0.5)   # operand rows in, ONE partial (lower triangle) out
        us = pst["k_step_regression"]["avg_us"]
        row = {"kernel": "k_wide_regression_fold (+ k_wide_xrows)", "avg_launch_us": us, "algorithmic_bytes": alg, "achieved_GBs": alg / (us * 1e-6) / 1e9,
               "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        c = traffic.get("k_wide_regression_fold", {}).get("hbm_bytes_per_launch")
        if c:
            row.update(counter_bytes_per_launch=c, counter_GBs=c / (us * 1e-6) / 1e9, counter_frac=c / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                       note="counter bytes = %.1f x algorithmic (round 6: the launch's (chain, block) pairs are dealt out XCD by XCD, a posterior's operand "
                            "rows come from ONE L2 after the first touch; with blockIdx.y = chain every XCD fetched them again: 317 MB per launch, 4.5 x)" % (c / alg))
        hbm["k_wide_regression_fold"] = row
    roof["hbm_kernels"] = hbm
    if sq:
        roof["occupancy"] = {q: {"avg_waves_per_simd": round(v.get("avg_waves_per_simd", 0.0), 3), "parked_share": round(v.get("parked_share", 0.0), 3),
                                 "lds_conflict_share": round(v.get("lds_conflict_share", 0.0), 3), "median_us_alone": v.get("median_us")}
                             for q, v in sq.items() if q in ("k_tridiag_many", "k_tri_solve_many", "k_tri_back_many", "k_tri_gemm_many", "k_wide_regression_fold",
                                                             "k_wide_instance<8>", "k_wide_instance<7>", "k_wide_filter", "k_wide_resolve", "k_posterior_factor_tiles", "k_wide_xrows")}
        roof["occupancy_source"] = ssrc + " (one kernel at a time: durations WITHOUT neighbours)"
    # MFMA utilisation of the projection where it IS GEMM-shaped: the folded regression of a submission's posteriors
    mfma = None
    if "k_step_regression" in pst:
        flops = n_chains * projection_flops(model, setup)
        us = pst["k_step_regression"]["avg_us"]
        msrc, mf = _tracked_json(("r06_pmc_mfma.json",))
        mf = mf.get("wide_loop25", {})
        mfma = {"peak_TFLOPs": F64_MATRIX_TFLOPS, "measured_instruction_rate": F64_MATRIX_MEASURED, "kernels": {}, "counter_source": (msrc + " (wide_loop25)") if msrc else None,
                "formula": "busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (%d SIMDs x GRBM_GUI_ACTIVE / %d XCDs); flops_frac = algorithmic f64 flops per launch / avg "
                           "launch duration (HIP events, this run) / %.1f TFLOP/s" % (N_SIMD, N_XCD, F64_MATRIX_TFLOPS)}
        row = {"avg_launch_us": us, "posteriors_per_launch": n_chains, "algorithmic_f64_flops": flops, "achieved_TFLOPs": flops / (us * 1e-6) / 1e12,
               "flops_frac": flops / (us * 1e-6) / 1e12 / F64_MATRIX_TFLOPS,
               "note": "event id k_step_regression = k_wide_xrows + k_wide_regression_fold of the submission"}
        c = mf.get("k_wide_regression_fold", {})
        if c.get("busy_frac_median") is not None:
            row.update(busy_frac=c["busy_frac_median"], SQ_VALU_MFMA_BUSY_CYCLES=c["SQ_VALU_MFMA_BUSY_CYCLES"]["median"], GRBM_GUI_ACTIVE=c["GRBM_GUI_ACTIVE"]["median"],
                       counter_pass_us=c.get("median_us"))
        mfma["kernels"]["k_wide_regression_fold"] = row
        for q in ("k_tri_back_many", "k_tri_gemm_many"):
            c = mf.get(q, {})
            if c.get("busy_frac_median") is not None:
                fl = (16 * 2.0 * r ** 3) if q == "k_tri_back_many" else 2.0 * r ** 3   # (back-transformation of 16 problems / one r x r x r product)
                mfma["kernels"][q] = {"busy_frac": c["busy_frac_median"], "counter_pass_us": c.get("median_us"),
                                      "flops_frac_in_the_counter_pass": None if not c.get("median_us") else fl / (c["median_us"] * 1e-6) / 1e12 / F64_MATRIX_TFLOPS}
    return roof, mfma



def config4_leg(pkg, args, device):
    """configs[4] with short chains: 10 targets x 10 random-init chains through sharding.run_batch on this GPU, chains of 50 steps (the
    size round 3's 1,320-1,450 it/s was quoted on) and of 300; whole job each (contexts, chains, steps, records) — what `--config 4
    --steps 50|300` times.  (`--config 4 --steps 3000`: the reference-sized chains, 30 s.)"""
    model = face_model(pkg, args)
    targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(10)]
    make_setup = lambda m, t: pkg.bfm_fitting_partial(m, t, evaluator="collective", fused=args.fused)
    pkg.expect_contexts(device, 25)  # (beside the warm-up item, as in run_config4: the streams of a submission's contexts)
    pkg.sharding.run_batch(pkg, model, targets[:1], n_chains=1, n_steps=5, make_setup=make_setup, dist=None, device_index=device)
    out = {"unit": "iterations/s", "targets": 10, "chains": 10,
           "workload": "BASELINE.json configs[4] with short chains: 10 targets x 10 random-init chains on the BFM-sized stand-in (N=%d, rank %d; 0.4 pose + "
                       "0.55 ICP + 0.05 random walk, collective boundary-aware evaluator), whole job on one GPU" % (model.n_points, model.rank)}
    for n_steps in (50, 300):
        t0 = time.perf_counter()
        items, recs, stats = pkg.sharding.run_batch(pkg, model, targets, n_chains=10, n_steps=n_steps, make_setup=make_setup, dist=None,
                                                    device_index=device, return_stats=True)
        dt = time.perf_counter() - t0
        out["steps_%d" % n_steps] = {"value": len(items) * n_steps / dt, "job_s": dt, "items": len(items), "accepted": int(sum(r[:, 1].sum() for r in recs)),
                                     "contexts_built": int(stats["contexts_built"]), "phase_ms": stats.get("phase_ms")}
    out["value"] = out["steps_300"]["value"]
    try:  # the steady state of one submission under HIP events + the tracked counter passes of the same regime
        out["roofline"], out["mfma"] = config4_roofline(pkg, args, model, targets[0], make_setup, device)
    except Exception as e:
        out["roofline_error"] = str(e)[:300]
    out["runtime_stats"] = pkg._native.runtime_stats()
    out["step_paths_process"] = pkg._native.step_paths() if hasattr(pkg._native, "step_paths") else None
    if args.parallel_cpu_budget > 0 and args.cpu_steps > 0:
        # the batch job on the host cores, the reference's way: one single-threaded B1 chain per work item, as many at a time as there
        # are cores (bounded: 32 processes of ~0.5 GB each)
        try:
            out["cpu_baseline"] = dict(b1_parallel(args, 4, 100, args.parallel_cpu_budget, max_procs=32), kind="port")
        except Exception as e:
            out["cpu_baseline"] = {"error": str(e)[:300]}
    return out


def extra_config_leg(pkg, args, cfg_i, device, sampler="eigen"):
    """A short chain of another configuration (2: 400 steps after 40 of warm-up; 3: 600 after 100 — the face chain's first two hundred
    steps accept three ICP proposals in four, each of which waits for the basis of the one before, and say little about the 10,000-step
    chains of the reference's apps) on the same GPU."""
    wl = build_workload(pkg, cfg_i, args.subdiv, args.fused, args)
    wl["setup"].sampler = sampler
    ctx = pkg.IcpContext(wl["model"], wl["target"], device=device)
    chain = pkg.SamplingRegistration(ctx, wl["setup"], wl["init"](0), seed=1024)
    n_w, n = (40, 400) if cfg_i == 2 else (100, 600)
    chain.run(n_w, want_records=False)
    t0 = time.perf_counter()
    rec = chain.run(n)
    dt = time.perf_counter() - t0
    out = {"value": n / dt, "unit": "iterations/s", "steps": n, "warmup": n_w, "ms_per_step": 1e3 * dt / n, "accepted": int(rec[:, 1].sum()),
           "icp_proposals": int((rec[:, 2] < 2).sum()), "workload": wl["name"], "runtime_stats": ctx.runtime_stats(), "sampler": sampler}
    try:
        out["roofline"] = leg_roofline(pkg, ctx, chain, wl, 100 if cfg_i == 3 else 60, out["value"], out["accepted"] / n, out["icp_proposals"] / n)
    except Exception as e:
        out["roofline_error"] = str(e)[:200]
    out["step_paths"] = ctx.step_paths()
    if cfg_i == 2 and out["accepted"] == 0:
        out["note"] = ("no step accepted: with all 1,622 model points as correspondences the ICP posterior is so narrow that the reference's own "
                       "transition ratio rejects (almost) every proposal — GPU and oracle agree on every decision "
                       "(tests/test_gpu_chain.py::test_femur100_all_points_symmetric_58k_target_matches_oracle), but the oracle is not pinned to "
                       "Scalismo (DESIGN §2), so whether the reference's chain moves here cannot be adjudicated; the rate is that of the REJECTED path")
    chain.close()
    if cfg_i == 2:
        # the same chain started where the reference's experiments start it — next to the posterior mean, from the deterministic ICP fit
        # (apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:148): there about one proposal in ten is accepted, and the accepted
        # rank-101 step (decomposition of the new posterior on the tridiagonal route, then the next proposal) is timed
        try:
            n_pts = wl["model"].n_points
            fit = pkg.IcpBasedSurfaceFitting(ctx, 1.0, "ModelSampling", modelPointIds=np.arange(n_pts, dtype=np.int32)).runfitting(
                10, initialModelParameters=wl["init"](3))
            ch2 = pkg.SamplingRegistration(ctx, wl["setup"], fit, seed=1024)
            ch2.run(n_w, want_records=False)
            t1 = time.perf_counter()
            rec2 = ch2.run(n)
            dt2 = time.perf_counter() - t1
            out["from_deterministic_fit"] = {"value": n / dt2, "unit": "iterations/s", "steps": n, "ms_per_step": 1e3 * dt2 / n,
                                             "accepted": int(rec2[:, 1].sum()),
                                             "note": "start = IcpBasedSurfaceFitting(all model points, 10 iterations x 3 noise levels) from the random shape 3"}
            ch2.close()
            # the configuration's headline is the chain that MOVES (verdict r05 #11): the random start's rate is that of the rejected path
            out["from_random_start"] = {k: out[k] for k in ("value", "ms_per_step", "accepted", "steps")}
            out["value"], out["ms_per_step"], out["accepted"] = n / dt2, 1e3 * dt2 / n, int(rec2[:, 1].sum())
            out["value_is"] = "from_deterministic_fit (the random start, every proposal rejected: from_random_start)"

        except Exception as e:
            out["from_deterministic_fit"] = {"error": str(e)[:200]}
    ctx.close()
    return out


def femur200_leg(pkg, args, device):
    """The reference's largest bundled model, femur_gp_model_200-components.h5 — rank 201 (apps/femur/CreateGPModel.scala:93) — through the
    femur mixture of apps/femur/IcpProposalRegistration.scala:59-85 against the bundled target: one chain (the wide step: ranks above 116)
    and five chains inside the on-device loop (RunMHRandomInitComparison.scala:66's five).  Until round 6 rank 201 was one past the
    tridiagonal route and every posterior took the per-stage generic decomposition."""
    model, target = pkg.data.load_femur_model_and_target(200)
    setup = pkg.femur_icp_proposal_registration(model, target, fused=args.fused if args.fused in (2, 3) else 2)
    out = {"unit": "iterations/s", "workload": "femur 200-component GPMM (N=%d, rank %d) vs the bundled aligned target (M=%d); 0.9 ICP(Target+Model sampling, "
                                               "K=%d) + 0.1 random walk; prior x independent Gaussian(0,2) on %d points"
                                               % (model.n_points, model.rank, target.n_points, 2 * model.rank, 4 * model.rank)}
    ctx = pkg.IcpContext(model, target, device=device)
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    chain.run(40, want_records=False)
    t0 = time.perf_counter()
    rec = chain.run(300)
    dt = time.perf_counter() - t0
    out["one_chain"] = {"value": 300 / dt, "steps": 300, "warmup": 40, "accepted": int(rec[:, 1].sum()), "step_paths": ctx.step_paths()}
    chain.close()
    ctx.close()
    B = 5
    ctxs = [pkg.IcpContext(model, target, device=device) for _ in range(B)]
    chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
    pkg.run_chains_batched(chains, 30, want_records=False)
    t0 = time.perf_counter()
    recs = pkg.run_chains_batched(chains, 200)
    dt = time.perf_counter() - t0
    out["five_chains"] = {"value": B * 200 / dt, "steps_per_chain": 200, "accepted": int(sum(r[:, 1].sum() for r in recs)), "step_paths": ctxs[0].step_paths()}
    out["value"] = out["one_chain"]["value"]
    out["runtime_stats"] = pkg._native.runtime_stats()
    for c in chains:
        c.close()
    for c in ctxs:
        c.close()
    return out


def dropin_leg(pkg, args, cfg_i, device):
    """The drop-in contract measured: the chain stepped the way Scalismo's MetropolisHastings.next steps it — logValue(current), propose,
    logValue(proposal), and through the mixture every ICP proposal's logTransitionProbability both ways, ONE native call each
    (api/sampling/SamplingRegistration.scala:52-58, MixedProposalDistributions.scala:48-68) — (a) as is (`per_method`: every call a
    device round trip), (b) over a chain bound once with icp_chain_bind (`per_method_bound`: the first call of a step submits the whole
    step), beside (c) the harness' own icp_chain_step loop (`icp_chain_step`: the headline's mode, with its next-step pre-launch, which
    needs the next step's random numbers ahead — something a Scalismo caller cannot give).  Same chain, same seed, same window as the
    driver's line (20 steps after 5) and a longer stretch; every mode a fresh context."""
    out = {"unit": "iterations/s", "config": cfg_i}
    wl = None
    for key, fused in (("per_method", 0), ("per_method_bound", 3), ("icp_chain_step", 2)):
        wl = build_workload(pkg, cfg_i, args.subdiv, fused, args)
        n_long = 1000 if cfg_i == 1 else 300
        leg = {}
        for win, n_w, n in (("driver_window", 5, 20), ("long", 100, n_long)):
            ctx = pkg.IcpContext(wl["model"], wl["target"], device=device)
            chain = pkg.SamplingRegistration(ctx, wl["setup"], wl["init"](0), seed=1024)
            chain.run(n_w, want_records=False)
            c0 = chain.native_calls()
            t0 = time.perf_counter()
            rec = chain.run(n)
            dt = time.perf_counter() - t0
            c1 = chain.native_calls()
            leg[win] = {"value": n / dt, "steps": n, "warmup": n_w, "accepted": int(rec[:, 1].sum()),
                        "native_per_method_calls_per_step": (c1["proposal_calls"] + c1["log_value_calls"] - c0["proposal_calls"] - c0["log_value_calls"]) / n}
            if fused == 3:
                leg[win]["whole_steps_submitted_per_step"] = (c1["bound_steps_from_propose"] + c1["bound_steps_from_log_value"]
                                                              - c0["bound_steps_from_propose"] - c0["bound_steps_from_log_value"]) / n
                leg[win]["parked_densities_per_step"] = (c1["parked_transition_hits"] - c0["parked_transition_hits"]) / n
            leg["step_paths"] = ctx.step_paths()
            leg["runtime_stats"] = ctx.runtime_stats()
            chain.close()
            ctx.close()
        out[key] = leg
    out["workload"] = wl["name"]
    for win in ("driver_window", "long"):
        out["bound_over_chain_step_" + win] = out["per_method_bound"][win]["value"] / out["icp_chain_step"][win]["value"]
        out["bound_over_unbound_" + win] = out["per_method_bound"][win]["value"] / out["per_method"][win]["value"]
    return out


def leg_roofline(pkg, ctx, chain, wl, n_prof, rate, accepted_share, icp_share):
    """The roofline block of a short leg: the time-dominant kernel of a profiled stretch of the same chain (HIP events on the launch
    streams), its algorithmic bytes per launch against its average duration, and the latency floor of the step's launch structure."""
    model, target, setup = wl["model"], wl["target"], wl["setup"]
    ctx.profile_start(max_launches=160 * n_prof + 4096)
    chain.run(n_prof, want_records=False)
    raw = ctx.profile_stop()
    stats = {k: dict(v) for k, v in raw.items() if not k.startswith("count.") and not k.endswith(".device_wait")}
    for name, v in stats.items():  # (a launch's time WITHOUT what it waits on the device for another stream's word)
        w = raw.get(name + ".device_wait")
        if w is not None:
            v["total_ms"] = max(v["total_ms"] - w["total_ms"], 0.0)
            v["avg_us"] = 1e3 * v["total_ms"] / max(v["calls"], 1)
    if not stats:
        return None
    dominant = max(stats, key=lambda k: stats[k]["total_ms"])
    k = stats[dominant]
    alg = kernel_algorithmic_bytes(dominant, model, target, setup)
    one_cu = dominant.startswith(SINGLE_WORKGROUP)
    has_boundary = bool(pkg.data.boundary_vertex_flags(target).any())
    bytes_step, flops_step = algorithmic_step(model, target, setup, has_boundary)
    # (the stream-time term from the BYTES only: the two-level filter executes a small fraction of the brute-force pair count, so the
    # brute-force flop figure at the vector peak — 1.2 ms for the full-mesh Hausdorff distance — is no floor of anything)
    lat = latency_floor_model(model, setup, icp_share, accepted_share, bytes_step, 0.0)
    lat["measured_us_per_step"] = 1e6 / rate
    lat["frac"] = lat["floor_us_per_step"] / lat["measured_us_per_step"]
    roof = {"bound": "latency" if one_cu else "hbm", "kernel": dominant, "avg_launch_us": k["avg_us"], "launches": k["calls"],
            "algorithmic_bytes": alg, "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "latency": lat, "whole_step": {"algorithmic_bytes_per_step": bytes_step, "hbm_frac": bytes_step * rate / (HBM_PEAK_GBS * 1e9)},
            "kernel_us_per_step": {name: round(v["total_ms"] * 1e3 / n_prof, 2) for name, v in stats.items()}}
    if alg is not None:
        roof["achieved"] = alg / (k["avg_us"] * 1e-6) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    reg = "k_step_regression"
    if reg in stats:
        flops = {reg: projection_flops(model, setup), "k_tri_gemm": 2.0 * model.rank ** 3}
        roof["mfma"] = mfma_block("config%d" % wl["config"], [reg] + (["k_tri_gemm"] if model.rank > 64 else []), {reg: stats[reg]["avg_us"]}, flops)
    for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile):
            per_kernel = json.load(open(tfile)).get("config%d" % wl["config"], {})
            t = per_kernel.get(dominant, {}).get("hbm_bytes_per_launch")
            if t is None and dominant == "k_posterior_eigen" and "k_tridiag" in per_kernel:
                t = sum(per_kernel[q]["hbm_bytes_per_launch"] * (3 if q == "k_tri_gemm" else 1) for q in ("k_tridiag", "k_tri_solve", "k_tri_gemm") if q in per_kernel)
            if t is not None:
                roof["traffic"] = t
                roof["traffic_source"] = "profiles/" + tname
                break
    return roof


def run_config4(pkg, args, dist, torch, rank, world, local_rank):
    """BASELINE.json configs[4]: --targets x --chains (target, random initial shape) work items, target-major over the ranks, every
    chain --steps long; the timed region is the whole job: contexts, chains, ONE ragged gather of the records."""
    model = face_model(pkg, args)
    targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(args.targets)]
    make_setup = lambda m, t: pkg.bfm_fitting_partial(m, t, evaluator="collective", fused=args.fused)
    # chains of one target side by side (the wide step): all of them unless --chains-per-gpu says otherwise (1 = one after the other)
    cpl = args.chains_per_gpu if args.chains_per_gpu > 0 else 0  # (0: sharding.run_batch's default — three targets' chains per submission)
    # warm-up: one item per rank (builds the communicator, pages the kernels in) — and, as a host that knows it is about to make a
    # submission's worth of contexts would at its start, the hint that lets their streams be made meanwhile (icp_ctx_expect; until
    # round 6 the first keyed context of a model did that unasked, i.e. also here)
    pkg.expect_contexts(local_rank, 25)
    pkg.sharding.run_batch(pkg, model, targets[:1], n_chains=world, n_steps=max(1, args.warmup), make_setup=make_setup, dist=dist,
                           device_index=local_rank)
    if dist is not None:
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    items, recs, stats = pkg.sharding.run_batch(pkg, model, targets, n_chains=args.chains, n_steps=args.steps, make_setup=make_setup, dist=dist,
                                                device_index=local_rank, chains_per_launch=cpl, return_stats=True)
    if dist is not None:
        torch.cuda.synchronize(); dist.barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        st = torch.tensor([stats["contexts_built"], stats["items"], stats["gather_ms"], stats["chain_ms"]], dtype=torch.float64, device="cuda")
        allst = [torch.empty_like(st) for _ in range(world)]
        dist.all_gather(allst, st)
        per_rank = [[float(v) for v in a.cpu()] for a in allst]
    else:
        per_rank = [[stats["contexts_built"], stats["items"], stats["gather_ms"], stats["chain_ms"]]]
    n_items = len(items)
    assert all(r is not None and r.shape == (args.steps, 14 + model.rank) for r in recs), "a work item's records are missing"
    multi_gpu = rank_report(dist, torch, rank, world, local_rank, stats["chain_ms"], stats["gather_ms"])
    roofline = None
    if rank == 0 and world == 1:
        # ---- outside the timed region: one submission's steady state under HIP events (config4_roofline)
        try:
            roofline, mfma4 = config4_roofline(pkg, args, model, targets[0], make_setup, local_rank)
            if roofline is not None and mfma4 is not None:
                roofline["mfma"] = mfma4
        except Exception as e:  # (the leg is a report, not the measurement)
            roofline = {"error": str(e)[:300]}
    if rank == 0:
        best = max(range(n_items), key=lambda k: recs[k][:, 3].max())
        line = {"metric": METRIC, "value": n_items * args.steps / dt, "unit": "iterations/s", "n_gpus": world, "steps": args.steps,
                "warmup": args.warmup, "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
                "dtype": "f64", "data": "synthetic",
                "config": {"workload": "BASELINE.json configs[4]: batch registration, %d targets x %d random-init chains of %d steps on the "
                                       "BFM-sized stand-in (N=%d, rank %d; 0.4 pose + 0.55 ICP + 0.05 random walk, collective boundary-aware "
                                       "evaluator); work items target-major over %d rank(s), one ragged log gather"
                                       % (args.targets, args.chains, args.steps, model.n_points, model.rank, world),
                           "baseline_config_index": 4, "items": n_items, "items_per_s": n_items / dt, "job_s": dt,
                           "items_per_rank": [int(p[1]) for p in per_rank], "contexts_built_per_rank": [int(p[0]) for p in per_rank],
                           "gather_ms_per_rank": [round(p[2], 3) for p in per_rank], "chain_ms_per_rank": [round(p[3], 1) for p in per_rank],
                           "chains_per_launch": cpl if cpl > 0 else max(1, min(32, (3 if args.steps >= 150 else 2 if args.steps >= 30 else 1) * args.chains)), "best_item": [int(v) for v in items[best]],
                           "accepted": int(sum(r[:, 1].sum() for r in recs)), "phase_ms_rank0": stats.get("phase_ms")},
                "roofline": roofline, "cpu_baseline": None, "multi_gpu": multi_gpu, "runtime_stats": pkg._native.runtime_stats()}
        print(json.dumps(line))


LAUNCH_BOUNDARY_US = 2.4   # dependent launch on one stream, measured (tools/ubench2; DESIGN §6 "measured building blocks")
SHADER_GHZ = 2.4           # /opt/skills/guides/MI355X_MICROARCH.md
# one workgroup on one CU: kernels whose duration is a chain of dependent steps, whatever the bytes they move
SINGLE_WORKGROUP = ("k_posterior_eigen", "k_step_finish", "k_posterior_factor", "k_tridiag", "k_tri_solve", "k_transition_tail", "k_propose")


def latency_floor_model(model, setup, icp_share, accept_share, bytes_step, flops_step):
    """What bounds one chain's step when its kernels are far below both rooflines: dependent launch boundaries plus, per kernel,
    the larger of its stream time and — for the one-workgroup iterations — its chain of dependent steps.  Per MH step, in µs:
      rejected : launches 1-4 of the successor (begin, filter, resolve, regression) — 4 boundaries; launch 5 runs beside them
      accepted : + the KL basis of the new state, which the next proposal draws from: one boundary + the iteration's dependent
                 chain: sweeps x (n - 1) rounds x (barrier ~50 clk + LDS round trip ~77 clk + the rotation's ~14 dependent f64
                 operations at 6-7.5 clk) (ranks <= 64); Householder: (n - 2) steps x (barrier + LDS round trip + wave reduction
                 + the step's multiply-adds on the one workgroup that holds the matrix in registers) (ranks > 64)
    plus the step's algorithmic bytes and flops at the peaks (sub-microsecond at femur size)."""
    r = model.rank
    stream_us = max(bytes_step / (HBM_PEAK_GBS * 1e9), flops_step / (F32_VECTOR_TFLOPS * 1e12)) * 1e6
    n_launch = 4 if len(setup.icp) else 2
    rejected = n_launch * LAUNCH_BOUNDARY_US + stream_us
    if r <= 64:
        round_clk = 50 + 77 + 14 * 7
        eig = 2 * (((r + 1) & ~1) - 1) * round_clk / (SHADER_GHZ * 1e3)
    else:
        # the matrix lives in the registers of ONE workgroup (8 waves, two per SIMD): ceil(r/64)·ceil(r/8) entries per lane, three
        # f64 multiply-adds each per step (rank-2 update + the next product) at 4 clk per wave instruction, two waves sharing a SIMD,
        # the live part of the matrix shrinking to nothing over the steps (a third on average)
        entries = ((r + 63) // 64) * ((r + 7) // 8)
        step_clk = 50 + 77 + 60 + entries * 3 * 4 * 2 / 3.0
        eig = (r - 2) * step_clk / (SHADER_GHZ * 1e3) + LAUNCH_BOUNDARY_US  # (reduction + the solve launch behind it)
    accepted = rejected + LAUNCH_BOUNDARY_US + eig
    a = accept_share * icp_share   # share of steps whose accepted proposal needs a new KL basis
    return {"rejected_step_us": rejected, "accepted_step_us": accepted, "eigen_chain_us": eig,
            "floor_us_per_step": (1 - a) * rejected + a * accepted,
            "model": "dependent launches x %.1f us + one-workgroup dependent chains + stream time at the peaks (bench.py: latency_floor_model)" % LAUNCH_BOUNDARY_US}


def roofline_leg(pkg, args, wl, ctx, chains, B, rate, line):
    model, target, setup = wl["model"], wl["target"], wl["setup"]
    ctx.profile_start(max_launches=80 * args.profile_steps + 2048)   # (a batch's launches go out on the first chain's context)
    if B == 1:
        chains[0].run(args.profile_steps, want_records=False)
    else:
        pkg.run_chains_batched(chains, args.profile_steps, want_records=False)
    stats = ctx.profile_stop()
    # a short leg of its own for the executed-test counts (the counting atomics slow the filter launches down: not for timing)
    n_count = min(60, args.profile_steps)
    ctx.profile_start(max_launches=80 * n_count + 2048, count_searches=True)
    if B == 1:
        chains[0].run(n_count, want_records=False)
    else:
        pkg.run_chains_batched(chains, n_count, want_records=False)
    cstats = ctx.profile_stop()
    counts = {k[len("count."):]: int(v["calls"]) for k, v in cstats.items() if k.startswith("count.")}
    stats = {k: v for k, v in stats.items() if not k.startswith("count.")}
    waits = {"k_step_begin": stats.pop("k_step_begin.device_wait", None), "k_posterior_eigen": stats.pop("k_posterior_eigen.device_wait", None)}
    busy = {}
    for name, s in stats.items():
        t = s["total_ms"]
        w = waits.get(name)
        if w is not None:  # its launches include the time they WAIT on the device for another stream's word
            t = max(t - w["total_ms"], 0.0)
        busy[name] = t
    if not busy:
        return None
    dominant = max(busy, key=busy.get)
    k = stats[dominant]
    avg_us = 1e3 * busy[dominant] / max(k["calls"], 1)
    chains_per_launch = 1.0
    alg = kernel_algorithmic_bytes(dominant, model, target, setup)
    if B > 1:
        chains_per_launch = B * args.profile_steps / max(k["calls"], 1)
        alg = alg * chains_per_launch if alg is not None else None
    traffic = traffic_source = None
    for tname in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json"):
        tfile = os.path.join(ROOT, "profiles", tname)
        if os.path.exists(tfile) and B == 1:
            per_kernel = json.load(open(tfile)).get("config%d" % args.config, {})
            traffic = per_kernel.get(dominant, {}).get("hbm_bytes_per_launch")
            if traffic is None and dominant == "k_posterior_eigen" and "k_tridiag" in per_kernel:
                # ranks above 64: "k_posterior_eigen" of the event timing is the tridiagonal route's launch sequence
                traffic = sum(per_kernel[k]["hbm_bytes_per_launch"] * (3 if k == "k_tri_gemm" else 1) for k in ("k_tridiag", "k_tri_solve", "k_tri_gemm") if k in per_kernel)
            if traffic is not None:
                traffic_source = ("profiles/" + tname + ": 2 x FETCH_SIZE + WRITE_SIZE of a separate rocprofv3 --pmc pass of this command, a TRACKED FILE — "
                                  "not measured in this run (counter passes cannot share a run with the timing)")
                break
    has_boundary = bool(pkg.data.boundary_vertex_flags(target).any())
    bytes_step, flops_step = algorithmic_step(model, target, setup, has_boundary)
    icp_share = line["config"]["icp_proposals"] / max(args.steps * B, 1)
    accept_share = line["config"]["accepted"] / max(args.steps * B, 1)
    one_cu = dominant.startswith(SINGLE_WORKGROUP)
    roof = {"bound": "latency" if one_cu else "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": traffic,
            "traffic_source": traffic_source, "kernel": dominant, "avg_launch_us": avg_us, "launches": k["calls"], "algorithmic_bytes": alg,
            "chains_per_launch": chains_per_launch,
            "selection": "time-dominant kernel of this run: largest sum of launch durations (HIP events on the launch streams), "
                         "WITHOUT the time a launch waits on the device for another stream's word (k_step_begin: the previous step's "
                         "finish launch / the decomposition it draws from; k_posterior_eigen: its input, when started ahead)"}
    if alg is not None:
        roof["achieved"] = alg / (avg_us * 1e-6) / 1e9
        roof["frac"] = roof["achieved"] / HBM_PEAK_GBS
    if one_cu:
        roof["bound_note"] = ("the dominant kernel is ONE workgroup iterating on a matrix in one CU's LDS: its duration is a chain of dependent "
                              "steps, so the HBM fraction above says nothing about it (it is printed because the contract asks for it); the "
                              "bound that applies is `latency`")
    # ---- the latency bound of one chain's step (see latency_floor_model) against the measured step
    lat = latency_floor_model(model, setup, icp_share, accept_share, bytes_step, flops_step)
    lat["measured_us_per_step"] = 1e6 / rate * B
    lat["frac"] = lat["floor_us_per_step"] / lat["measured_us_per_step"]
    lat["accepted_share"] = accept_share
    roof["latency"] = lat
    # the whole step against both rooflines: which one binds
    hbm_frac = bytes_step * rate / (HBM_PEAK_GBS * 1e9)
    f32_frac = flops_step * rate / (F32_VECTOR_TFLOPS * 1e12)
    f64_frac = flops_step * rate / (F64_VECTOR_TFLOPS * 1e12)
    whole = {"algorithmic_bytes_per_step": bytes_step, "algorithmic_flops_per_step": flops_step, "iterations_per_s": rate,
             "hbm_frac": hbm_frac, "brute_force_flops_x_rate_over_f32_vector_peak": f32_frac,
             "brute_force_flops_x_rate_over_f64_vector_peak": f64_frac,
             "note": "SURVEY.md §8d's BRUTE-FORCE algorithmic figures x measured rate.  The flops figures are a work-equivalent rate, NOT a "
                     "utilisation: the two-level filter decides every (query, element) pair but executes far fewer tests (`executed`), so "
                     "the quotient may exceed 1.  Steps whose proposal is a random walk or a pose move (%.0f %% here) do less."
                     % (100 * (1 - icp_share))}
    if counts:
        # executed tests per step, counted on the device during this leg (per wave): what the searches really did
        n_st = max(n_count * B, 1)
        ex = {kk: vv / n_st for kk, vv in counts.items()}
        # flops actually executed by the searches: ball test ~11, sphere test ~11 (packed f32), exact point-triangle ~60 (f64),
        # exact point-vertex 8 (f64)
        ex_f32 = 11.0 * (ex.get("surface_ball_tests", 0) + ex.get("surface_sphere_tests", 0))
        ex_f64 = 60.0 * ex.get("surface_exact_tests", 0) + 8.0 * (ex.get("vertex_filter_tests", 0) + ex.get("vertex_exact_tests", 0))
        whole["executed"] = {"per_step": ex, "search_flops_f32_per_step": ex_f32, "search_flops_f64_per_step": ex_f64,
                             "f32_vector_frac": ex_f32 * rate / (F32_VECTOR_TFLOPS * 1e12),
                             "f64_vector_frac": ex_f64 * rate / (F64_VECTOR_TFLOPS * 1e12),
                             "brute_force_pairs_per_step": flops_step / 60.0,
                             "note": "tests executed by the searches, counted per wave on the device in this leg; the fractions are utilisations (<= 1)"}
    whole["binding"] = "latency" if lat["frac"] > max(hbm_frac, whole.get("executed", {}).get("f32_vector_frac", 0.0)) else \
        ("flops" if whole.get("executed", {}).get("f32_vector_frac", f32_frac) > hbm_frac else "hbm")
    roof["whole_step"] = whole
    # the distance kernel (north_star: HBM GB/s of the N x M search), whichever kernel dominates
    for dk in ("k_step_filter", "k_surface_filter"):
        if dk in stats:
            dalg = kernel_algorithmic_bytes(dk, model, target, setup)
            davg = stats[dk]["avg_us"]
            if B > 1:
                dalg *= B * args.profile_steps / max(stats[dk]["calls"], 1)
            roof["distance_kernel"] = {"kernel": dk, "avg_launch_us": davg, "launches": stats[dk]["calls"], "algorithmic_bytes": dalg,
                                       "achieved_GBs": dalg / (davg * 1e-6) / 1e9, "frac_hbm": dalg / (davg * 1e-6) / 1e9 / HBM_PEAK_GBS}
            break
    # the projection on the matrix cores (north_star): counter-based busy fraction + algorithmic flops against the matrix-f64 peak
    reg = "k_step_regression"
    if reg in stats:
        per_launch = B * args.profile_steps / max(stats[reg]["calls"], 1) if B > 1 else 1.0
        roof["mfma"] = mfma_block("config%d" % args.config if B == 1 else "many_chains", [reg], {reg: stats[reg]["avg_us"]},
                                  {reg: projection_flops(model, setup) * per_launch})
    if args.events_out:
        # per-kernel durations of this leg, device-side waiting taken out (the rocprof average of a launch that spins on the
        # device for its input contains the wait; these do not) — copied into profiles/ by tools/r5_profiles.sh
        table = {}
        for name, s_ in stats.items():
            w = waits.get(name)
            wait_ms = w["total_ms"] if w is not None else 0.0
            table[name] = {"calls": s_["calls"], "avg_us_with_device_wait": s_["avg_us"], "device_wait_avg_us": 1e3 * wait_ms / max(s_["calls"], 1),
                           "avg_us": 1e3 * max(s_["total_ms"] - wait_ms, 0.0) / max(s_["calls"], 1), "min_us": s_["min_us"], "max_us": s_["max_us"],
                           "algorithmic_bytes_per_launch": kernel_algorithmic_bytes(name, model, target, setup)}
            ab = table[name]["algorithmic_bytes_per_launch"]
            if ab and table[name]["avg_us"] > 0:
                table[name]["algorithmic_GBs"] = ab / (table[name]["avg_us"] * 1e-6) / 1e9
                table[name]["frac_hbm"] = table[name]["algorithmic_GBs"] / HBM_PEAK_GBS
        json.dump({"command": "python3 " + " ".join(sys.argv), "source": "HIP events around every launch on its own stream (icp_ctx_profile_*), "
                   "%d steps after the timed region; `avg_us` excludes the time a launch waits ON THE DEVICE for another stream's word "
                   "(k_step_begin: the decomposition it draws from; k_posterior_eigen: its input when started ahead)" % args.profile_steps,
                   "steps": args.profile_steps, "iterations_per_s": rate, "dominant": dominant, "kernels": table}, open(args.events_out, "w"), indent=1)
    line["kernel_us_per_step"] = {name: round(s["total_ms"] * 1e3 / args.profile_steps, 2) for name, s in stats.items()}
    for name, w in waits.items():
        if w is not None:
            line["kernel_us_per_step"][name + ".device_wait"] = round(w["total_ms"] * 1e3 / args.profile_steps, 2)
    line["kernel_us_per_step_note"] = ("HIP events around each launch (these add ~2-3 us per launch: quote fractions from profiles/*.md where both exist); a step's "
                                       "launches alternate between two streams and overlap the previous step's finish launch; *.device_wait is the part of "
                                       "that kernel's time spent waiting ON THE DEVICE for another stream's launch (already inside the kernel's own figure)")
    return roof


def cpu_baseline_leg(pkg, args, wl, ctx):
    """BASELINE.md §3.  B1 "reference-shaped": the oracle's chain (oracle/icp_oracle.c: sequential, long-form N-point regressions
    as Scalismo does them, posterior/likelihood of the current state carried over like the reference's Memoize) with a KD-tree for
    findClosestPoint and a bounding-volume hierarchy for closestPointOnSurface, the structures of the current model instance rebuilt
    for every new state, ONE thread.  B2: the same chain with brute-force scans over ALL host cores (OpenMP).  Identical results
    (tests/test_oracle.py::test_chain_identical_under_every_search_backend); the first GPU records are checked against B1 here."""
    from oracle import oracle as O
    model, target, setup = wl["model"], wl["target"], wl["setup"]
    om, ot, cfg = oracle_chain_setup(O, wl)  # (the whole mixture, pose walks included: the oracle's chain has them since round 3)
    theta0 = wl["init"](0)
    # cores this process may run on (a container's CPU set, not the machine's count); the B2 scans run per query, so beyond a
    # few dozen threads the fork/join of every scan costs more than it spreads (measured: 256 threads on 116k triangles: 125 s per step)
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = max(1, min(avail, 32))
    out = {}
    def timed(n):
        t = time.perf_counter()
        res = O.run_chain(om, ot, cfg, theta0, 1024, n)
        return res, time.perf_counter() - t

    try:
        # sample sizes: bounded by --cpu-steps AND by time (a first step of each variant is timed as a probe: at BFM size one CPU
        # step takes seconds), so that the default run stays within minutes
        O.set_search_backend(O.SEARCH_TREES)
        _, probe = timed(1)  # (also builds the static target's structures, as the reference does once)
        n1 = int(min(4 * args.cpu_steps, max(1, 15.0 / max(probe, 1e-6))))
        (acc_o, comp_o, _, states_o), d1 = timed(n1)
        kd, bvh = O.search_stats()
        out["B1"] = {"value": n1 / d1, "unit": "iterations/s", "cores": 1,
                     "what": "reference-shaped: sequential chain, KD-tree + bounding-volume hierarchy rebuilt per new state (%d + %d builds so "
                             "far), long-form regressions, Memoize-like carry-over; oracle/icp_oracle.c + icp_spatial.c" % (kd, bvh),
                     "sample": "%d MH steps" % n1}
        O.set_search_backend(O.SEARCH_BRUTE_OMP, cores)
        _, probe = timed(1)
        n2 = int(min(args.cpu_steps, max(1, 10.0 / max(probe, 1e-6))))
        _, d2 = timed(n2)
        out["B2"] = {"value": n2 / d2, "unit": "iterations/s", "cores": cores,
                     "what": "the same chain, every brute-force scan spread over %d threads with OpenMP (the process may use %d of the host's %d "
                             "logical cores; more threads per scan only add fork/join time)" % (cores, avail, os.cpu_count() or 0),
                     "sample": "%d MH steps" % n2}
    finally:
        O.set_search_backend(O.SEARCH_BRUTE)
    # the first records of a fresh GPU chain must reproduce the oracle's decisions (accept/reject, mixture component) and states
    chk = pkg.SamplingRegistration(ctx, setup, theta0, seed=1024)
    crec = chk.run(n1)
    same = bool(np.array_equal(crec[:, 1].astype(np.uint8), acc_o)) and bool(np.array_equal(crec[:, 2].astype(np.int32), comp_o)) and \
        float(np.abs(crec[:, 14:] - states_o[:, 10:]).max()) <= 1e-5 * max(float(np.abs(states_o[:, 10:]).max()), 1e-30)
    chk.close()
    b1p = None
    if args.parallel_cpu_budget > 0 and args.config == 1:
        # the many-chains leg's counterpart on the host: one B1 chain per core (the reference's `.par` over chains), as many as that leg steps
        try:
            b1p = b1_parallel(args, args.config, max(args.many_chains, 1), args.parallel_cpu_budget)
        except Exception as e:
            b1p = {"error": str(e)[:300]}
    return {"value": out["B1"]["value"], "unit": "iterations/s", "cores": 1, "kind": "port", "B1_parallel": b1p,
            "sample": "B1 of BASELINE.md §3 (the baseline the >= 50x target is defined on): %s of the same workload, one thread; host has %d logical cores%s"
                      % (out["B1"]["sample"], os.cpu_count() or 0, ""),
            "B1": out["B1"], "B2": out["B2"], "gpu_matches_oracle_on_sample": same}


if __name__ == "__main__":
    main()
