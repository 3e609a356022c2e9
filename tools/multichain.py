"""dev: B independent chains on ONE GPU (metric workload).  threads: one host thread + one context each, every chain
stepping on its own (icp_chain_step); batched: one thread, one icp_chain_step_batched submission per lockstep step;
batched2: two threads with half of the chains each (one batch computes while the other's host side runs)."""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
modes = sys.argv[1].split(",") if len(sys.argv) > 1 else ["threads", "batched", "batched2"]
sizes = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 8, 16, 32]
model, target = pkg.data.synthetic_femur_target()
for mode in modes:
    for B in sizes:
        ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
        chains = [pkg.SamplingRegistration(ctxs[i], pkg.femur_icp_proposal_registration(model, target), pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
        if mode == "threads":
            groups = [[c] for c in chains]
            def work(grp, n): grp[0].run(n, want_records=False)
        else:
            k = 2 if (mode == "batched2" and B >= 2) else 1
            groups = [chains[i::k] for i in range(k)]
            def work(grp, n): pkg.run_chains_batched(grp, n, want_records=False)
        n = 1500 if mode == "threads" else max(200, 3000 // B)
        for steps in (100, n):
            ths = [threading.Thread(target=work, args=(grp, steps)) for grp in groups]
            t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.perf_counter() - t0
        print(f"{mode:9s} B={B:2d}: {B * n / dt:9.0f} it/s total, {n / dt:8.0f} per chain, {1e6 * dt / n:7.1f} us per lockstep step", flush=True)
        [c.close() for c in chains]; [c.close() for c in ctxs]
print("runtime stats:", pkg._native.runtime_stats(), flush=True)
