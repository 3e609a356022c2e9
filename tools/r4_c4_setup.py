#!/usr/bin/env python3
"""Where a batch-registration target's wall time goes (configs[4]): contexts, chains, first steps, steady steps."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
model = pkg.data.synthetic_face_model()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 10
for t in range(3):
    target = pkg.data.synthetic_partial_target(model, seed=100 + t)
    setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    t0 = time.perf_counter()
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
    t1 = time.perf_counter()
    chains = [pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, k), seed=1024 + k) for k, cx in enumerate(ctxs)]
    t2 = time.perf_counter()
    pkg.run_chains_batched(chains, 2, want_records=False)
    t3 = time.perf_counter()
    pkg.run_chains_batched(chains, 50, want_records=False)
    t4 = time.perf_counter()
    pkg.run_chains_batched(chains, 300, want_records=False)
    t5 = time.perf_counter()
    for ch in chains: ch.close()
    for cx in ctxs: cx.close()
    t6 = time.perf_counter()
    print("target %d: contexts %.0f ms, chains %.0f ms, first 2 steps %.0f ms, 50 steps %.0f ms (%.0f it/s), 300 steps %.0f ms (%.0f it/s), close %.0f ms" % (
        t, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), B * 50 / (t4 - t3), 1e3 * (t5 - t4), B * 300 / (t5 - t4), 1e3 * (t6 - t5)), flush=True)
print(pkg._native.step_paths(), pkg._native.runtime_stats())
