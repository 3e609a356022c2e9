#!/usr/bin/env python3
"""configs[4]: chains of SEVERAL targets in one submission (they share the model; the wide step only needs that)."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
model = pkg.data.synthetic_face_model()
nT = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B = 10
targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(nT)]
ctxs, chains = [], []
for t in range(nT):
    setup = pkg.bfm_fitting_partial(model, targets[t], evaluator="collective", fused=2)
    for k in range(B):
        cx = pkg.IcpContext(model, targets[t], device=0)
        ctxs.append(cx)
        chains.append(pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, k), seed=1024 + 1000 * t + k))
pkg.run_chains_batched(chains, 20, want_records=False)
for n in (100, 300):
    t0 = time.perf_counter()
    pkg.run_chains_batched(chains, n, want_records=False)
    dt = time.perf_counter() - t0
    print("targets %d chains %d steps %d: %.0f it/s (%.2f ms per round)" % (nT, len(chains), n, len(chains) * n / dt, 1e3 * dt / n), flush=True)
print(pkg._native.runtime_stats())
for ch in chains: ch.close()
for cx in ctxs: cx.close()
