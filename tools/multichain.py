"""dev: B independent chains on ONE GPU, one host thread + one context (stream) each."""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target()
for B in (1, 2, 4, 8, 16):
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
    chains = [pkg.SamplingRegistration(ctxs[i], pkg.femur_icp_proposal_registration(model, target), pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
    def work(ch, n): ch.run(n, want_records=False)
    ths = [threading.Thread(target=work, args=(c, 100)) for c in chains]
    [t.start() for t in ths]; [t.join() for t in ths]
    n = 1500
    ths = [threading.Thread(target=work, args=(c, n)) for c in chains]
    t0 = time.perf_counter(); [t.start() for t in ths]; [t.join() for t in ths]; dt = time.perf_counter() - t0
    print(f"B={B:2d}: {B * n / dt:9.0f} it/s total, {n / dt:8.0f} per chain", flush=True)
    [c.close() for c in chains]; [c.close() for c in ctxs]
