#!/usr/bin/env python3
"""Two host threads, each stepping a batch of femur chains on the same GPU (a fresh process): per-call wall times and the library's
fall-back counters.  (tools/multichain.py batched2 16 did not finish in round 4.)"""
import sys, time, threading, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
model, target = pkg.data.synthetic_femur_target()
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], pkg.femur_icp_proposal_registration(model, target), pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
groups = [chains[i::2] for i in range(2)]
def work(k, grp):
    for rep in range(6):
        t0 = time.perf_counter()
        pkg.run_chains_batched(grp, 5, want_records=False)
        print("thread %d call %d: %.1f ms for 5 steps of %d chains" % (k, rep, 1e3 * (time.perf_counter() - t0), len(grp)), flush=True)
ths = [threading.Thread(target=work, args=(k, grp)) for k, grp in enumerate(groups)]
[t.start() for t in ths]; [t.join() for t in ths]
print("runtime stats:", pkg._native.runtime_stats(), flush=True)
[c.close() for c in chains]; [c.close() for c in ctxs]
