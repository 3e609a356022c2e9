"""dev: the on-device MH loop against the host-stepped lockstep path (ICP_HOST_DEVICE_LOOP=0): same records, rates."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
n = int(sys.argv[2]) if len(sys.argv) > 2 else 300
sampler = sys.argv[3] if len(sys.argv) > 3 else "eigen"
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
setup.sampler = sampler
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
pkg.run_chains_batched(chains, 40, want_records=False)
t0 = time.perf_counter(); recs = pkg.run_chains_batched(chains, n); dt = time.perf_counter() - t0
t0 = time.perf_counter(); pkg.run_chains_batched(chains, n, want_records=False); dt2 = time.perf_counter() - t0
acc = sum(r[:, 1].sum() for r in recs) / (B * n)
print("mode=%s sampler=%s chains=%d steps=%d accepted=%.3f rate=%.0f it/s (without records %.0f) stats=%s" % (
    os.environ.get("ICP_HOST_DEVICE_LOOP", "device"), sampler, B, n, acc, B * n / dt, B * n / dt2, pkg._native.runtime_stats()), flush=True)
np.save(sys.argv[4] if len(sys.argv) > 4 else "/tmp/r3_dl.npy", np.stack(recs))
[c.close() for c in chains]; [c.close() for c in ctxs]
