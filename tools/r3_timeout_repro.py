"""Round 3: root cause of the 50 ms device-side time-out that round 2's 64-chain profile showed
(profiles/r02_v2_batched64_kernel_stats.md: k_step_begin_batch_reg max 50,062 us, k_posterior_eigen_rr<24> max 50,089 us).

Run with the test-hooks build of the library (ICP_LIBRARY_PATH=icp-proposal_amd/libicp_proposal_amd_testhooks.so):
  arm "r2"    ICP_TEST_EIGEN_CHUNK=24 ICP_TEST_NO_GATE=1 — round 2's schedule: more than 24 decompositions per batch go out as
              several launches on ONE stream (each waits for the one before it), and the batch's chip-wide first launch, whose
              workgroups spin on those decompositions' completion words, is not held back;
  arm "gate"  ICP_TEST_EIGEN_CHUNK=24 — the same several launches, but the launch sequence gated on their residency;
  arm "r3"    nothing set — one launch for all decompositions + the gate (the shipped schedule).
Prints the fall-back counters (icp_ctx_runtime_stats) and the rate of 64 chains x 120 steps from a cold start (burn-in: ~85 % of
the steps are accepted, i.e. ~54 decompositions per batch of 32 chains)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
B, n = int(sys.argv[1]) if len(sys.argv) > 1 else 64, int(sys.argv[2]) if len(sys.argv) > 2 else 120
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
t0 = time.perf_counter()
recs = pkg.run_chains_batched(chains, n)
dt = time.perf_counter() - t0
acc = sum(r[:, 1].sum() for r in recs) / (B * n)
print("arm=%s chains=%d steps=%d accepted=%.2f rate=%.0f it/s  stats=%s" % (
    os.environ.get("ARM", "?"), B, n, acc, B * n / dt, pkg._native.runtime_stats()), flush=True)
[c.close() for c in chains]; [c.close() for c in ctxs]
