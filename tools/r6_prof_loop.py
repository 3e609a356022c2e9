"""dev: does the HIP-event instrumentation (icp_ctx_profile_*) see the launches of the wide on-device loop?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
B = int(sys.argv[1]) if len(sys.argv) > 1 else 25
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=100)
setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
pkg.expect_contexts(0, B)
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=7 + i) for i in range(B)]
pkg.run_chains_batched(chains, 30, want_records=False)
ctxs[0].profile_start(max_launches=200 * 60 + 4096)
t0 = time.perf_counter()
pkg.run_chains_batched(chains, 60, want_records=False)
dt = time.perf_counter() - t0
st = ctxs[0].profile_stop()
print("%.0f it/s, %.2f ms per round" % (B * 60 / dt, 1e3 * dt / 60))
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
    print("%-28s calls %5d avg %8.1f us total %8.2f ms" % (k, v["calls"], v["avg_us"], v["total_ms"]))
print(ctxs[0].step_paths())
