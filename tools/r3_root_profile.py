"""dev: per-kernel HIP-event times of the metric chain with either sampler."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target()
for sampler in sys.argv[1:] or ["eigen", "cholesky-root"]:
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    setup.sampler = sampler
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    chain.run(200, want_records=False)
    t = time.perf_counter(); rec = chain.run(2000); dt = time.perf_counter() - t
    print(sampler, "rate", 2000 / dt, "acc", rec[:, 1].mean())
    ctx.profile_start(max_launches=40000)
    chain.run(300, want_records=False)
    st = ctx.profile_stop()
    for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
        print("   %-32s calls %5d avg %8.2f us  min %8.2f max %8.2f" % (k, v["calls"], v["avg_us"], v["min_us"], v["max_us"]))
    chain.close(); ctx.close()
