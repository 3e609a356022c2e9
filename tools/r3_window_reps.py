import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target(n_subdiv=6)
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
for rep in range(4):
    ctx = pkg.IcpContext(model, target, device=0)
    ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    ch.run(5, want_records=False)
    t0 = time.perf_counter(); rec = ch.run(20); dt = time.perf_counter() - t0
    print("rep", rep, "window rate %.0f" % (20 / dt), "accepted", int(rec[:, 1].sum()), flush=True)
    ch.close(); ctx.close()
