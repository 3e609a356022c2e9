"""dev: k_step_filter time vs number of surface queries (evaluator points), femur-50 vs 58k target."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
model, target = pkg.data.synthetic_femur_target()
for ke in (102, 204, 408, 816, 1622):
    ctx = pkg.IcpContext(model, target, device=0)
    setup = pkg.femur_icp_proposal_registration(model, target, n_eval_points=ke)
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    chain.run(200, want_records=False)
    ctx.profile_start(max_launches=64 * 300)
    chain.run(300, want_records=False)
    st = ctx.profile_stop()
    print(ke, {k: round(v["avg_us"], 2) for k, v in st.items() if k in ("k_step_filter", "k_step_resolve", "k_step_begin")}, flush=True)
    chain.close(); ctx.close()
