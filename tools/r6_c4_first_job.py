"""dev: the first 10 x 10 x 50 job of a process phase by phase (bench.py config4_leg's flow), this tree or another (argv[1] = repo root)"""
import os, sys, time
root = sys.argv[1] if len(sys.argv) > 1 else os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(10)]
make_setup = lambda m, t: pkg.bfm_fitting_partial(m, t, evaluator="collective", fused=2)
if hasattr(pkg, "expect_contexts"): pkg.expect_contexts(0, 25)
pkg.sharding.run_batch(pkg, model, targets[:1], n_chains=1, n_steps=5, make_setup=make_setup, dist=None, device_index=0)
for n_steps in (50, 50, 300):
    t0 = time.perf_counter()
    items, recs, stats = pkg.sharding.run_batch(pkg, model, targets, n_chains=10, n_steps=n_steps, make_setup=make_setup, dist=None, device_index=0, return_stats=True)
    dt = time.perf_counter() - t0
    print(n_steps, "%.0f it/s" % (len(items) * n_steps / dt), stats.get("phase_ms"), flush=True)
