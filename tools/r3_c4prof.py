"""Where a configs[4] work item's time goes on the host: cProfile over one target's ten 50-step chains (developer tool)."""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(2)]
make_setup = lambda m, t: pkg.bfm_fitting_partial(m, t, evaluator="collective")
pkg.sharding.run_batch(pkg, model, targets[:1], n_chains=1, n_steps=5, make_setup=make_setup)
t0 = time.perf_counter()
pr = cProfile.Profile(); pr.enable()
items, recs, stats = pkg.sharding.run_batch(pkg, model, targets, n_chains=10, n_steps=50, make_setup=make_setup, return_stats=True)
pr.disable()
dt = time.perf_counter() - t0
print("items", len(items), "job_s", dt, "it/s", len(items) * 50 / dt, stats)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
