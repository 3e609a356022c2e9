"""Executed tests of the full-mesh Hausdorff evaluator's two searches at configs[3] size, with hints from a neighbouring state
(developer tool): how many exact point-triangle evaluations the resolve stage runs, and how many queries overflow their lists."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
hd = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
th = pkg.initial_parameters(model)
rng = np.random.default_rng(1)
hd.logValue(th)
for k in range(3):
    th = th.copy(); th[10:] += 0.02 * rng.normal(size=model.rank); th[1:4] += 0.2 * rng.normal(size=3)
    ctx.profile_start(64, count_searches=True)
    v = hd.logValue(th)
    rows = ctx.profile_stop()
    print("state %d value %.6f" % (k, v))
    for name, r in rows.items():
        print("    %-32s calls %10d  total %.1f us" % (name, r["calls"], 1e3 * r["total_ms"]))
print("N", model.n_points, "T_target", target.n_cells, "V_target", target.n_points, "T_model", model.n_cells)
