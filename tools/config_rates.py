"""dev: steady-state MH iterations/s of the other BASELINE.json configurations on one GPU (context for DESIGN.md §8)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()

def rate(ctx, setup, theta0, n_warm, n, seed=5):
    ch = pkg.SamplingRegistration(ctx, setup, theta0, seed=seed)
    ch.run(n_warm, want_records=False)
    t0 = time.perf_counter(); ch.run(n, want_records=False); dt = time.perf_counter() - t0
    _, _, steps, acc = ch.state()
    ch.close()
    return n / dt, acc / steps

# configs[2]: femur-100, all points, symmetric evaluator
model, target = pkg.data.load_femur_model_and_target(100)
ctx = pkg.IcpContext(model, target, device=0)
r, a = rate(ctx, pkg.femur_random_init_comparison(model, target), pkg.random_initial_parameters(model, chain_index=3), 20, 200)
print(f"configs[2] femur-100 all points (K = N = 1622, rank 101, symmetric evaluator): {r:8.1f} it/s, acceptance {a:.2f}", flush=True)
ctx.close()
# configs[3]: face stand-in
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
for sampler in ("eigen", "cholesky-root"):
    for ev in ("collective", "hausdorff"):
        setup = pkg.bfm_fitting_partial(model, target, evaluator=ev)
        setup.sampler = sampler
        r, a = rate(ctx, setup, pkg.initial_parameters(model), 10, 200)
        print(f"configs[3] face stand-in (N = {model.n_points}, rank {model.rank}), {ev} evaluator, {sampler} sampler: {r:8.1f} it/s, acceptance {a:.2f}", flush=True)
ctx.close()
