"""dev: where a step of configs[3] (face stand-in, rank 200, boundary-aware, collective evaluator) spends its time."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
ev = sys.argv[1] if len(sys.argv) > 1 else "collective"
ch = pkg.SamplingRegistration(ctx, pkg.bfm_fitting_partial(model, target, evaluator=ev), pkg.initial_parameters(model), seed=5)
ch.run(10, want_records=False)
n = 40
ctx.profile_start()
t0 = time.perf_counter(); ch.run(n, want_records=False); dt = time.perf_counter() - t0
st = ctx.profile_stop()
print(f"{n / dt:.1f} it/s with event timing; per step (us):")
for k, v in sorted(st.items(), key=lambda kv: -kv[1]["total_ms"]):
    print(f"  {k:28s} calls/step {v['calls'] / n:6.1f}  avg {v['avg_us']:9.1f}  per step {1e3 * v['total_ms'] / n:10.1f}")
print(f"  sum of kernels per step {sum(1e3 * v['total_ms'] for v in st.values()) / n:.1f} us; wall per step {1e6 * dt / n:.1f} us")
