"""dev: where the creation of a chain object (SamplingRegistration) goes on the Python side (cProfile)"""
import os, sys, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=100)
setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(12)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=7 + i) for i in range(2)]
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
more = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=7 + i) for i in range(2, 12)]
dt = time.perf_counter() - t0
pr.disable()
print("10 chain objects %.2f ms each" % (1e2 * dt))
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
