"""dev: what icp_ctx_expect buys — 25 contexts of one keyed model + their chain objects, with and without the hint (ICP_NO_STREAM_PREWARM=1)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(2)]
setup = pkg.bfm_fitting_partial(model, targets[0], evaluator="collective", fused=2)
t0 = time.perf_counter(); pkg.expect_contexts(0, 25); t_hint = time.perf_counter() - t0
t0 = time.perf_counter(); c0 = pkg.IcpContext(model, targets[0], device=0); t_first = time.perf_counter() - t0
ch0 = pkg.SamplingRegistration(c0, setup, pkg.random_initial_parameters(model, 0), seed=7); ch0.run(5); ch0.close()
if len(sys.argv) > 1: time.sleep(float(sys.argv[1]))
t0 = time.perf_counter(); ctxs = [pkg.IcpContext(model, targets[0], device=0) for _ in range(24)]; t_ctx = time.perf_counter() - t0
t0 = time.perf_counter()
chains = [pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, i), seed=7 + i) for i, cx in enumerate([c0] + ctxs)]
t_ch = time.perf_counter() - t0
print("hint %.2f ms, first context %.1f ms, 24 more contexts %.1f ms (%.2f each), 25 chain objects %.1f ms (%.2f each)" % (
    1e3 * t_hint, 1e3 * t_first, 1e3 * t_ctx, 1e3 * t_ctx / 24, 1e3 * t_ch, 1e3 * t_ch / 25))
