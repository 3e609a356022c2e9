"""dev: configs[3] chain rate after k small contexts were made and closed before its own (stream -> hardware queue assignment?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
k = int(sys.argv[1])
fm, ft = pkg.data.load_femur_model_and_target(50)
for i in range(k):
    c = pkg.IcpContext(fm, ft, device=0); c.close()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff")
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
ch.run(100, want_records=False)
t0 = time.perf_counter(); ch.run(600, want_records=False); dt = time.perf_counter() - t0
print("prior contexts %d (pool %s): %.0f it/s" % (k, os.environ.get("ICP_NO_POOL", "on"), 600 / dt), flush=True)
