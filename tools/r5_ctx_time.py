import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=100)
c0 = pkg.IcpContext(model, target, device=0)
c1 = pkg.IcpContext(model, target, device=0)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
t0 = time.perf_counter()
cs = [pkg.IcpContext(model, target, device=0) for _ in range(10)]
dt = time.perf_counter() - t0
pr.disable()
print("10 contexts %.2f ms each" % (1e2 * dt))
pstats.Stats(pr).sort_stats("cumulative").print_stats(12)
