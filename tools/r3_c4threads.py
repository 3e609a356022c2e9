"""configs[4]-style chains of ONE target stepped concurrently from host threads, one context per chain (developer experiment):
aggregate iterations/s against the number of threads."""
import os, sys, time
from concurrent.futures import ThreadPoolExecutor
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import __graft_entry__ as g
pkg = g.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=100)
setup = pkg.bfm_fitting_partial(model, target, evaluator=sys.argv[1] if len(sys.argv) > 1 else "collective")
steps = 50
for nthr in (1, 2, 4, 6, 10):
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(nthr)]
    chains = [pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, k, 1024), seed=2000 + k) for k, cx in enumerate(ctxs)]
    for ch in chains: ch.run(3, want_records=False)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(nthr) as ex:
        recs = list(ex.map(lambda ch: ch.run(steps), chains))
    dt = time.perf_counter() - t0
    print("threads %2d: %8.1f it/s aggregate (%.1f ms for %d steps each)" % (nthr, nthr * steps / dt, 1e3 * dt, steps), flush=True)
    for ch in chains: ch.close()
    for cx in ctxs: cx.close()
print(pkg._native.runtime_stats())
