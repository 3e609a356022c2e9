#!/usr/bin/env python3
"""Cost of making and destroying the objects of one face-model chain (context, proposal + evaluator through SamplingRegistration), with
the resource pools warm: tools/r4_pool.sh compares ICP_NO_POOL=1."""
import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as graft
pkg = graft.load_package()
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=100)
setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
keep = pkg.IcpContext(model, target, device=0)   # (keeps the model's device data alive, and is the process's first context)
for rep in range(4):
    t0 = time.perf_counter()
    cx = pkg.IcpContext(model, target, device=0)
    t1 = time.perf_counter()
    ch = pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, 1), seed=5)
    t2 = time.perf_counter()
    ch.run(2, want_records=False)
    t3 = time.perf_counter()
    ch.close()
    t4 = time.perf_counter()
    cx.close()
    t5 = time.perf_counter()
    print("rep %d: context %.2f ms, chain %.2f ms, two steps %.2f ms, chain close %.2f ms, context close %.2f ms" % (
        rep, 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2), 1e3 * (t4 - t3), 1e3 * (t5 - t4)), flush=True)
keep.close()
