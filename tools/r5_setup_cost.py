"""dev: where the set-up of a batch registration goes (configs[4]): contexts, set_target, chain objects — wall time per phase; run under
rocprofv3 --hip-runtime-trace --stats for the HIP API calls behind them.  usage: r5_setup_cost.py [n_contexts] [n_chain_rounds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
nc = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
model = pkg.data.synthetic_face_model()
targets = [pkg.data.synthetic_partial_target(model, seed=100 + t) for t in range(4)]
make_setup = lambda m, t: pkg.bfm_fitting_partial(m, t, evaluator="collective", fused=2)
t0 = time.perf_counter(); c0 = pkg.IcpContext(model, targets[0], device=0); t_first = time.perf_counter() - t0
t0 = time.perf_counter(); ctxs = [c0] + [pkg.IcpContext(model, targets[0], device=0) for _ in range(nc - 1)]; t_ctx = time.perf_counter() - t0
setups = [make_setup(model, t) for t in targets]
print("first context %.1f ms; %d more contexts %.1f ms (%.2f ms each)" % (1e3 * t_first, nc - 1, 1e3 * t_ctx, 1e3 * t_ctx / max(nc - 1, 1)), flush=True)
for rd in range(rounds):
    t0 = time.perf_counter()
    for cx in ctxs:
        cx.setTarget(targets[(rd + 1) % 4])
    t_set = time.perf_counter() - t0
    t0 = time.perf_counter()
    chains = [pkg.SamplingRegistration(cx, setups[(rd + 1) % 4], pkg.random_initial_parameters(model, i), seed=7 + i) for i, cx in enumerate(ctxs)]
    t_ch = time.perf_counter() - t0
    t0 = time.perf_counter(); pkg.run_chains_batched(chains, 5, want_records=False); t_run = time.perf_counter() - t0
    t0 = time.perf_counter(); [c.close() for c in chains]; t_close = time.perf_counter() - t0
    print("round %d: set_target %.2f ms each, chain objects %.2f ms each, 5 steps %.1f ms, close %.2f ms each" % (
        rd, 1e3 * t_set / nc, 1e3 * t_ch / nc, 1e3 * t_run, 1e3 * t_close / nc), flush=True)
t0 = time.perf_counter(); [c.close() for c in ctxs]; print("context close %.2f ms each" % (1e3 * (time.perf_counter() - t0) / nc))
