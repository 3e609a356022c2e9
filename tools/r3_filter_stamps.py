"""dev: where a surface-filter workgroup spends its time (library built with -DICP_FILTER_STAMPS, loaded through ICP_LIBRARY_PATH):
single chain (configs[1]), configs[2], and the 64-chain on-device loop."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
lib = ctypes.CDLL(os.environ["ICP_LIBRARY_PATH"])
def stamps(tag, reset=True):
    out = (ctypes.c_ulonglong * 16)()
    lib.icp_debug_filter_stamps(out, 1 if reset else 0)
    n = max(1, out[0])
    print("%s: workgroups %d | per workgroup (us): spheres+ball %.2f, staging %.2f, tests %.2f, settle %.2f, total %.2f | ball groups/wg %.1f, survivors/wg %.2f" % (
        tag, out[0], out[1] / n / 100, out[2] / n / 100, out[3] / n / 100, out[4] / n / 100, out[5] / n / 100, out[6] / n / 64.0, out[7] / n), flush=True)
model, target = pkg.data.synthetic_femur_target()
# single chain, configs[1]
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctx = pkg.IcpContext(model, target, device=0)
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=5)
ch.run(30, want_records=False); stamps("warm-up", True)
ch.run(300, want_records=False); stamps("configs[1] single chain, 300 steps")
ch.close(); ctx.close()
B = 64
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
pkg.run_chains_batched(chains, 40, want_records=False); stamps("warm-up", True)
pkg.run_chains_batched(chains, 300, want_records=False); stamps("64 chains, device loop, 300 steps")
