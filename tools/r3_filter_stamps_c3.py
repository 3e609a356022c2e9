"""dev: the per-stage surface filter's workgroup times on the configs[3] chain (library built with -DICP_FILTER_STAMPS)."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
lib = ctypes.CDLL(os.environ["ICP_LIBRARY_PATH"])
def stamps(tag):
    out = (ctypes.c_ulonglong * 16)()
    lib.icp_debug_filter_stamps_geometry(out, 1)
    n = max(1, out[0])
    print("%s: workgroups %d | per workgroup (us): spheres+ball %.2f, staging %.2f, tests %.2f, settle %.2f, total %.2f | ball groups/wg %.1f, survivors/wg %.2f" % (
        tag, out[0], out[1] / n / 100, out[2] / n / 100, out[3] / n / 100, out[4] / n / 100, out[5] / n / 100, out[6] / n / 64.0, out[7] / n), flush=True)
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
ctx = pkg.IcpContext(model, target, device=0)
setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff")
setup.sampler = sys.argv[1] if len(sys.argv) > 1 else "cholesky-root"
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=5)
ch.run(100, want_records=False); stamps("warm-up")
ch.run(300, want_records=False); stamps("configs[3], 300 steps")
