"""dev: which part of 'a context was made before' slows the configs[3] chain down?  mode: none | streams | malloc | ctx_open | face_ctx"""
import os, sys, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as g
pkg = g.load_package()
mode = sys.argv[1]
hip = ctypes.CDLL("libamdhip64.so")
keep = None
if mode == "streams":
    lo, hi = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi))
    ss = []
    for i in range(3):
        s = ctypes.c_void_p(); hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, hi.value); ss.append(s)
    for s in ss: hip.hipStreamDestroy(s)
elif mode == "streams_default":
    ss = []
    for i in range(3):
        s = ctypes.c_void_p(); hip.hipStreamCreateWithFlags(ctypes.byref(s), 1); ss.append(s)
    for s in ss: hip.hipStreamDestroy(s)
elif mode == "malloc":
    ps = []
    for i in range(200):
        p = ctypes.c_void_p(); hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(1 << 20)); ps.append(p)
    for p in ps: hip.hipFree(p)
elif mode == "ctx_open":
    fm, ft = pkg.data.load_femur_model_and_target(50)
    keep = pkg.IcpContext(fm, ft, device=0)
elif mode in ("femur_closed", "femur_closed_release", "femur100_closed"):
    fm, ft = pkg.data.load_femur_model_and_target(100 if mode == "femur100_closed" else 50)
    c = pkg.IcpContext(fm, ft, device=0); c.close()
    if mode == "femur_closed_release":
        pkg._native.lib().icp_release_cached_models()
elif mode == "femur50_then_100":
    for nc in (50, 100):
        fm, ft = pkg.data.load_femur_model_and_target(nc)
        c = pkg.IcpContext(fm, ft, device=0); c.close()
elif mode == "femur50_x2_open_closed":   # two femur-50 contexts alive together (six streams), both closed: the pool holds six
    fm, ft = pkg.data.load_femur_model_and_target(50)
    c1 = pkg.IcpContext(fm, ft, device=0); c2 = pkg.IcpContext(fm, ft, device=0); c1.close(); c2.close()
elif mode == "face_ctx":
    pass
model = pkg.data.synthetic_face_model()
target = pkg.data.synthetic_partial_target(model, seed=7)
if mode == "face_ctx":
    c = pkg.IcpContext(model, target, device=0); c.close()
ctx = pkg.IcpContext(model, target, device=0)
setup = pkg.bfm_fitting_partial(model, target, evaluator="hausdorff")
ch = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
ch.run(100, want_records=False)
t0 = time.perf_counter(); ch.run(600, want_records=False); dt = time.perf_counter() - t0
print("before the context: %s -> %.0f it/s" % (mode, 600 / dt), flush=True)
