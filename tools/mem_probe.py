"""Developer probe: device memory a further context + chain takes, by stage (hipMemGetInfo of the runtime the library uses)."""
import ctypes, sys
import numpy as np
sys.path.insert(0, '.')
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
first = pkg.IcpContext(model, target, device=0)
ch0 = pkg.SamplingRegistration(first, setup, pkg.initial_parameters(model), seed=1024); ch0.run(30)
with open("/proc/self/maps") as f:
    paths = {line.split()[-1] for line in f if "libamdhip64" in line}
hip = ctypes.CDLL(sorted(paths)[0])
def free():
    a, b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipDeviceSynchronize(); hip.hipMemGetInfo(ctypes.byref(a), ctypes.byref(b)); return a.value
f0 = free()
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(8)]
f1 = free(); print("context:            %.2f MB each" % ((f0 - f1) / 8e6))
chains = [pkg.SamplingRegistration(c, setup, pkg.initial_parameters(model), seed=1024) for c in ctxs]
f2 = free(); print("chain (proposals, evaluator): %.2f MB each" % ((f1 - f2) / 8e6))
for ch in chains: ch.run(30)
f3 = free(); print("first 30 steps:     %.2f MB each" % ((f2 - f3) / 8e6))
recs = pkg.run_chains_batched(chains, 30)
f4 = free(); print("batched steps:      %.2f MB each (total %.2f MB per context)" % ((f3 - f4) / 8e6, (f0 - f4) / 8e6))
