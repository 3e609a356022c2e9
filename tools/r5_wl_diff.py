#!/usr/bin/env python3
"""which part of the posterior differs between the merged step and the wide loop at femur rank 101 (developer experiment)"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as graft
pkg = graft.load_package(); nat = pkg._native
model, target = pkg.data.load_femur_model_and_target(100)
r = model.rank
tp = pkg.data.decimated_point_subset(target, 2 * r)
def objects(ctx):
    props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.TargetSampling, True, decimatedTargetPoints=tp),
             pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True, decimatedTargetPoints=tp)]
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r, decimatedTargetPoints=tp)
    return ev, props
theta = pkg.random_initial_parameters(model, 0)
# --- merged step: one ICP step with generator 0 and the z the loop draws at step 0 … instead: run the loop first, read what it proposed
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(2)]
objs = [objects(c) for c in ctxs]
B = 2
mix = nat.MhMixture(C.sizeof(nat.MhMixture), (C.c_double * 2)(1.0, 0.0), 1.0, 0.0, 0.1)   # always ICP proposal 0
eh = (C.c_void_p * B)(*[o[0].h for o in objs]); ph = (C.c_void_p * (2 * B))(*[p.h for o in objs for p in o[1]])
seeds = (C.c_uint64 * B)(9, 10); first = (C.c_int64 * B)(0, 0)
th = [theta.copy() for _ in range(B)]
thp = (nat.c_double_p * B)(*[t.ctypes.data_as(nat.c_double_p) for t in th])
logp = np.full(B, -1e300); acc = (C.c_int64 * B)(0, 0)
rec = [np.zeros((1, 14 + r)) for _ in range(B)]
recp = (nat.c_double_p * B)(*[x.ctypes.data_as(nat.c_double_p) for x in rec])
rc = nat.lib().icp_chains_run_on_device(B, eh, 2, ph, C.byref(mix), seeds, first, thp, logp.ctypes.data_as(nat.c_double_p), 1, recp, acc)
print("loop rc", rc, nat.lib().icp_last_error(), "accepted", list(acc), ctxs[0].step_paths())
new = th[0].copy()
post_loop = [p.icpPosterior(new) for p in objs[0][1]]
# --- the same state's posterior the per-method way, on a fresh context
ref = pkg.IcpContext(model, target, device=0)
ev, props = objects(ref)
post_ref = [p.icpPosterior(new) for p in props]
for i in range(2):
    a, b = post_loop[i], post_ref[i]
    print("proposal", i, "alpha", np.abs(a.alpha - b.alpha).max(), "M", np.abs(a.M - b.M).max(), "V", np.abs(a.V - b.V).max(), "S", np.abs(a.S - b.S).max(),
          "corr", np.array_equal(a.corr_id, b.corr_id), np.abs(a.corr_point - b.corr_point).max())
# --- and through the merged step from the initial state with the loop's z
import math
M64 = (1 << 64) - 1
def sm(x):
    x = (x + 0x9E3779B97F4A7C15) & M64
    x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
    x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
    return x ^ (x >> 31)
def uni(seed, step, lane):
    h = sm(sm(sm(seed) ^ ((step * 0xD1342543DE82EF95) & M64)) ^ ((lane * 0x2545F4914F6CDD1D) & M64))
    return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0)
def normal(seed, step, lane):
    return math.sqrt(-2.0 * math.log(uni(seed, step, 2 * lane + 1000))) * math.cos(2.0 * math.pi * uni(seed, step, 2 * lane + 1001))
z = np.array([normal(9, 0, j) for j in range(r)])
if z is not None:
    ref2 = pkg.IcpContext(model, target, device=0)
    ev2, props2 = objects(ref2)
    out = pkg.chain_step(ev2, props2, theta, 0, z=z)
    print("merged step proposes the same state:", np.array_equal(out[0], new), np.abs(out[0] - new).max(), ref2.step_paths())
    post_m = [p.icpPosterior(out[0]) for p in props2]
    for i in range(2):
        a, b = post_m[i], post_ref[i]
        print("merged vs per-method", i, "alpha", np.abs(a.alpha - b.alpha).max(), "M", np.abs(a.M - b.M).max(), "V", np.abs(a.V - b.V).max(), "S", np.abs(a.S - b.S).max())
