"""GPU: round-5 robustness items, each through the C ABI.

* the launch context's rings hold ICP_MAX_BATCHES_IN_FLIGHT batches: one ticket more is refused with ICP_ERR_BUSY and nothing is issued;
  chains stepped in FIVE and EIGHT lockstep groups (more than the four the rings held until round 4) give the records of the same
  chains stepped one by one;
* the device-buffer pool gives its blocks back to the runtime and retries when an allocation fails (test-hooks build: the n-th
  hipMalloc of the process "fails");
* the rotation-convention check of icp_ctx_set_rotation (ModelFittingParameters.scala:79-86): a context whose registered matrices all
  agree with the library's Rz·Ry·Rx runs the pose mixture inside the on-device loop, decision for decision the oracle's chain; one
  whose matrix disagreed is refused there and keeps working on the host-stepped path;
* icp_mh_mixture::struct_size guards the extended mixture structure.
"""
import ctypes
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

ICP_ERR_INVALID_ARG, ICP_ERR_BUSY = -1, -6
MAX_BATCHES = 8  # include/icp_proposal.h: ICP_MAX_BATCHES_IN_FLIGHT


def _chain_objects(pkg, ctx, r, tp):
    props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.TargetSampling, True, decimatedTargetPoints=tp),
             pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True, decimatedTargetPoints=tp)]
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r, decimatedTargetPoints=tp)
    return ev, props


def test_launch_context_refuses_more_tickets_than_its_rings_hold(pkg, femur50):
    """icp_chain_step_batched_issue advances the launch context's rings (pinned argument slots, eigen records, gate words) modulo
    ICP_MAX_BATCHES_IN_FLIGHT: a ninth uncollected ticket on one launch context would rewrite what the first one's launches still read.
    It is refused (ICP_ERR_BUSY, nothing issued, its contexts stay free); after one ticket is collected the same issue succeeds, and
    every ticket's step equals the step of a fresh context."""
    model, target = femur50
    r = model.rank
    rng = np.random.default_rng(5)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    n = MAX_BATCHES + 1
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(n)]
    objs = [_chain_objects(pkg, c, r, tp) for c in ctxs]
    thetas, zs = [], []
    for b in range(n):
        t = pkg.initial_parameters(model)
        t[10:] = 0.3 * rng.normal(size=r)
        thetas.append(t)
        zs.append(rng.normal(size=r))
    ref_ctx = pkg.IcpContext(model, target, device=0)
    want = []
    for b in range(n):
        ev, props = _chain_objects(pkg, ref_ctx, r, tp)
        want.append(pkg.chain_step(ev, props, thetas[b], b % 2, z=zs[b]))
        ev.close(); [p.close() for p in props]
    ref_ctx.close()
    lead = ctxs[0]
    tickets = [pkg.BatchedStepTicket([objs[b][0]], [objs[b][1]], [thetas[b]], [b % 2], z=[zs[b]], launch_ctx=lead) for b in range(MAX_BATCHES)]
    last = MAX_BATCHES
    with pytest.raises(pkg._native.IcpNativeError) as ei:
        pkg.BatchedStepTicket([objs[last][0]], [objs[last][1]], [thetas[last]], [last % 2], z=[zs[last]], launch_ctx=lead)
    assert ei.value.status == ICP_ERR_BUSY
    assert ctxs[last].transformedMesh(thetas[last]).shape == (model.n_points, 3)  # (the refused ticket's context was not left busy)
    results = [tickets[0].collect()]
    tickets.append(pkg.BatchedStepTicket([objs[last][0]], [objs[last][1]], [thetas[last]], [last % 2], z=[zs[last]], launch_ctx=lead))
    results += [t.collect() for t in tickets[1:]]
    for b, (out, val, fwd, bwd, status) in enumerate(results):
        assert list(status) == [0]
        assert np.abs(out[0] - want[b][0]).max() <= 1e-9 and abs(val[0] - want[b][1]) <= 1e-9 * abs(want[b][1])
        assert np.allclose(fwd[0], want[b][2], rtol=1e-9, atol=0) and np.allclose(bwd[0], want[b][3], rtol=1e-9, atol=0)
    for ev, props in objs:
        ev.close(); [p.close() for p in props]
    for c in ctxs:
        c.close()


_GROUPS_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.load_femur_model_and_target(50)
B = {B}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=77 + i) for i in range(B)]
rec = pkg.run_chains_batched(chains, {n})
np.savez({out!r}, rec=np.stack(rec), stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


@pytest.mark.parametrize("groups", ["5", "8", "12"])
def test_more_than_four_lockstep_groups_match_chains_stepped_alone(pkg, groups, tmp_path):
    """Round 4 raised the harness' group limit from 4 to 8 while every group submits through groups[0]'s launch context, whose rings
    held 4 batches: with 5..8 groups ticket N + 4 rewrote ticket N's pinned arguments and eigen records.  The rings now hold
    ICP_MAX_BATCHES_IN_FLIGHT = 8 and the harness never makes more groups than that (a forced 12 becomes 8): 16 chains in 5, 8 and
    "12" groups give, chain by chain, the records of the same chain stepped alone; no fall-back counter moves."""
    B, n = 16, 24
    path = str(tmp_path / f"g{groups}.npz")
    subprocess.run([sys.executable, "-c", _GROUPS_SCRIPT.format(root=ROOT, B=B, n=n, out=path)], check=True,
                   env={**os.environ, "ICP_LOCKSTEP_GROUPS": groups, "ICP_HOST_DEVICE_LOOP": "0"}, timeout=900)
    got = np.load(path)
    assert np.all(got["stats"] == 0), got["stats"]
    model, target = pkg.data.load_femur_model_and_target(50)
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    for i in (0, 3, 7, 12, 15):
        ctx = pkg.IcpContext(model, target, device=0)
        ch = pkg.SamplingRegistration(ctx, setup, pkg.random_initial_parameters(model, i), seed=77 + i)
        want = pkg.run_chains_batched([ch], n)[0]
        assert np.array_equal(got["rec"][i], want), i
        ch.close(); ctx.close()


_POOL_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.load_femur_model_and_target(50)
r = model.rank
theta = pkg.initial_parameters(model); theta[10:] = 0.25 * np.random.default_rng(3).normal(size=r)
z = np.random.default_rng(4).normal(size=r)
out = []
for rep in range(2):   # the first context's buffers go to the pool when it is destroyed; the second context allocates with the pool full
    ctx = pkg.IcpContext(model, target, device=0)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True)
    out.append(prop.propose(theta, z))
    prop.close(); ctx.close()
    if rep == 0:       # a model of another rank: none of the kept blocks fits, every allocation is a fresh hipMalloc
        model2, target2 = pkg.data.load_femur_model_and_target(100)
        ctx2 = pkg.IcpContext(model2, target2, device=0)
        p2 = pkg.NonRigidIcpProposal(ctx2, 0.1, 10.0, 5.0, 64, pkg.ModelSampling, True)
        th2 = pkg.initial_parameters(model2)
        out2 = p2.propose(th2, np.zeros(model2.rank))
        p2.close(); ctx2.close()
np.savez({out!r}, a=out[0], b=out[1], c=out2)
"""


def test_device_pool_is_drained_and_the_allocation_retried(pkg, tmp_path):
    """the pool of freed device buffers keeps blocks by exact size; when sizes change between jobs the kept blocks fit
    nothing and an allocation that fails must give them back to the runtime and try again instead of failing with gigabytes idle.
    Test-hooks build: the 600th fresh hipMalloc of the process — in the middle of the second model's context, with the first model's
    blocks in the pool — reports out-of-memory once.  The run must complete with the results of an undisturbed run."""
    hooks = os.path.join(ROOT, "icp-proposal_amd", "libicp_proposal_amd_testhooks.so")
    assert os.path.exists(hooks), "build the test-hooks library (python -c 'import __graft_entry__ as g; g.build()')"
    res = {}
    for tag, env in (("plain", {}), ("failing", {"ICP_TEST_FAIL_MALLOC_AT": "600"})):
        path = str(tmp_path / (tag + ".npz"))
        done = subprocess.run([sys.executable, "-c", _POOL_SCRIPT.format(root=ROOT, out=path)], check=True, capture_output=True, text=True,
                              env={**os.environ, "ICP_LIBRARY_PATH": hooks, **env}, timeout=600)
        res[tag] = np.load(path)
        if env:  # the forced failure happened, and with blocks in the pool (they were given back before the retry)
            hook = [ln for ln in done.stderr.splitlines() if "[icp test hook] hipMalloc" in ln]
            assert hook and int(hook[0].split("gave back")[1].split()[0]) > 0, done.stderr[-2000:]
    for k in ("a", "b", "c"):
        assert np.array_equal(res["plain"][k], res["failing"][k]), k
    assert np.array_equal(res["plain"]["a"], res["plain"]["b"])


def _rotation(pkg, angles):
    """the library's own convention, through the oracle's copy of include/icp_sincos.h (Rz(phi)·Ry(theta)·Rx(psi))"""
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    return np.asarray(O.rotation_matrix(*angles), dtype=np.float64).reshape(3, 3)


_CONVENTION_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import oracle as O
model, target = pkg.data.load_femur_model_and_target(50)
def make_setup():
    s = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    s.pose_rot_sigma = (0.01, 0.012, 0.008); s.pose_trans_sigma = (0.1, 0.15, 0.08); s.rw_sigma = 0.02
    return s
B = {B}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
starts = [pkg.random_initial_parameters(model, i) for i in range(B)]
rng = np.random.default_rng(8)
for i, cx in enumerate(ctxs):
    # what the Scala adapter does before every call: the host's own matrix for the theta's angles — here the oracle's (the library's
    # convention, from the shared sine/cosine), plus a few more triples a chain's first host-stepped states would have registered
    for k, ang in enumerate([starts[i][4:7]] + [0.05 * rng.normal(size=3) for _ in range(3)]):
        R = np.asarray(O.rotation_matrix(*ang), dtype=np.float64).reshape(3, 3)
        if {mismatch} and i == 0 and k > 0:
            R = R.T.copy()      # another convention (the inverse rotation): still a rotation, but not Rz·Ry·Rx
        cx.setRotation(ang, R)
conv = [cx.rotationConvention() for cx in ctxs]
chains = [pkg.SamplingRegistration(ctxs[i], make_setup(), starts[i], seed=500 + i) for i in range(B)]
rec = pkg.run_chains_batched(chains, {n})
paths = [c.step_paths() for c in ctxs]
np.savez({out!r}, rec=np.stack(rec), loop=np.array([p["device_loop"] for p in paths]), other=np.array([p["merged"] + p["wide"] + p["per_stage"] for p in paths]),
         verified=np.array([c["verified"] for c in conv]), mismatched=np.array([c["mismatched"] for c in conv]),
         stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


def _oracle_chain_config(oracle, setup):
    from test_gpu_chain import oracle_chain_config
    return oracle_chain_config(oracle, setup)


@pytest.mark.parametrize("mismatch", [False, True])
def test_registered_rotations_of_the_librarys_convention_keep_the_device_loop_open(pkg, femur50, femur50_oracle, oracle, mismatch, tmp_path):
    """the Scala adapters register Scalismo's matrix before every call, and a context with registered matrices was
    refused by the on-device loop for mixtures with pose walks.  Now every registered matrix is compared with the library's Rz·Ry·Rx:
    contexts whose matrices all agreed (here: the oracle's matrices) run the pose mixture of apps/bfm/BfmFittingPartial.scala:70 inside
    icp_chains_run_on_device, decision for decision the oracle's chain; with ONE disagreeing matrix (the transpose) on one context the
    loop refuses the run and the harness steps the same chains on the host — the same records, since that triple never occurs."""
    model, target = femur50
    om, ot = femur50_oracle
    B, n = 3, 60
    path = str(tmp_path / "conv.npz")
    script = _CONVENTION_SCRIPT.format(root=ROOT, B=B, n=n, out=path, mismatch=repr(bool(mismatch)))
    subprocess.run([sys.executable, "-c", script], check=True, env={**os.environ, "ICP_HOST_DEVICE_LOOP": "1"}, timeout=900)
    got = np.load(path)
    if mismatch:
        assert got["mismatched"][0] == 3 and np.all(got["mismatched"][1:] == 0)
        assert np.all(got["loop"] == 0) and np.all(got["other"] == n), "a context with a foreign convention must not enter the device loop"
    else:
        assert np.all(got["verified"] == 4) and np.all(got["mismatched"] == 0)
        assert np.all(got["loop"] == n) and np.all(got["other"] == 0), "the chains did not run inside the on-device loop"
    assert np.all(got["stats"] == 0)
    setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    setup.pose_rot_sigma = (0.01, 0.012, 0.008); setup.pose_trans_sigma = (0.1, 0.15, 0.08); setup.rw_sigma = 0.02
    cfg = _oracle_chain_config(oracle, setup)
    leaves = set()
    for b in range(B):
        acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, cfg, pkg.random_initial_parameters(model, b), 500 + b, n)
        rec = got["rec"][b]
        assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), f"chain {b}: accept/reject sequences differ"
        assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), f"chain {b}: mixture components differ"
        assert np.abs(rec[:, 14:] - states_o[:, 10:]).max() <= 1e-5 * np.abs(states_o[:, 10:]).max()
        assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
        leaves |= set(comp_o.tolist())
    assert leaves & {3, 4, 5, 6, 7, 8} and 0 in leaves


def test_mixture_struct_size_is_checked(pkg, femur50):
    """icp_mh_mixture grew in round 4 (pose walks): struct_size must be sizeof(icp_mh_mixture) of this header, anything else is refused
    before the structure is read any further."""
    model, target = femur50
    r = model.rank
    nat = pkg._native
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    ctx = pkg.IcpContext(model, target, device=0)
    ev, props = _chain_objects(pkg, ctx, r, tp)
    theta = pkg.initial_parameters(model)
    C = ctypes
    for size, want in ((40, ICP_ERR_INVALID_ARG), (96, ICP_ERR_INVALID_ARG), (C.sizeof(nat.MhMixture), 0)):
        mix = nat.MhMixture(size, (C.c_double * 2)(0.5, 0.5), 0.9, 0.1, 0.1)
        evs = (C.c_void_p * 1)(ev.h)
        prs = (C.c_void_p * 2)(props[0].h, props[1].h)
        seeds = (C.c_uint64 * 1)(9)
        first = (C.c_int64 * 1)(0)
        th = np.ascontiguousarray(theta, dtype=np.float64).copy()
        thp = (nat.c_double_p * 1)(th.ctypes.data_as(nat.c_double_p))
        logp = np.array([ -1e300 ])
        acc = (C.c_int64 * 1)(0)
        rc = nat.lib().icp_chains_run_on_device(1, evs, 2, prs, C.byref(mix), seeds, first, thp, logp.ctypes.data_as(nat.c_double_p), 3, None, acc)
        assert rc == want, (size, rc, nat.lib().icp_last_error())
        if want:
            assert b"struct_size" in nat.lib().icp_last_error()
    ev.close(); [p.close() for p in props]; ctx.close()
