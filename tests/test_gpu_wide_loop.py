"""GPU: the on-device Metropolis–Hastings loop for chains that take the WIDE step (SURVEY.md §8f row 4's remainder: open targets, the
full-mesh Hausdorff evaluator, ranks 65..200, pose walks — apps/bfm/BfmFittingPartial.scala:62-96 — and the femur mixture at ranks
65..116, apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala).  icp_chains_run_on_device replays the wide step's own launches from
device-resident records; mixture draw, proposals' inputs, MetropolisHastings.next and the records are kernels of the step's stream.

Compared (a) with the SAME chains stepped by the host harness through icp_chain_step_batched — identical records wherever the
decomposition has no warm start (ranks above 64), identical decisions and states to 1e-8 where it has (the host-stepped wide step
decomposes ahead, also states that are then rejected: its Jacobi iteration starts from another basis) —, (b) with the oracle's chain
(orc_run_chain) decision for decision, at a reduced size over 80 steps and at the full configs[3] size.
The harness reads ICP_HOST_DEVICE_LOOP once per process: every run is a process of its own."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT

ICP_ERR_INVALID_ARG = -1

pytestmark = pytest.mark.gpu

_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
kind = {kind!r}
if kind == "face100":       # open target, 0.4 pose walks + 0.55 ICP (ModelSampling, boundary-aware) + 0.05 shape walk, collective evaluator
    model = pkg.data.synthetic_face_model(grid=41, rank=100)
    target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
    mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
elif kind == "hausdorff":   # the same with the full-mesh Hausdorff evaluator
    model = pkg.data.synthetic_face_model(grid=41, rank=100)
    target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
    mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="hausdorff", fused=2)
elif kind == "face200":     # rank 200: the four-slot tridiagonalisation, two launches of decompositions side by side
    model = pkg.data.synthetic_face_model(grid=41, rank=200)
    target = pkg.data.synthetic_partial_target(model, n_remove=90, seed=7)
    mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
elif kind == "facefull":    # configs[3] / configs[4] size (N = 28,561, rank 200): the instance launch's head is a thirtieth of its blocks
    model = pkg.data.synthetic_face_model()
    target = pkg.data.synthetic_partial_target(model, seed=100)
    mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
elif kind == "face40":      # rank <= 64 on an open target: the warm-started Jacobi iteration behind the decision, three groups
    model = pkg.data.synthetic_face_model(grid=31, rank=40)
    target = pkg.data.synthetic_partial_target(model, n_remove=60, seed=7)
    mk = lambda: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
elif kind == "femur100":    # closed target, two ICP directions + shape walk at rank 101: the five merged launches host-stepped
    model, target = pkg.data.load_femur_model_and_target(100)
    mk = lambda: pkg.femur_icp_proposal_registration(model, target, fused=2)
elif kind == "femur50open": # TWO ICP directions (ModelSampling + TargetSampling) against a target WITH boundary, rank 51
    sys.path.insert(0, {root!r} + "/tests")
    from conftest import open_patch_target
    model, closed = pkg.data.load_femur_model_and_target(50)
    pts, cells = open_patch_target(closed)
    target = pkg.data.TriangleMesh(pts, cells)
    mk = lambda: pkg.femur_icp_proposal_registration(model, target, fused=2)
elif kind == "face40root":  # the opt-in Cholesky-root sampler inside the wide loop (ranks <= 64)
    model = pkg.data.synthetic_face_model(grid=31, rank=40)
    target = pkg.data.synthetic_partial_target(model, n_remove=60, seed=7)
    def mk():
        s = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2); s.sampler = "cholesky-root"; return s
B, n1, n2 = {B}, {n1}, {n2}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], mk(), pkg.random_initial_parameters(model, i), seed=300 + i) for i in range(B)]
a = pkg.run_chains_batched(chains, n1)
single = chains[0].run(6)                       # a chain goes on by itself (host-stepped) from where the loop left it …
b = pkg.run_chains_batched(chains[1:], n2)      # … and the others through a second run
states = [c.state() for c in chains]
paths = [c.step_paths() for c in ctxs]
np.savez({out!r}, a=np.stack(a), single=single, b=np.stack(b), theta=np.stack([s[0] for s in states]), logp=np.array([s[1] for s in states]),
         n=np.array([s[2] for s in states]), acc=np.array([s[3] for s in states]), loop=np.array([p["device_loop"] for p in paths]),
         stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


@pytest.mark.parametrize("kind,B,n1,n2,exact", [("face100", 4, 70, 12, True), ("hausdorff", 3, 40, 8, True), ("face200", 18, 30, 6, True),
                                                ("face40", 13, 140, 10, False), ("femur100", 3, 70, 12, True),
                                                ("femur50open", 3, 60, 10, False), ("face40root", 3, 60, 10, True)])
def test_wide_loop_matches_host_stepped_chains(kind, B, n1, n2, exact, tmp_path):
    out = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"wl_{mode}.npz")
        script = _SCRIPT.format(root=ROOT, kind=kind, B=B, n1=n1, n2=n2, out=path)
        subprocess.run([sys.executable, "-c", script], check=True, env={**os.environ, "ICP_HOST_DEVICE_LOOP": mode}, timeout=900)
        out[mode] = np.load(path)
    dev, host = out["1"], out["0"]
    assert np.all(dev["loop"][1:] == n1 + n2) and dev["loop"][0] == n1 and np.all(host["loop"] == 0)
    assert np.all(dev["stats"] == 0) and np.all(host["stats"] == 0)
    for key in ("a", "single", "b"):
        d, h = dev[key], host[key]
        assert d.shape == h.shape
        assert np.array_equal(d[..., :3], h[..., :3]), key + ": index / decision / mixture component"
        if exact:
            assert np.array_equal(d, h), key
        else:
            dmax, scale = np.abs(d[..., 4:] - h[..., 4:]).max(), np.abs(h[..., 14:]).max()
            assert dmax <= 1e-8 * scale, (key, dmax, scale)
            pmax, pscale = np.abs(d[..., 3] - h[..., 3]).max(), np.abs(h[..., 3]).max()
            assert pmax <= 1e-8 * pscale, (key, pmax, pscale)
    assert np.array_equal(dev["n"], host["n"]) and np.array_equal(dev["acc"], host["acc"])
    a = dev["a"]
    assert a.shape == (B, n1, 14 + model_rank(kind)) and np.array_equal(a[:, :, 0], np.tile(np.arange(n1), (B, 1)))
    assert 0.05 < a[:, :, 1].mean() < 0.97
    leaves = set(a[:, :, 2].astype(int).ravel())
    assert leaves == ({0, 1, 2} if kind.startswith("femur") else {0, 2, 3, 4, 5, 6, 7, 8}), leaves



def test_wide_loop_layouts_give_the_same_records(tmp_path):
    """Round 6 changed WHERE the loop's launches go, not what they compute: the instance launch's head ahead of the rest
    (ICP_WIDE_LOOP_HEAD_SPLIT), the decompositions in two parts around the decision (ICP_WIDE_LOOP_EIG_SPLIT), the step's critical chain on
    one queue (ICP_WIDE_LOOP_CHAIN_MAIN).  Test-hooks build: each switched back in turn, six chains of the full-size face model (where the
    head applies: it is at most a quarter of the instance's blocks) — records bit for bit."""
    hooks = os.path.join(ROOT, "icp-proposal_amd", "libicp_proposal_amd_testhooks.so")
    assert os.path.exists(hooks), "build the test-hooks library (python -c 'import __graft_entry__ as g; g.build()')"
    out = {}
    for tag, env in (("default", {}), ("no_head", {"ICP_WIDE_LOOP_HEAD_SPLIT": "0"}), ("no_chain_main", {"ICP_WIDE_LOOP_CHAIN_MAIN": "0"}),
                     ("no_eig_split", {"ICP_WIDE_LOOP_EIG_SPLIT": "0"})):
        path = str(tmp_path / f"wl_{tag}.npz")
        script = _SCRIPT.format(root=ROOT, kind="facefull", B=6, n1=16, n2=4, out=path)
        subprocess.run([sys.executable, "-c", script], check=True, timeout=900,
                       env={**os.environ, "ICP_HOST_DEVICE_LOOP": "1", "ICP_LIBRARY_PATH": hooks, **env})
        out[tag] = np.load(path)
    ref = out["default"]
    assert np.all(ref["loop"][1:] == 16 + 4) and np.all(ref["stats"] == 0)
    for tag, got in out.items():
        for key in ("a", "single", "b", "theta", "logp"):
            assert np.array_equal(ref[key], got[key]), (tag, key)

def model_rank(kind):
    return {"face100": 100, "hausdorff": 100, "face200": 200, "face40": 40, "femur100": 101, "femur50open": 51, "face40root": 40}[kind]


_ORACLE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
size = {size!r}
if size == "small":
    model = pkg.data.synthetic_face_model(grid=41, rank=100)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
else:                       # configs[3] / configs[4]: N = 28,561, rank 200, K = 400, K_e = 800
    model = pkg.data.synthetic_face_model()
    target = pkg.data.synthetic_partial_target(model)
setup = pkg.bfm_fitting_partial(model, target, evaluator={evaluator!r}, fused=2)
setup.pose_rot_sigma, setup.pose_trans_sigma = (0.02, 0.01, 0.004), (0.2, 0.1, 0.05)
B = {B}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
theta0 = [pkg.initial_parameters(model) if i == 0 else pkg.random_initial_parameters(model, i) for i in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], setup, theta0[i], seed=77 + i) for i in range(B)]
rec = pkg.run_chains_batched(chains, {n})
np.savez({out!r}, rec=np.stack(rec), theta0=np.stack(theta0), loop=np.array([c.step_paths()["device_loop"] for c in ctxs]),
         stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


@pytest.mark.parametrize("size,evaluator,B,n", [("small", "collective", 2, 80), ("small", "hausdorff", 2, 60), ("full", "collective", 2, 10)])
def test_wide_loop_matches_oracle_chain(pkg, oracle, size, evaluator, B, n, tmp_path):
    """apps/bfm/BfmFittingPartial.scala:62-96 inside icp_chains_run_on_device against orc_run_chain (the same counter-based random
    numbers bit for bit): every decision and mixture component identical, states within 1e-5, log values within 1e-6 — on a target
    WITH boundary at rank 100 over 60-80 steps (unequal pose sigmas: a wrong pairing of walk and parameter changes the chain), and at
    the full size of configs[3] (rank 200: the decompositions ahead of the decision, their bases handed over on acceptance)."""
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_chain import oracle_chain_config
    from test_gpu_face import compare_chain_with_oracle
    from conftest import oracle_chains_parallel
    path = str(tmp_path / "wlo.npz")
    script = _ORACLE_SCRIPT.format(root=ROOT, size=size, evaluator=evaluator, B=B, n=n, out=path)
    # (the GPU chains in a process of their own — the harness reads ICP_HOST_DEVICE_LOOP once — WHILE the oracle's chains run here, a
    # thread each)
    gpu = subprocess.Popen([sys.executable, "-c", script], env={**os.environ, "ICP_HOST_DEVICE_LOOP": "1"})
    try:
        if size == "small":
            model = pkg.data.synthetic_face_model(grid=41, rank=100)
            target = pkg.data.synthetic_partial_target(model, n_remove=90)
        else:
            model = pkg.data.synthetic_face_model()
            target = pkg.data.synthetic_partial_target(model)
        om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
        setup = pkg.bfm_fitting_partial(model, target, evaluator=evaluator, fused=2)
        setup.pose_rot_sigma, setup.pose_trans_sigma = (0.02, 0.01, 0.004), (0.2, 0.1, 0.05)
        theta0 = [pkg.initial_parameters(model) if i == 0 else pkg.random_initial_parameters(model, i) for i in range(B)]  # (= the script's)
        cfg = oracle_chain_config(oracle, setup)
        want = oracle_chains_parallel(oracle, [(om, ot, cfg, theta0[i], 77 + i, n) for i in range(B)])
    finally:
        assert gpu.wait(timeout=900) == 0
    got = np.load(path)
    assert np.all(got["loop"] == n) and np.all(got["stats"] == 0)
    assert np.array_equal(got["theta0"], np.stack(theta0))
    seen = set()
    n_acc = 0
    for i in range(B):
        acc_o, comp_o, logp_o, states_o = want[i]
        compare_chain_with_oracle(got["rec"][i], acc_o, comp_o, logp_o, states_o)
        seen |= set(comp_o.tolist())
        n_acc += int(acc_o.sum())
    assert n_acc > 0 and 0 in seen and any(k >= 3 for k in seen)
    if size == "small":
        assert {0, 2} <= seen and len(seen & {3, 4, 5, 6, 7, 8}) >= 4


def test_wide_loop_refusals(pkg):
    """What the wide loop does not take says so and leaves the chains alone (the harness then steps them itself): the Cholesky-root
    sampler above rank 64, chains on different models in one run."""
    import ctypes as C
    nat = pkg._native
    model = pkg.data.synthetic_face_model(grid=31, rank=80)
    other = pkg.data.synthetic_face_model(grid=31, rank=80, seed=5)
    target = pkg.data.synthetic_partial_target(model, n_remove=60, seed=7)
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 4 * r)

    def objects(ctx):
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, pkg.ModelSampling, True)
        ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.0, 1.0, 1.0, pkg.SymmetricEvaluation, 4 * r, decimatedTargetPoints=tp)
        return ev, prop

    def run(evs, props, thetas):
        B = len(evs)
        mix = nat.MhMixture(C.sizeof(nat.MhMixture), (C.c_double * 2)(1.0, 0.0), 0.9, 0.1, 0.1)
        eh = (C.c_void_p * B)(*[e.h for e in evs])
        ph = (C.c_void_p * B)(*[p.h for p in props])
        seeds = (C.c_uint64 * B)(*range(9, 9 + B))
        first = (C.c_int64 * B)(*([0] * B))
        th = [np.ascontiguousarray(t, dtype=np.float64).copy() for t in thetas]
        thp = (nat.c_double_p * B)(*[t.ctypes.data_as(nat.c_double_p) for t in th])
        logp = np.full(B, -1e300)
        acc = (C.c_int64 * B)(*([0] * B))
        rc = nat.lib().icp_chains_run_on_device(B, eh, 1, ph, C.byref(mix), seeds, first, thp, logp.ctypes.data_as(nat.c_double_p), 3, None, acc)
        return rc, nat.lib().icp_last_error().decode(), list(acc)

    ctxs = [pkg.IcpContext(model, target, device=0), pkg.IcpContext(model, target, device=0), pkg.IcpContext(other, target, device=0)]
    objs = [objects(c) for c in ctxs]
    thetas = [pkg.random_initial_parameters(model, i) for i in range(3)]
    rc, msg, _ = run([objs[0][0], objs[2][0]], [objs[0][1], objs[2][1]], [thetas[0], thetas[2]])
    assert rc == ICP_ERR_INVALID_ARG and "model" in msg, (rc, msg)
    for _, p in objs[:2]:
        p.setSampler("cholesky-root")
    rc, msg, _ = run([objs[0][0], objs[1][0]], [objs[0][1], objs[1][1]], thetas[:2])
    assert rc == ICP_ERR_INVALID_ARG and "Cholesky-root" in msg, (rc, msg)
    for _, p in objs[:2]:
        p.setSampler("eigen")
    rc, msg, acc = run([objs[0][0], objs[1][0]], [objs[0][1], objs[1][1]], thetas[:2])   # … and what it does take, on the same objects
    assert rc == 0, (rc, msg)
    assert all(c.step_paths()["device_loop"] == 3 for c in ctxs[:2]) and ctxs[2].step_paths()["device_loop"] == 0
    for ev, p in objs:
        ev.close(); p.close()
    for c in ctxs:
        c.close()
