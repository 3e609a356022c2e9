"""GPU: the C++ host harness (SamplingRegistration mirror over the C ABI) against the oracle's chain.

Same counter-based random numbers on both sides => the accept/reject sequence and the chosen mixture components
must be IDENTICAL and the chain states equal within 1e-5 relative."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def oracle_chain_config(oracle, setup):
    icp = []
    for p in setup.icp:
        icp.append(oracle.proposal_params(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], p.get("boundary_aware", True),
                                          n_model_ids=p.get("n_model_ids", 0), target_pts=p.get("target_pts")))
    e = setup.eval
    ep = oracle.evaluator_params(e["kind"], e["mode"], n_model_ids=e["n_model_ids"], target_pts=e["target_pts"],
                                 p0=e["gauss_mean"] if e["kind"] != 1 else e["exp_rate"], p1=e["gauss_sigma"], p2=e["exp_rate"])
    return oracle.chain_config(icp, [p.get("weight", 0.5) for p in setup.icp], setup.w_icp, setup.w_rw, setup.rw_sigma, ep,
                               w_pose=setup.w_pose, pose_rot_sigma=setup.pose_rot_sigma, pose_trans_sigma=setup.pose_trans_sigma)


@pytest.mark.parametrize("fused", [2, 1, 0])
def test_femur50_chain_matches_oracle(pkg, femur50, femur50_oracle, oracle, fused):
    model, target = femur50
    om, ot = femur50_oracle
    n_steps, seed = 60, 1024
    setup = pkg.femur_icp_proposal_registration(model, target, fused=fused)
    theta0 = pkg.initial_parameters(model)
    acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n_steps)
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n_steps)
    assert np.array_equal(rec[:, 0], np.arange(n_steps))
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), "mixture components differ"
    assert acc_o.sum() > 5 and (comp_o == 2).sum() > 0 and (comp_o == 0).sum() > 0 and (comp_o == 1).sum() > 0
    scale = np.abs(states_o[:, 10:]).max()
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
    theta, logp, n, a = chain.state()
    assert n == n_steps and a == acc_o.sum()
    chain.close()
    ctx.close()


def test_femur100_all_points_symmetric_matches_oracle(pkg, oracle):
    """BASELINE.json configs[2] (RunMHRandomInitComparison analogue): femur-100 GPMM (rank 101), every model point a sample
    point of proposal and evaluator (K = N = 1622), ModelSampling ICP mixture, symmetric evaluation, random initial shape.

    With 1622 correspondences the ICP posterior is so narrow that the reference's transition ratio rejects almost every
    step (identical on both sides); the step's ingredients are therefore compared one by one as well."""
    model, target = pkg.data.load_femur_model_and_target(100)
    assert model.rank == 101
    r, n = model.rank, model.n_points
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    setup = pkg.femur_random_init_comparison(model, target)
    theta0 = pkg.random_initial_parameters(model, chain_index=3)
    ctx = pkg.IcpContext(model, target, device=0)
    # ---- one step, ingredient by ingredient, through the merged-launch entry point and through the per-method ones
    pp = oracle.proposal_params(0.1, 10.0, 5.0, oracle.MODEL_SAMPLING, True, n_model_ids=n)
    e = setup.eval
    ep = oracle.evaluator_params(e["kind"], e["mode"], n_model_ids=e["n_model_ids"], target_pts=e["target_pts"],
                                 p0=e["gauss_mean"], p1=e["gauss_sigma"], p2=e["exp_rate"])
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, n, "ModelSampling", True)
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 2, n, decimatedTargetPoints=e["target_pts"])
    rng = np.random.default_rng(5)
    for trial in range(2):
        theta = theta0.copy()
        theta[10:] += 0.2 * rng.normal(size=r)
        z = rng.normal(size=r)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.corr_point, po.corr_pt)   # bit-exact
        assert np.abs(post.alpha - po.alpha).max() <= 1e-9 * np.abs(po.alpha).max()
        assert np.abs(post.S - po.S).max() <= 1e-9 * np.abs(po.S).max()
        want = oracle.propose(om, ot, pp, theta, z)
        got, val, fwd, bwd = pkg.chain_step(ev, [prop], theta, generator=0, z=z)
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want[10:]).max()
        lv, rc = oracle.evaluator_log_value(om, ot, ep, want)
        assert rc == 0 and abs(val - lv) <= 1e-7 * abs(lv)        # the states differ by ~1e-13
        lf, lb = oracle.log_transition(om, ot, pp, theta, want), oracle.log_transition(om, ot, pp, want, theta)
        assert abs(fwd[0] - lf) <= 1e-6 * abs(lf) and abs(bwd[0] - lb) <= 1e-6 * abs(lb)
        # (the merged launch and the per-method kernel sum the tail's products over 1,024 and over 256 threads at this rank: the same
        # value to the last bit or two, not always to the last)
        assert abs(prop.logTransitionProbability(theta, got) - fwd[0]) <= 1e-13 * abs(fwd[0]) and ev.logValue(got) == val
    prop.close()
    ev.close()
    # ---- a short chain: identical decisions
    n_steps, seed = 6, 1031
    acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n_steps)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n_steps)
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o)
    scale = np.abs(states_o[:, 10:]).max()
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
    chain.close()
    ctx.close()


_SPECULATION_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.load_femur_model_and_target(50)
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), 1024)
np.save({out!r}, chain.run(60))
chain.close(); ctx.close()
"""


@pytest.mark.parametrize("env", [{"ICP_SPECULATION": "1"}, {"ICP_SPECULATION": "1", "ICP_TEST_STARVE_SPECULATION": "1"},
                                 {"ICP_NO_PIPELINE": "1"}, {"ICP_TEST_STARVE_PIPELINE": "1"}],
                         ids=["speculation-on", "speculation-starved", "pipeline-off", "pipeline-starved"])
def test_speculative_decomposition_fallbacks(pkg, femur50, femur50_oracle, oracle, env, tmp_path):
    """icp_chain_step starts the KL basis of the proposed state before the caller decides (icp_abi.hip, speculate_eigen;
    ICP_SPECULATION=1: always, unset: adaptive).  It must give the same chain switched on, and starved — the decomposition never
    sees its input, gives up after its time-out, and the step that drew from it is repeated with an ordinary one.
    Likewise the two-stream step pipeline (on by default): switched off, and starved — a first launch times out on the
    word it waits for, and the context falls back to unpipelined steps."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    model, target = femur50
    om, ot = femur50_oracle
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), pkg.initial_parameters(model), 1024, 60)
    out = str(tmp_path / "rec.npy")
    script = _SPECULATION_SCRIPT.format(root=ROOT, out=out)
    if any(k.startswith("ICP_TEST_") for k in env):
        # the starvation hooks exist only in the test-hooks build of the library (csrc/Makefile: -DICP_TEST_HOOKS); the shipped
        # library does not read these variables
        hooks = os.path.join(ROOT, "icp-proposal_amd", "libicp_proposal_amd_testhooks.so")
        assert os.path.exists(hooks), "build the test-hooks library (python -c 'import __graft_entry__ as g; g.build()')"
        env = {**env, "ICP_LIBRARY_PATH": hooks}
    subprocess.run([sys.executable, "-c", script], check=True, env={**os.environ, **env}, timeout=300)
    rec = np.load(out)
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), "mixture components differ"
    scale = np.abs(states_o[:, 10:]).max()
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()


_PIPELINE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.synthetic_femur_target()
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), 1024)
np.save({out!r}, np.concatenate([chain.run(n) for n in (300, 1, 2, 297)]))
chain.close(); ctx.close()
"""


def test_pipelined_steps_are_bit_identical_to_unpipelined(tmp_path):
    """BASELINE.json configs[1] (the metric configuration), 600 steps in four runs: the default chain step — steps
    alternating between two streams, the next step's first four launches issued ahead under the rejection assumption and
    dropped on acceptance, decompositions in pairs on a third stream — must reproduce the records of the
    unpipelined one (ICP_NO_PIPELINE=1) bit for bit: any ordering hazard between the streams would show here."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    recs = []
    for tag, env in (("pipelined", {}), ("plain", {"ICP_NO_PIPELINE": "1"})):
        out = str(tmp_path / (tag + ".npy"))
        subprocess.run([sys.executable, "-c", _PIPELINE_SCRIPT.format(root=ROOT, out=out)], check=True,
                       env={**os.environ, **env}, timeout=300)
        recs.append(np.load(out))
    assert recs[0].shape == (600, recs[0].shape[1]) and recs[0][:, 1].sum() > 100
    assert np.array_equal(recs[0], recs[1])


def test_metric_configuration_chain_matches_oracle(pkg, oracle):
    """BASELINE.json configs[1] — the configuration bench.py times (femur-50 model against the 58,322-vertex synthetic
    target, pipelined chain step): accept/reject sequence and mixture components identical to the oracle's chain, states
    within 1e-5 relative, over 48 steps (the oracle's brute-force scan of 116,640 triangles runs at ~4 steps/s)."""
    model, target = pkg.data.synthetic_femur_target()
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    n_steps, seed = 48, 1024
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    theta0 = pkg.initial_parameters(model)
    acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n_steps)
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = np.concatenate([chain.run(20), chain.run(28)])
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), "mixture components differ"
    assert acc_o.sum() > 5
    scale = np.abs(states_o[:, 10:]).max()
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
    chain.close()
    ctx.close()


def test_batched_chains_are_bit_identical_to_single_chains(pkg, femur50):
    """icp_chain_step_batched (B chains per launch): chains stepped in lockstep through ONE sequence of
    launches per step give, chain by chain, exactly the records of the same chains run on their own — including a chain
    with pose proposals (those steps leave the batch) and a batch whose members differ in their proposal kind per step."""
    model, target = femur50
    n_steps, B = 50, 9   # (from 8 chains on the runner keeps two groups in flight, half a step apart)
    setups = [pkg.femur_icp_proposal_registration(model, target, fused=2) for _ in range(B)]
    inits = [pkg.initial_parameters(model)] + [pkg.random_initial_parameters(model, chain_index=i) for i in range(1, B)]

    def make():
        ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
        return ctxs, [pkg.SamplingRegistration(ctxs[i], setups[i], inits[i], seed=1024 + i) for i in range(B)]

    ctxs, chains = make()
    single = [c.run(n_steps) for c in chains]
    [c.close() for c in chains]; [c.close() for c in ctxs]
    ctxs, chains = make()
    first = pkg.run_chains_batched(chains, 20)
    second = pkg.run_chains_batched(chains, n_steps - 20)  # (a second call continues the chains)
    for b in range(B):
        got = np.vstack([first[b], second[b]])
        assert np.array_equal(got, single[b]), f"chain {b} differs"
        assert single[b][:, 1].sum() > 3
    # a small batch (one group), and a chain that alternates between batched and single stepping
    more_b = pkg.run_chains_batched(chains[:3], 5)[0]
    more_s = chains[0].run(5)
    [c.close() for c in chains]; [c.close() for c in ctxs]
    ctx = pkg.IcpContext(model, target, device=0)
    ref = pkg.SamplingRegistration(ctx, setups[0], inits[0], seed=1024)
    want = ref.run(n_steps + 10)
    assert np.array_equal(np.vstack([more_b, more_s]), want[n_steps:])
    ref.close(); ctx.close()


def test_chain_step_batched_entry_point(pkg, femur50):
    """The C entry point itself: per chain the values of icp_chain_step; a second chain on the same context and a chain
    with a generator-less (host-made) proposal are accepted and give the same values."""
    model, target = femur50
    r = model.rank
    rng = np.random.default_rng(5)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    B = 3
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]

    def objects(ctx):
        props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.TargetSampling, True, decimatedTargetPoints=tp),
                 pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True, decimatedTargetPoints=tp)]
        ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r, decimatedTargetPoints=tp)
        return ev, props

    objs = [objects(c) for c in ctxs]
    thetas = []
    for b in range(B):
        t = pkg.initial_parameters(model)
        t[10:] = 0.3 * rng.normal(size=r)
        thetas.append(t)
    zs = [rng.normal(size=r) for _ in range(B)]
    rw = thetas[2].copy()
    rw[10:] += 0.1 * rng.normal(size=r)
    gens = [0, 1, -1]
    # reference values: one chain at a time on fresh objects
    want = []
    ref_ctx = pkg.IcpContext(model, target, device=0)
    for b in range(B):
        ev, props = objects(ref_ctx)
        want.append(pkg.chain_step(ev, props, thetas[b], gens[b], z=zs[b], theta_prop=rw if gens[b] < 0 else None))
        ev.close(); [p.close() for p in props]
    ref_ctx.close()
    out, val, fwd, bwd, status = pkg.chain_step_batched([o[0] for o in objs], [o[1] for o in objs], thetas, gens, z=zs,
                                                        theta_prop=[None, None, rw])
    assert list(status) == [0, 0, 0]
    for b in range(B):
        assert np.array_equal(out[b], want[b][0]) and val[b] == want[b][1]
        assert np.array_equal(fwd[b], want[b][2]) and np.array_equal(bwd[b], want[b][3])
    # the same step again, now with chain 1 given the context of chain 0: it is stepped behind the batch, same values
    ev1, props1 = objects(ctxs[0])
    out2, val2, fwd2, bwd2, status2 = pkg.chain_step_batched([objs[0][0], ev1, objs[2][0]], [objs[0][1], props1, objs[2][1]], thetas, gens,
                                                             z=zs, theta_prop=[None, None, rw])
    assert list(status2) == [0, 0, 0]
    assert np.array_equal(out2, out) and np.array_equal(val2, val) and np.array_equal(fwd2, fwd) and np.array_equal(bwd2, bwd)
    ev1.close(); [p.close() for p in props1]
    for ev, props in objs:
        ev.close(); [p.close() for p in props]
    [c.close() for c in ctxs]


def test_batched_chains_rank_101(pkg):
    """Ranks above 64 decompose through the library on each chain's own stream; the batch waits for them on the host."""
    model, target = pkg.data.load_femur_model_and_target(100)
    n_steps, B = 12, 3
    setups = [pkg.femur_icp_proposal_registration(model, target, fused=2) for _ in range(B)]
    inits = [pkg.random_initial_parameters(model, chain_index=i) for i in range(B)]

    def make():
        ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
        return ctxs, [pkg.SamplingRegistration(ctxs[i], setups[i], inits[i], seed=7 + i) for i in range(B)]

    ctxs, chains = make()
    single = [c.run(n_steps) for c in chains]
    [c.close() for c in chains]; [c.close() for c in ctxs]
    ctxs, chains = make()
    got = pkg.run_chains_batched(chains, n_steps)
    for b in range(B):
        assert np.array_equal(got[b], single[b]), f"chain {b} differs"
    assert sum(s[:, 1].sum() for s in single) > 3
    [c.close() for c in chains]; [c.close() for c in ctxs]


def test_batched_chains_with_pose_proposals_and_one_icp_direction(pkg, femur50):
    """A mixture with pose proposals (those steps leave the submission and step on their own), ONE ICP direction and the
    collective evaluator, model-to-target, on the closed target: the merged launches cover it with one posterior per chain."""
    model, target = femur50
    n_steps, B = 40, 5
    def setup():
        s = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
        s.eval["mode"] = 0   # ModelToTargetEvaluation: the configuration the merged launches cover
        return s
    inits = [pkg.random_initial_parameters(model, chain_index=i) for i in range(B)]

    def make():
        ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
        return ctxs, [pkg.SamplingRegistration(ctxs[i], setup(), inits[i], seed=77 + i) for i in range(B)]

    ctxs, chains = make()
    single = [c.run(n_steps) for c in chains]
    [c.close() for c in chains]; [c.close() for c in ctxs]
    ctxs, chains = make()
    got = pkg.run_chains_batched(chains, n_steps)
    leaves = set()
    for b in range(B):
        assert np.array_equal(got[b], single[b]), f"chain {b} differs"
        leaves |= set(single[b][:, 2].astype(int))
    assert leaves & {3, 4, 5, 6, 7, 8} and 0 in leaves and 2 in leaves   # pose walks, the ICP proposal, the shape walk
    [c.close() for c in chains]; [c.close() for c in ctxs]


def _chain_objects(pkg, ctx, r, tp):
    props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.TargetSampling, True, decimatedTargetPoints=tp),
             pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True, decimatedTargetPoints=tp)]
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r, decimatedTargetPoints=tp)
    return ev, props


def test_batched_ticket_marks_contexts_busy_and_can_be_abandoned(pkg, femur50):
    """Between icp_chain_step_batched_issue and _collect / _abandon no mutex is held: the member contexts answer every other
    entry point with ICP_ERR_BUSY; an abandoned ticket leaves them usable and records nothing (the same step, issued again
    and collected, gives the values of a fresh context)."""
    model, target = femur50
    r = model.rank
    rng = np.random.default_rng(11)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    B = 2
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
    objs = [_chain_objects(pkg, c, r, tp) for c in ctxs]
    thetas = []
    for b in range(B):
        t = pkg.initial_parameters(model)
        t[10:] = 0.3 * rng.normal(size=r)
        thetas.append(t)
    zs = [rng.normal(size=r) for _ in range(B)]
    gens = [0, 1]
    ref_ctx = pkg.IcpContext(model, target, device=0)
    want = []
    for b in range(B):
        ev, props = _chain_objects(pkg, ref_ctx, r, tp)
        want.append(pkg.chain_step(ev, props, thetas[b], gens[b], z=zs[b]))
        ev.close(); [p.close() for p in props]
    ref_ctx.close()
    tk = pkg.BatchedStepTicket([o[0] for o in objs], [o[1] for o in objs], thetas, gens, z=zs)
    for c in ctxs:
        with pytest.raises(pkg._native.IcpNativeError) as ei:
            c.transformedMesh(thetas[0])
        assert ei.value.status == -6
    with pytest.raises(pkg._native.IcpNativeError) as ei:  # a second batch over a busy context is refused, too
        pkg.BatchedStepTicket([objs[0][0]], [objs[0][1]], [thetas[0]], [0], z=[zs[0]])
    assert ei.value.status == -6
    tk.abandon()
    assert ctxs[0].transformedMesh(thetas[0]).shape == (model.n_points, 3)
    out, val, fwd, bwd, status = pkg.BatchedStepTicket([o[0] for o in objs], [o[1] for o in objs], thetas, gens, z=zs).collect()
    assert list(status) == [0, 0]
    for b in range(B):
        assert np.abs(out[b] - want[b][0]).max() <= 1e-9 and abs(val[b] - want[b][1]) <= 1e-9 * abs(want[b][1])
        assert np.allclose(fwd[b], want[b][2], rtol=1e-9, atol=0) and np.allclose(bwd[b], want[b][3], rtol=1e-9, atol=0)
    for ev, props in objs:
        ev.close(); [p.close() for p in props]
    for c in ctxs:
        c.close()


def test_dropped_half_step_is_ordered_before_its_replacement(pkg):
    """A half step launched ahead for an outcome that does not happen (icp_chain_step_prelaunch with a wrong guess) writes the
    same scratch, hints, state slot and memo entries as the step that replaces it.  The replacement must run BEHIND it — also
    when it waits for no decomposition (a random-walk proposal, generator < 0) and the dropped launches are long (58k-vertex
    target): every value must equal what a context that never saw the dropped half step computes."""
    model, target = pkg.data.synthetic_femur_target()  # the metric workload: 58,322 vertices
    r = model.rank
    rng = np.random.default_rng(23)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    ctx, ref_ctx = pkg.IcpContext(model, target, device=0), pkg.IcpContext(model, target, device=0)
    ev, props = _chain_objects(pkg, ctx, r, tp)
    rev, rprops = _chain_objects(pkg, ref_ctx, r, tp)
    cur = pkg.initial_parameters(model)
    cur[10:] = 0.2 * rng.normal(size=r)
    for it in range(12):
        z = rng.normal(size=r)
        gen = it % 2
        got = pkg.chain_step(ev, props, cur, gen, z=z)
        want = pkg.chain_step(rev, rprops, cur, gen, z=z)
        assert np.array_equal(got[0], want[0]) and got[1] == want[1]
        # the caller "expects a rejection" and launches the next ICP half step ahead …
        pkg.chain_step_prelaunch(ev, props, cur, 1 - gen, z=rng.normal(size=r))
        # … but the step is accepted and the next proposal is a random walk from the NEW state: the half step is dropped
        cur = got[0].copy()
        rw = cur.copy()
        rw[10:] += 0.05 * rng.normal(size=r)
        got = pkg.chain_step(ev, props, cur, -1, theta_prop=rw)
        want = pkg.chain_step(rev, rprops, cur, -1, theta_prop=rw)
        assert got[1] == want[1], (it, got[1], want[1])
        assert np.array_equal(got[2], want[2]) and np.array_equal(got[3], want[3])
        pm, pw = props[1].icpPosterior(rw, with_aux=False), rprops[1].icpPosterior(rw, with_aux=False)
        assert np.array_equal(pm.corr_id, pw.corr_id) and np.array_equal(pm.corr_point, pw.corr_point)
    for e_, ps in ((ev, props), (rev, rprops)):
        e_.close(); [p.close() for p in ps]
    ctx.close(); ref_ctx.close()


def test_femur100_all_points_symmetric_58k_target_matches_oracle(pkg, oracle):
    """configs[2] on the target SURVEY.md §8d assigns to it (58,322 vertices / 116,640 triangles): femur-100, every model point a
    sample point of the proposal and of the symmetric evaluator, random-init chains.  Two chains of a few steps against the
    oracle's chain (tree back end: bit-identical to its scans): identical decisions, states within 1e-5, and the correspondence
    indices of the final states bit for bit."""
    model, target = pkg.data.synthetic_femur_target(n_components=100)
    assert target.n_points == 58322 and model.rank == 101
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    setup = pkg.femur_random_init_comparison(model, target)
    cfg = oracle_chain_config(oracle, setup)
    ctx = pkg.IcpContext(model, target, device=0)
    n_steps = 5
    try:
        oracle.set_search_backend(oracle.SEARCH_TREES)
        for trial in (1, 2):
            theta0 = pkg.random_initial_parameters(model, chain_index=trial)
            seed = 1024 + trial
            acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, cfg, theta0, seed, n_steps)
            chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
            rec = chain.run(n_steps)
            assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
            assert np.array_equal(rec[:, 2].astype(np.int32), comp_o)
            scale = np.abs(states_o[:, 10:]).max()
            assert np.abs(rec[:, 14:] - states_o[:, 10:]).max() <= 1e-5 * scale
            assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
            chain.close()
            # correspondences of all 1622 model points against the 116,640 triangles, at the chain's last state
            theta = rec[-1, 4:].copy()
            pp = oracle.proposal_params(0.1, 10.0, 5.0, oracle.MODEL_SAMPLING, True, n_model_ids=model.n_points)
            prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, model.n_points, "ModelSampling", True)
            post, po = prop.icpPosterior(theta, with_aux=False), oracle.icp_posterior(om, ot, pp, theta)
            assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.corr_point, po.corr_pt) and np.array_equal(post.keep, po.keep)
            prop.close()
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
    ctx.close()


def test_contexts_share_model_and_target_memory(pkg):
    """one context per chain must not duplicate the immutable model / target data nor hold worst-case candidate
    lists: at the metric size (58,322-vertex target) a further context with its chain costs < 32 MB of HBM (measured: 28 MB, of which
    about 10 MB are the runtime's own per-stream allocations — 64 chains: 1.8 GB, against 7 GB with per-context copies and worst-case
    lists), and chains on such contexts still give the values of a chain on a context of its own."""
    import ctypes
    # the HIP runtime the library itself is linked to (the very file this process has mapped: another copy would be another runtime)
    with open("/proc/self/maps") as f:
        paths = {line.split()[-1] for line in f if "libamdhip64" in line}
    assert len(paths) >= 1
    hip = ctypes.CDLL(sorted(paths)[0])

    def free_bytes():
        free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
        assert hip.hipDeviceSynchronize() == 0 and hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        return free.value

    model, target = pkg.data.synthetic_femur_target()
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    first = pkg.IcpContext(model, target, device=0)
    ch0 = pkg.SamplingRegistration(first, setup, pkg.initial_parameters(model), seed=1024)
    want = ch0.run(30)
    free0 = free_bytes()
    n = 16
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(n)]
    chains = [pkg.SamplingRegistration(c, setup, pkg.initial_parameters(model), seed=1024) for c in ctxs]
    recs = pkg.run_chains_batched(chains, 30)
    free1 = free_bytes()
    per_ctx = (free0 - free1) / n
    assert per_ctx < 32e6, f"{per_ctx / 1e6:.1f} MB per further context"
    for rec in recs:
        assert np.array_equal(rec[:, 1:3], want[:, 1:3])
        assert np.abs(rec[:, 14:] - want[:, 14:]).max() <= 1e-9 * np.abs(want[:, 14:]).max()
    for ch in chains + [ch0]:
        ch.close()
    for c in ctxs + [first]:
        c.close()


def test_no_fallback_in_normal_runs(pkg, femur50):
    """The step schedules take their cross-stream order on the device, with time-outs behind them (icp_ctx_runtime_stats).  A normal
    run must never need one: the metric chain over 3,000 steps (pipelined single-chain step, speculative decompositions) and 64 chains
    x 300 steps through the batched step, burn-in included — where round 2's profile showed a 50 ms time-out (more than 24
    decompositions per batch went out as several launches on one stream, and the chip-wide first launch of the step, spinning on
    THEIR completion words, kept the later ones from becoming resident) — leave every counter at zero."""
    model, _ = femur50
    _, target = pkg.data.synthetic_femur_target()
    before = pkg._native.runtime_stats()
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
    rec = chain.run(3000)
    assert 500 < rec[:, 1].sum() < 2000
    assert all(v == 0 for v in ctx.runtime_stats().values()), ctx.runtime_stats()
    chain.close(); ctx.close()
    B = 64
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
    chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, chain_index=i), seed=2000 + i) for i in range(B)]
    recs = pkg.run_chains_batched(chains, 300)
    assert sum(r[:, 1].sum() for r in recs) > 64 * 60          # burn-in: most early steps are accepted
    per_ctx = [c.runtime_stats() for c in ctxs]
    assert all(v == 0 for st in per_ctx for v in st.values()), [st for st in per_ctx if any(st.values())]
    [c.close() for c in chains]; [c.close() for c in ctxs]
    after = pkg._native.runtime_stats()
    assert after == before, (before, after)


_DEVICE_LOOP_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.load_femur_model_and_target(50)
kind = {kind!r}
def make_setup():
    if kind == "metric":      # apps/femur/IcpProposalRegistration.scala:59-85: two ICP directions + shape walk, independent evaluator
        s = pkg.femur_icp_proposal_registration(model, target, fused=2)
    elif kind == "root":      # the same with the opt-in Cholesky-root sampler
        s = pkg.femur_icp_proposal_registration(model, target, fused=2); s.sampler = "cholesky-root"
    else:                     # ONE ICP direction + shape walk, collective evaluator (model-to-target) on the closed target
        s = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
        s.eval["mode"] = 0; s.w_pose = 0.0; s.w_icp, s.w_rw = 0.7, 0.3; s.rw_sigma = 0.02
    return s
B = {B}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], make_setup(), pkg.random_initial_parameters(model, i), seed=300 + i) for i in range(B)]
a = pkg.run_chains_batched(chains, {n1})
single = chains[0].run(7)                       # a chain goes on by itself (host-stepped) from where the device loop left it …
b = pkg.run_chains_batched(chains[1:], {n2})    # … and the others through a second run
states = [c.state() for c in chains]
np.savez({out!r}, a=np.stack(a), single=single, b=np.stack(b), theta=np.stack([s[0] for s in states]), logp=np.array([s[1] for s in states]),
         n=np.array([s[2] for s in states]), acc=np.array([s[3] for s in states]), stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


@pytest.mark.parametrize("kind,B,n1,n2", [("metric", 9, 150, 40), ("metric", 50, 140, 30), ("root", 9, 60, 20), ("one-direction", 5, 60, 20)])
def test_device_loop_matches_host_stepped_chains(kind, B, n1, n2, tmp_path):
    """the whole Metropolis–Hastings loop on the device (icp_chains_run_on_device — mixture draw, the proposals'
    inputs, MetropolisHastings.next and the records by kernels of the step's own stream) against the same chains stepped by the host
    harness through icp_chain_step_batched: IDENTICAL records — every decision, every mixture component, states and log values bit
    for bit — over 140-150 steps (more than 128 decompositions per proposal: the cold restart is on the same step), with chains that
    go on afterwards on either path, for two ICP directions + shape walk, one direction + shape walk with the collective evaluator, and
    the Cholesky-root sampler."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = {}
    for mode in ("1", "0"):
        path = str(tmp_path / f"dl_{mode}.npz")
        script = _DEVICE_LOOP_SCRIPT.format(root=ROOT, kind=kind, B=B, n1=n1, n2=n2, out=path)
        subprocess.run([sys.executable, "-c", script], check=True, env={**os.environ, "ICP_HOST_DEVICE_LOOP": mode}, timeout=600)
        out[mode] = np.load(path)
    dev, host = out["1"], out["0"]
    for key in ("a", "single", "b", "theta", "logp", "n", "acc"):
        assert np.array_equal(dev[key], host[key]), key
    assert np.all(dev["stats"] == 0) and np.all(host["stats"] == 0)
    a = dev["a"]
    assert a.shape == (B, n1, 14 + 51) and np.array_equal(a[:, :, 0], np.tile(np.arange(n1), (B, 1)))
    assert 0.02 < a[:, :, 1].mean() < 0.95
    leaves = set(a[:, :, 2].astype(int).ravel())
    assert leaves == ({0, 1, 2} if kind != "one-direction" else {0, 2})
    assert np.all(dev["n"][1:] == n1 + n2) and dev["n"][0] == n1 + 7


_DEVICE_LOOP_ORACLE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.load_femur_model_and_target(50)
kind = {kind!r}
def make_setup():
    if kind == "femur":   # apps/femur/IcpProposalRegistration.scala:59-85: two ICP directions + shape walk, independent evaluator
        return pkg.femur_icp_proposal_registration(model, target, fused=2)
    # apps/bfm/BfmFittingPartial.scala:62-83 on the (closed) femur target: pose walks + one ICP direction + shape walk, collective evaluator
    s = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
    s.pose_rot_sigma = (0.01, 0.012, 0.008); s.pose_trans_sigma = (0.1, 0.15, 0.08); s.rw_sigma = 0.02
    return s
B = {B}
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
chains = [pkg.SamplingRegistration(ctxs[i], make_setup(), pkg.random_initial_parameters(model, i), seed=500 + i) for i in range(B)]
rec = pkg.run_chains_batched(chains, {n})
paths = [c.step_paths() for c in ctxs]
np.savez({out!r}, rec=np.stack(rec), loop=np.array([p["device_loop"] for p in paths]), other=np.array([p["merged"] + p["wide"] + p["per_stage"] for p in paths]),
         stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


@pytest.mark.parametrize("kind,B,n", [("femur", 4, 70), ("pose", 4, 90)])
def test_device_loop_matches_oracle_chain(pkg, femur50, femur50_oracle, oracle, kind, B, n, tmp_path):
    """The on-device Metropolis–Hastings loop (icp_chains_run_on_device) DIRECTLY against the oracle's chain (orc_run_chain: the same
    counter-based random numbers bit for bit) — api/sampling/SamplingRegistration.scala:52-85 semantics: every decision and mixture
    component identical, states within 1e-5, log values within 1e-6 — for the femur mixture (two ICP directions + shape walk,
    apps/femur/IcpProposalRegistration.scala:70-72) and for the mixture WITH the six pose walks (api/sampling/proposals/PoseProposals.scala:31-90
    in the order of apps/bfm/BfmFittingPartial.scala:70), whose proposed pose is made on the device with include/icp_sincos.h — the
    sines and cosines the oracle uses."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    model, target = femur50
    om, ot = femur50_oracle
    path = str(tmp_path / "dlo.npz")
    script = _DEVICE_LOOP_ORACLE_SCRIPT.format(root=ROOT, kind=kind, B=B, n=n, out=path)
    subprocess.run([sys.executable, "-c", script], check=True, env={**os.environ, "ICP_HOST_DEVICE_LOOP": "1"}, timeout=900)
    got = np.load(path)
    assert np.all(got["loop"] == n) and np.all(got["other"] == 0), "the chains did not run inside the on-device loop"
    assert np.all(got["stats"] == 0)
    if kind == "femur":
        setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    else:
        setup = pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=2)
        setup.pose_rot_sigma = (0.01, 0.012, 0.008); setup.pose_trans_sigma = (0.1, 0.15, 0.08); setup.rw_sigma = 0.02
    cfg = oracle_chain_config(oracle, setup)
    leaves = set()
    from conftest import oracle_chains_parallel
    want = oracle_chains_parallel(oracle, [(om, ot, cfg, pkg.random_initial_parameters(model, b), 500 + b, n) for b in range(B)], trees=False)
    for b in range(B):
        acc_o, comp_o, logp_o, states_o = want[b]
        rec = got["rec"][b]
        assert np.array_equal(rec[:, 0], np.arange(n))
        assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), f"chain {b}: accept/reject sequences differ"
        assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), f"chain {b}: mixture components differ"
        scale = np.abs(states_o[:, 10:]).max()
        assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
        assert np.abs(rec[:, 4:14] - states_o[:, :10]).max() <= 1e-12 * max(1.0, np.abs(states_o[:, :10]).max())
        assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
        assert acc_o.sum() > 3
        leaves |= set(comp_o.tolist())
    if kind == "pose":
        assert leaves & {3, 4, 5} and leaves & {6, 7, 8} and 0 in leaves, leaves  # rotations, translations and ICP proposals all occurred
        moved = [b for b in range(B) if np.abs(got["rec"][b][-1, 4:14] - pkg.random_initial_parameters(model, b)[:10]).max() > 0]
        assert moved, "no pose walk was accepted in any chain"


def test_femur100_chain_from_the_deterministic_fit_accepts_and_matches_oracle(pkg, oracle):
    """BASELINE.json configs[2] with ACCEPTED steps.  From a random start this chain (K = N = 1622 correspondences, rank 101) rejects
    every ICP proposal under the reference's own transition ratio: the backward density evaluates (2 − step)·(c − α) against a posterior
    of precision M, log T_fwd − log T_bwd ≈ ½(2 − step)²·(c − α)ᵀM(c − α) − ½‖z‖² — 1,026 at a random start against a likelihood gain of
    349.  The reference's experiments start such chains next to α: the deterministic ICP fit
    (apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:148, IcpRegistration.fitting with all model points).  From there (c − α)ᵀM(c − α)
    is ≈ 5 and about one proposal in ten is accepted: >= 20 accepted rank-101 steps, decision for decision against the oracle."""
    model, target = pkg.data.load_femur_model_and_target(100)
    r, n = model.rank, model.n_points
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    oracle.set_search_backend(oracle.SEARCH_TREES)  # (bit-identical to the scans: tests/test_oracle.py)
    try:
        setup = pkg.femur_random_init_comparison(model, target)
        ctx = pkg.IcpContext(model, target, device=0)
        start = pkg.random_initial_parameters(model, chain_index=3)
        fit = pkg.IcpBasedSurfaceFitting(ctx, 1.0, "ModelSampling", modelPointIds=np.arange(n, dtype=np.int32)).runfitting(10, initialModelParameters=start)
        fit_o = oracle.fit_deterministic(om, ot, start, 10, direction=oracle.MODEL_SAMPLING, model_ids=np.arange(n, dtype=np.int32))
        assert np.abs(fit - fit_o).max() <= 1e-7 * np.abs(fit_o[10:]).max()
        n_steps, seed = 260, 1024
        acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), fit, seed, n_steps)
        assert acc_o.sum() >= 20, acc_o.sum()
        chain = pkg.SamplingRegistration(ctx, setup, fit, seed)
        rec = chain.run(n_steps)
        assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
        assert np.array_equal(rec[:, 2].astype(np.int32), comp_o)
        scale = np.abs(states_o[:, 10:]).max()
        assert np.abs(rec[:, 14:] - states_o[:, 10:]).max() <= 1e-5 * scale
        assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
        assert ctx.step_paths()["merged"] == n_steps  # (the merged symmetric step of rank 101)
        chain.close()
        ctx.close()
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)


_TWO_THREADS_SCRIPT = r"""
import sys, threading, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model, target = pkg.data.synthetic_femur_target()
B = 32
ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
chains = [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for i in range(B)]
recs = [None] * B
def work(k):
    mine = list(range(k, B, 2))
    parts = [pkg.run_chains_batched([chains[i] for i in mine], n) for n in (3, 17)]   # (two calls: a first submission and a continuation)
    for j, i in enumerate(mine):
        recs[i] = np.concatenate([p[j] for p in parts])
ths = [threading.Thread(target=work, args=(k,)) for k in range(2)]
[t.start() for t in ths]; [t.join() for t in ths]
np.savez({out!r}, rec=np.stack(recs), stats=np.array(list(pkg._native.runtime_stats().values())))
[c.close() for c in chains]; [c.close() for c in ctxs]
"""


def test_two_host_threads_with_a_batch_each_in_a_fresh_process(pkg, tmp_path):
    """Two host threads, each stepping its own batch of 16 metric-configuration chains on one GPU, from the first call of a fresh process
    on — the device is busy with the other thread's launches while a launch context makes its buffers.  (Round 4: a batch gate's arrival
    counter was zeroed by a hipMemset that landed AFTER the first arrivals — hipMemset does not wait for the device, and the null stream
    is not ordered against non-blocking streams — and every later gate of that ring slot opened two seconds late.)  No fall-back
    counter may move, and the chains — independent of each other — must give the records of the same chains stepped by one thread."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    out = str(tmp_path / "two.npz")
    subprocess.run([sys.executable, "-c", _TWO_THREADS_SCRIPT.format(root=ROOT, out=out)], check=True, env=dict(os.environ), timeout=600)
    got = np.load(out)
    assert np.all(got["stats"] == 0), got["stats"]
    model, target = pkg.data.synthetic_femur_target()
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    idx = [0, 1, 14, 31]
    ctxs = [pkg.IcpContext(model, target, device=0) for _ in idx]
    chains = [pkg.SamplingRegistration(cx, setup, pkg.random_initial_parameters(model, i), seed=1024 + i) for cx, i in zip(ctxs, idx)]
    want = [np.concatenate([pkg.run_chains_batched([ch], n)[0] for n in (3, 17)]) for ch in chains]
    for w, i in zip(want, idx):
        assert np.array_equal(got["rec"][i], w), i
    for ch in chains: ch.close()
    for cx in ctxs: cx.close()
