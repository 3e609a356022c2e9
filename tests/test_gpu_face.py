"""GPU: BASELINE.json configs[3] / configs[4] — the Basel-Face-Model-sized configurations on the synthetic stand-in
(icp-proposal_amd/data.py: synthetic_face_model / synthetic_partial_target; the BFM itself is not redistributable).

Parity against the oracle at a reduced size the CPU restatement finishes in seconds, on a target WITH boundaries (every
boundary-aware branch of the proposal and the collective evaluator is live); at the full size (N = 28,561, rank 200,
K = 400, K_e = 800) the brute-force correspondence indices are still compared bit for bit, the rest through
size-independent properties, and the configuration's chain (pose + ICP + random walk) is run through the host harness."""
import numpy as np
import pytest

from conftest import make_theta

pytestmark = pytest.mark.gpu


def face_theta(model, seed, pose=True):
    t = make_theta(model, seed, shape_scale=0.4, pose=pose)
    return t


@pytest.fixture(scope="module")
def small(pkg, oracle):
    model = pkg.data.synthetic_face_model(grid=41, rank=40)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    yield model, target, ctx, oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    ctx.close()


def test_small_face_proposal_matches_oracle(pkg, oracle, small):
    model, target, ctx, om, ot = small
    r = model.rank
    assert pkg.data.boundary_vertex_flags(target).sum() > 0
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
    rng = np.random.default_rng(11)
    for seed in (1, 2):
        theta = face_theta(model, seed)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.corr_aux, po.corr_aux)
        assert np.array_equal(post.keep, po.keep) and np.array_equal(post.corr_point, po.corr_pt)
        assert np.abs(post.alpha - po.alpha).max() <= 1e-9 * np.abs(po.alpha).max()
        assert np.abs(post.S - po.S).max() <= 1e-9 * np.abs(po.S).max()
        z = rng.normal(size=r)
        got, want = prop.propose(theta, z), oracle.propose(om, ot, pp, theta, z)
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want[10:]).max()
        lf, lo = prop.logTransitionProbability(theta, got), oracle.log_transition(om, ot, pp, theta, want)
        assert abs(lf - lo) <= 1e-7 * abs(lo)
    prop.close()


@pytest.mark.parametrize("kind", ["collective", "hausdorff"])
def test_small_face_evaluators_match_oracle(pkg, oracle, small, kind):
    model, target, ctx, om, ot = small
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 4 * r)
    if kind == "collective":
        ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, 2, 4 * r, decimatedTargetPoints=tp)
        ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, 2, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
    else:
        ev = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
        ep = oracle.evaluator_params(oracle.EVAL_HAUSDORFF, 2, p0=1.0)
    for seed in (3, 4):
        theta = face_theta(model, seed)
        want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
        got, aux = ev.logValue(theta, return_aux=True)
        assert rc == 0 and abs(got - want) <= 1e-10 * abs(want)
    ev.close()


@pytest.mark.parametrize("rank", [100, 150, 193])
def test_tridiagonal_route_configurations_match_oracle(pkg, oracle, rank):
    """Ranks 65..200 are decomposed by Householder tridiagonalisation + multisection + twisted factorisation + one refinement step
    (icp_tridiag.hpp), in one of three register layouts: up to 128 (four waves), up to 192 (eight waves, three row slots), up to 200
    (four row slots, the first partly filled).  One posterior per layout against the oracle's Jacobi iteration (the full-size face
    tests and the femur-100 tests cover ranks 200 and 101 in the chain)."""
    model = pkg.data.synthetic_face_model(grid=41, rank=rank)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * rank)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * rank, "ModelSampling", True)
    rng = np.random.default_rng(rank)
    for seed in (5, 6):
        theta = face_theta(model, seed)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.keep, po.keep)
        assert np.abs(post.alpha - po.alpha).max() <= 1e-9 * np.abs(po.alpha).max()
        assert np.abs(post.S - po.S).max() <= 1e-11 * np.abs(po.S).max()
        assert np.abs(post.V.T @ post.V - np.eye(rank)).max() <= 1e-11
        assert np.abs(post.V - po.V).max() <= 1e-8
        z = rng.normal(size=rank)
        got, want = prop.propose(theta, z), oracle.propose(om, ot, pp, theta, z)
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want[10:]).max()
    prop.close()
    ctx.close()


@pytest.mark.parametrize("rank", [120, 128, 230])
def test_register_tiled_factorisation_ranks(pkg, oracle, rank):
    """Ranks 117..250 are factored entirely in registers (k_posterior_factor_tiles: three 2 x 4 tiles per thread up to rank 221, four
    above; finished columns to global scratch, staged back substitution).  The posterior mean and the transition density — both
    straight out of that factorisation — against the oracle at the first rank past the LDS form, at a block boundary of the back
    substitution and in the four-tile configuration (ranks 150, 193 and 200 are covered by the tests above and the full-face chains)."""
    model = pkg.data.synthetic_face_model(grid=41, rank=rank)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * rank)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * rank, "ModelSampling", True)
    for seed in (5, 6):
        theta, theta2 = face_theta(model, seed), face_theta(model, seed)
        theta2[10:] += 0.01 * np.random.default_rng(seed).normal(size=rank)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.keep, po.keep)
        assert np.abs(post.alpha - po.alpha).max() <= 1e-9 * np.abs(po.alpha).max()
        got, want = prop.logTransitionProbability(theta, theta2), oracle.log_transition(om, ot, pp, theta, theta2)
        assert np.isfinite(want) and abs(got - want) <= 1e-6 * abs(want)
    prop.close()
    ctx.close()


def test_failed_factorisation_is_reported_through_the_relayed_status(pkg):
    """A state whose coefficients overflow the instance makes the posterior's normal equations non-finite; the factorisation says so in
    the posterior's status words, which the proposal kernel and the transition tails pass on to the call's result block (no status copy
    of their own): icp_proposal_propose and icp_chain_eval_step must both come back with an error, and the objects stay usable."""
    model = pkg.data.synthetic_face_model(grid=41, rank=150)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 300, "ModelSampling", True)
    ev = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
    good = face_theta(model, 3)
    bad = good.copy()
    bad[10:] = 1e200
    z = np.random.default_rng(0).normal(size=model.rank)
    with pytest.raises(pkg._native.IcpNativeError):
        prop.propose(bad, z)
    with pytest.raises(pkg._native.IcpNativeError):
        pkg.api.chain_eval_step(ev, [prop], good, bad)
    out = prop.propose(good, z)                       # … and the next good call is served
    val, fwd, bwd = pkg.api.chain_eval_step(ev, [prop], good, out)
    assert np.all(np.isfinite(out)) and np.isfinite(val) and np.isfinite(fwd[0]) and np.isfinite(bwd[0])
    prop.close(); ev.close(); ctx.close()


def test_model_cache_serves_contexts_made_one_after_the_other(pkg, oracle):
    """The derived data of a model stays alive between contexts built one after the other (configs[4]: one context per target), and
    icp_release_cached_models drops it: results before, between and after are the same."""
    model = pkg.data.synthetic_face_model(grid=41, rank=40)
    theta = face_theta(model, 9)
    vals = []
    for rnd in range(3):
        target = pkg.data.synthetic_partial_target(model, seed=7, n_remove=90)
        ctx = pkg.IcpContext(model, target, device=0)
        hd = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
        vals.append((hd.logValue(theta), ctx.transformedMesh(theta).copy()))
        hd.close(); ctx.close()
        if rnd == 1:
            pkg._native.lib().icp_release_cached_models()
    assert vals[0][0] == vals[1][0] == vals[2][0]
    assert np.array_equal(vals[0][1], vals[1][1]) and np.array_equal(vals[0][1], vals[2][1])


def test_multiple_eigenvalues_take_the_jacobi_fall_back(pkg, oracle):
    """Ranks above 64 are decomposed by tridiagonalisation + multisection + twisted factorisation, which needs eigenvalues it can
    tell apart.  The stand-in's variances come in equal pairs (modes (p, q) and (q, p)): without correspondences — and with a
    single one, a rank-3 change — the posterior's spectrum has exact multiples, the route reports them on the device and the Jacobi
    iteration takes over in the same stream; with the usual 2r correspondences it does not.  All three against the oracle."""
    model = pkg.data.synthetic_face_model(grid=41, rank=72)
    assert (np.diff(np.sort(model.variance)) == 0).sum() >= 30
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    theta = face_theta(model, 3)
    for K in (0, 1, 144):
        pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=K)
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, K, "ModelSampling", True)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert np.abs(post.S - po.S).max() <= 1e-12 * np.abs(po.S).max()
        assert np.abs(post.V.T @ post.V - np.eye(72)).max() <= 1e-10
        assert np.abs(post.V - po.V).max() <= 1e-9
        prop.close()
    ctx.close()


@pytest.fixture(scope="module")
def full_face(pkg):
    model = pkg.data.synthetic_face_model()          # N = 28,561, T = 56,448, rank 200
    target = pkg.data.synthetic_partial_target(model)
    ctx = pkg.IcpContext(model, target, device=0)
    yield model, target, ctx
    ctx.close()


def test_full_face_indices_and_properties(pkg, oracle, full_face):
    model, target, ctx = full_face
    r = model.rank
    assert model.n_points == 28561 and r == 200 and target.n_points > 27000
    theta = face_theta(model, 21)
    # brute-force correspondence search at full size: indices, points and boundary filter bit-identical to the oracle
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
    post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
    assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.corr_aux, po.corr_aux)
    assert np.array_equal(post.keep, po.keep) and np.array_equal(post.corr_point, po.corr_pt)
    assert np.abs(post.alpha - po.alpha).max() <= 1e-8 * np.abs(po.alpha).max()
    # size-independent properties of the posterior and the proposal
    assert np.all(np.linalg.eigvalsh(0.5 * (post.M + post.M.T)) >= 1.0 - 1e-9)
    assert np.all(post.S > 0) and np.all(np.diff(post.S) <= 1e-12) and post.S[0] <= model.variance.max() * (1 + 1e-9)
    assert np.allclose(post.V.T @ post.V, np.eye(r), atol=1e-9)
    got0 = prop.propose(theta, np.zeros(r))
    assert np.abs(got0[10:] - (theta[10:] + 0.1 * (post.alpha - theta[10:]))).max() < 1e-6 * np.abs(post.alpha).max()
    z = np.random.default_rng(2).normal(size=r)
    a, b = prop.propose(theta, z), prop.propose(theta, -z)
    assert np.allclose(0.5 * (a + b), got0, rtol=1e-8, atol=1e-11)
    assert np.isfinite(prop.logTransitionProbability(theta, a))
    other = a.copy(); other[4] += 0.01
    assert prop.logTransitionProbability(theta, other) == -np.inf
    prop.close()
    # collective evaluator at full size against the oracle (K_e = 800 per direction)
    tp = pkg.data.decimated_point_subset(target, 4 * r)
    ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, 2, 4 * r, decimatedTargetPoints=tp)
    ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, 2, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
    want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
    got, aux = ev.logValue(theta, return_aux=True)
    assert rc == 0 and abs(got - want) <= 1e-10 * abs(want)
    ev.close()
    # full-mesh Hausdorff evaluator (28,561 x 54,324 + 27,561 x 56,448 point-triangle pairs): bounds instead of the oracle
    hd = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
    val, haux = hd.logValue(theta, return_aux=True)
    assert np.isfinite(val) and haux[0] == max(haux[1], haux[2]) and abs(val - (-haux[0])) < 1e-12 * max(1.0, haux[0])
    assert haux[1] >= aux[1] - 1e-9 or haux[2] >= aux[1] - 1e-9   # the max over ALL points bounds the max over a subset
    hd.close()


@pytest.mark.parametrize("evaluator", ["collective", "hausdorff"])
def test_full_face_chain_runs(pkg, full_face, evaluator):
    """apps/bfm/BfmFittingPartial.scala:62-83 through the host harness: pose + ICP + random-walk mixture."""
    model, target, ctx = full_face
    setup = pkg.bfm_fitting_partial(model, target, evaluator=evaluator)
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=5)
    rec = chain.run(12)
    assert np.all(np.isfinite(rec)) and np.array_equal(rec[:, 0], np.arange(12))
    leaves = set(rec[:, 2].astype(int))
    assert leaves & {3, 4, 5, 6, 7, 8}, "no pose proposal drawn"   # leaf ids 3..8 = the six pose random walks
    chain.close()


def compare_chain_with_oracle(rec, acc_o, comp_o, logp_o, states_o):
    """Identical accept/reject and mixture-component sequences, states within 1e-5 relative."""
    n = rec.shape[0]
    assert np.array_equal(rec[:, 0], np.arange(n))
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), "mixture components differ"
    scale = max(np.abs(states_o[:, 10:]).max(), 1e-3)
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 4 + 1:4 + 7] - states_o[:, 1:7]).max() <= 1e-12       # pose parameters: host arithmetic on both sides
    assert np.array_equal(rec[:, 4 + 7:4 + 10], states_o[:, 7:10]) and np.array_equal(rec[:, 4], states_o[:, 0])
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()


@pytest.mark.parametrize("evaluator", ["collective", "hausdorff"])
def test_small_face_pose_chain_matches_oracle(pkg, oracle, small, evaluator):
    """apps/bfm/BfmFittingPartial.scala:62-83 at reduced size: 0.4 pose + 0.55 ICP(ModelSampling) + 0.05 shape walk, on a target
    WITH boundary, 150 steps, decision for decision against the oracle's chain (which has the six pose walks of
    api/sampling/proposals/PoseProposals.scala:31-90 since round 3) — with UNEQUAL pose sigmas, so that a wrong pairing of name,
    sigma and parameter (round 2: Yaw and Roll swapped) changes the chain."""
    from test_gpu_chain import oracle_chain_config
    model, target, ctx, om, ot = small
    setup = pkg.bfm_fitting_partial(model, target, evaluator=evaluator)
    setup.pose_rot_sigma, setup.pose_trans_sigma = (0.02, 0.01, 0.004), (0.2, 0.1, 0.05)
    theta0, seed, n = pkg.initial_parameters(model), 77, 150
    try:
        oracle.set_search_backend(oracle.SEARCH_TREES)
        acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n)
    compare_chain_with_oracle(rec, acc_o, comp_o, logp_o, states_o)
    assert 20 < acc_o.sum() < n and (comp_o == 0).sum() > 40 and (comp_o == 2).sum() > 0
    assert all((comp_o == k).sum() > 0 for k in range(3, 9)), "not every pose walk was drawn"
    # accepted moves of all six pose walks: each changed exactly the parameter the reference's axis names
    moved = set()
    prev = theta0
    for s in range(n):
        if acc_o[s] and comp_o[s] >= 3:
            assert list(np.flatnonzero(rec[s, 4:] != prev)) == [pkg.sampling.POSE_LEAF_PARAMETER[int(comp_o[s])]]
            moved.add(int(comp_o[s]))
        prev = rec[s, 4:]
    assert len(moved) >= 4
    # the whole mixture's transition density of a pose move (the Yaw walk evaluated on a Roll move is FINITE and enters the
    # log-sum-exp: PoseProposals.scala:47-49) and of a shape move, harness against oracle
    cfg = oracle_chain_config(oracle, setup)
    k = next(s for s in range(1, n) if acc_o[s] and comp_o[s] == 5)             # an accepted Roll move
    a, b = rec[k - 1, 4:].copy(), rec[k, 4:].copy()
    got, want = chain.logTransitionProbability(a, b), oracle.chain_log_transition(om, ot, cfg, a, b)
    hand = np.log(0.4) + np.log(sum(np.exp(-0.5 * (d / sd) ** 2) / (np.sqrt(2 * np.pi) * sd)
                                    for d, sd in ((0.0, 0.02), (0.0, 0.01), (b[4] - a[4], 0.004))) / 6.0)
    assert abs(got - want) <= 1e-13 * abs(want) and abs(got - hand) <= 1e-12 * abs(hand)
    k = next(s for s in range(1, n) if acc_o[s] and comp_o[s] == 0)             # an accepted ICP move
    a, b = rec[k - 1, 4:].copy(), rec[k, 4:].copy()
    got, want = chain.logTransitionProbability(a, b), oracle.chain_log_transition(om, ot, cfg, a, b)
    assert np.isfinite(want) and abs(got - want) <= 1e-6 * abs(want)
    chain.close()


@pytest.fixture(scope="module")
def full_face_oracle_chains(pkg, oracle, full_face):
    """The oracle's 12-step chains of the two full-size tests below, computed side by side (a host thread each: conftest)."""
    from conftest import oracle_chains_parallel
    from test_gpu_chain import oracle_chain_config
    model, target, ctx = full_face
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    kinds = ("collective", "hausdorff")
    jobs = [(om, ot, oracle_chain_config(oracle, pkg.bfm_fitting_partial(model, target, evaluator=k)), pkg.initial_parameters(model), 5, 12)
            for k in kinds]
    return dict(zip(kinds, oracle_chains_parallel(oracle, jobs)))


@pytest.mark.parametrize("evaluator", ["collective", "hausdorff"])
def test_full_face_pose_chain_matches_oracle(pkg, oracle, full_face, full_face_oracle_chains, evaluator):
    """configs[3] at FULL size (N = 28,561, rank 200, K = 400, K_e = 800 / the full-mesh Hausdorff evaluator): the first 12 steps of
    the BfmFittingPartial chain, decision for decision against the oracle's chain (tree back end, bit-identical to its scans)."""
    model, target, ctx = full_face
    setup = pkg.bfm_fitting_partial(model, target, evaluator=evaluator)
    theta0, seed, n = pkg.initial_parameters(model), 5, 12
    acc_o, comp_o, logp_o, states_o = full_face_oracle_chains[evaluator]
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n)
    compare_chain_with_oracle(rec, acc_o, comp_o, logp_o, states_o)
    assert (comp_o == 0).sum() > 0 and (comp_o >= 3).sum() > 0 and acc_o.sum() > 0
    chain.close()


def test_batch_registration_world1(pkg):
    """configs[4] plumbing on one GPU: (target, chain) work items, one context per target, records returned in item order."""
    model = pkg.data.synthetic_face_model(grid=41, rank=40)
    targets = [pkg.data.synthetic_partial_target(model, seed=s, n_remove=90) for s in (7, 8)]
    items, recs = pkg.sharding.run_batch(pkg, model, targets, n_chains=2, n_steps=5, make_setup=pkg.bfm_fitting_partial)
    assert items == [(0, 0), (0, 1), (1, 0), (1, 1)] and len(recs) == 4
    for k, rec in enumerate(recs):
        assert rec.shape == (5, 14 + model.rank) and np.all(rec[:, 0] == k) and np.all(np.isfinite(rec[:, 1:]))
    assert not np.array_equal(recs[0][:, 14:], recs[1][:, 14:])   # different initial shapes / seeds
    # the same job with the rank's chains stepped in lockstep, three per submission: identical records
    items3, recs3 = pkg.sharding.run_batch(pkg, model, targets, n_chains=2, n_steps=5, make_setup=pkg.bfm_fitting_partial, chains_per_launch=3)
    assert items3 == items and all(np.array_equal(a, b) for a, b in zip(recs, recs3))


def test_batch_registration_lockstep_femur(pkg, femur50):
    """Closed target: the lockstep chains really share their launches (icp_chain_step_batched); records as one by one."""
    model, target = femur50
    setup = lambda m, t: pkg.femur_icp_proposal_registration(m, t, fused=2)
    items, recs = pkg.sharding.run_batch(pkg, model, [target], n_chains=5, n_steps=25, make_setup=setup)
    items4, recs4 = pkg.sharding.run_batch(pkg, model, [target], n_chains=5, n_steps=25, make_setup=setup, chains_per_launch=4)
    assert items4 == items and all(np.array_equal(a, b) for a, b in zip(recs, recs4))
    assert sum(r[:, 1].sum() for r in recs) > 10


def test_batch_registration_lockstep_mixed_targets(pkg, femur50):
    """Chains of one submission may face targets of different sizes (their launches differ in grid size: the batch takes the
    largest and the smaller chains leave their surplus workgroups idle)."""
    model, target = femur50
    _, finer = pkg.data.synthetic_femur_target(n_subdiv=2)
    assert finer.n_points > 3 * target.n_points
    setup = lambda m, t: pkg.femur_icp_proposal_registration(m, t, fused=2)
    items, recs = pkg.sharding.run_batch(pkg, model, [target, finer], n_chains=3, n_steps=20, make_setup=setup)
    items6, recs6 = pkg.sharding.run_batch(pkg, model, [target, finer], n_chains=3, n_steps=20, make_setup=setup, chains_per_launch=6)
    assert items6 == items and all(np.array_equal(a, b) for a, b in zip(recs, recs6))
    assert sum(r[:, 1].sum() for r in recs) > 10


def test_full_face_hausdorff_matches_oracle(pkg, oracle, full_face):
    """configs[3]: the full-mesh Hausdorff evaluator (28,561 x 54,324 + 27,561 x 56,448 point-triangle pairs) against the oracle at
    FULL size — affordable with the oracle's tree back end (B1 of BASELINE.md §3; bit-identical to its brute-force scans,
    tests/test_oracle.py::test_search_backends_agree)."""
    model, target, ctx = full_face
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    ep = oracle.evaluator_params(oracle.EVAL_HAUSDORFF, 2, p0=1.0)
    hd = pkg.HausdorffDistanceEvaluator(ctx, 1.0)
    try:
        oracle.set_search_backend(oracle.SEARCH_TREES)
        for seed in (21, 22):
            theta = face_theta(model, seed)
            want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
            got, aux = hd.logValue(theta, return_aux=True)
            assert rc == 0 and abs(got - want) <= 1e-12 * abs(want), (got, want)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
    hd.close()


def test_batch_registration_full_face_matches_oracle(pkg, oracle):
    """configs[4] at FULL BFM size through the batch runner: 3 targets x 3 random-init chains (work items dealt as over 8 ranks,
    here world 1), a few steps each.  For every chain's final state the correspondence indices / points / boundary filter of the
    proposal and the collective likelihood are compared with the oracle (tree back end), and so is the decision sequence of one
    chain's first steps whose proposals were ICP or random-walk moves."""
    model = pkg.data.synthetic_face_model()
    r = model.rank
    targets = [pkg.data.synthetic_partial_target(model, seed=s) for s in (7, 8, 9)]
    n_steps = 6
    items, recs = pkg.sharding.run_batch(pkg, model, targets, n_chains=3, n_steps=n_steps, make_setup=pkg.bfm_fitting_partial)
    assert items == [(t, c) for t in range(3) for c in range(3)] and len(recs) == 9
    om = oracle.OracleModel.from_model(model)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    try:
        oracle.set_search_backend(oracle.SEARCH_TREES)
        for t in range(3):
            ot = oracle.OracleMesh(targets[t].points, targets[t].cells)
            ctx = pkg.IcpContext(model, targets[t], device=0)
            prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
            tp = pkg.data.decimated_point_subset(targets[t], 4 * r)
            ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, 2, 4 * r, decimatedTargetPoints=tp)
            ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, 2, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
            for c in range(3):
                rec = recs[3 * t + c]
                assert rec.shape == (n_steps, 14 + r) and np.all(np.isfinite(rec[:, 1:]))
                theta = rec[-1, 4:].copy()
                post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
                assert np.array_equal(post.corr_id, po.corr_id) and np.array_equal(post.corr_aux, po.corr_aux)
                assert np.array_equal(post.keep, po.keep) and np.array_equal(post.corr_point, po.corr_pt)
                want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
                got = ev.logValue(theta)
                assert rc == 0 and abs(got - want) <= 1e-10 * abs(want)
            ev.close(); prop.close(); ctx.close()
        # the decision sequence of one work item — (target 0, chain 1): random initial shape, seed as run_batch deals it — against
        # the oracle's chain (pose walks included)
        from test_gpu_chain import oracle_chain_config
        ot = oracle.OracleMesh(targets[0].points, targets[0].cells)
        setup = pkg.bfm_fitting_partial(model, targets[0])
        acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup),
                                                            pkg.random_initial_parameters(model, 1, 1024), 1024 + 1, n_steps)
        rec = recs[1].copy()
        rec[:, 0] = np.arange(n_steps)        # (run_batch stores the item id in the index field)
        compare_chain_with_oracle(rec, acc_o, comp_o, logp_o, states_o)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)


_SCHEDULE_SCRIPT = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
import __graft_entry__ as graft
pkg = graft.load_package()
model = pkg.data.synthetic_face_model(grid=41, rank={rank})
target = pkg.data.synthetic_partial_target(model, n_remove=90)
setup = pkg.bfm_fitting_partial(model, target, evaluator={evaluator!r})
ctx = pkg.IcpContext(model, target, device=0)
chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), 77)
np.save({out!r}, chain.run({n_steps}))
chain.close(); ctx.close()
"""


@pytest.mark.parametrize("rank,evaluator", [(72, "hausdorff"), (150, "collective")])
def test_step_schedule_at_large_ranks_does_not_change_the_chain(rank, evaluator, tmp_path):
    """Above rank 64 a chain step spreads over four streams (icp_chain_eval_step): factorisation and tails beside the evaluator's
    searches, the proposed state's decomposition — and the posterior of a proposed pose move — started ahead, on two eigen streams.
    With ICP_SPECULATION=0 the same step runs on one stream with nothing ahead.  Same seed, same records, bit for bit."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    recs = []
    for spec in (None, "0"):
        env = dict(os.environ)
        env.pop("ICP_SPECULATION", None)
        if spec is not None:
            env["ICP_SPECULATION"] = spec
        out = str(tmp_path / f"rec_{spec}.npy")
        script = _SCHEDULE_SCRIPT.format(root=ROOT, rank=rank, evaluator=evaluator, out=out, n_steps=150)
        subprocess.run([sys.executable, "-c", script], check=True, env=env, timeout=300)
        recs.append(np.load(out))
    a, b = recs
    assert a[:, 1].sum() > 20 and np.array_equal(a, b)


@pytest.mark.parametrize("rank", [72, 110, 150, 200])
def test_cholesky_root_sampler_at_large_ranks(pkg, rank):
    """The opt-in sampler above rank 64: the posterior's own factorisation hands its factor out (register-tiled kernel up to rank
    ~116, blocked kernel above), nothing is decomposed, and the proposal is a blocked back substitution: L·Lᵀ = M, the proposal is
    the closed form, logTransitionProbability is the eigen form's, and the BfmFittingPartial chain runs on it."""
    model = pkg.data.synthetic_face_model(grid=41, rank=rank)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    r = rank
    Q = model.basis * np.sqrt(model.variance)[None, :]
    P = np.linalg.inv(Q.T @ Q + 1e-5 * np.eye(r))
    pe = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
    pr = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True).setSampler("cholesky-root")
    rng = np.random.default_rng(rank)
    for seed in (5, 6):
        theta = face_theta(model, seed)
        a, b = pe.icpPosterior(theta), pr.icpPosterior(theta)
        assert np.array_equal(a.alpha, b.alpha) and np.array_equal(a.M, b.M) and np.array_equal(a.corr_id, b.corr_id)
        L = b.V
        assert np.allclose(np.tril(L), L) and np.abs(b.S * np.diag(L) - 1.0).max() <= 1e-13
        assert np.abs(L @ L.T - a.M).max() <= 1e-12 * np.abs(a.M).max()
        z = rng.normal(size=r)
        got = pr.propose(theta, z)
        w = a.alpha + np.linalg.solve(np.linalg.cholesky(0.5 * (a.M + a.M.T)).T, z)
        want = theta[10:] + 0.1 * ((w - 1e-5 * (P @ w)) - theta[10:])
        assert np.abs(got[10:] - want).max() <= 1e-8 * np.abs(want).max()
        assert pe.logTransitionProbability(theta, got) == pr.logTransitionProbability(theta, got)
    pe.close(); pr.close()
    setup = pkg.bfm_fitting_partial(model, target, evaluator="collective")
    setup.sampler = "cholesky-root"
    chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=9)
    rec = chain.run(80)
    assert np.all(np.isfinite(rec)) and 10 < rec[:, 1].sum() < 80 and (rec[:, 2] == 0).sum() > 20
    assert rec[-1, 3] > rec[0, 3]
    assert all(v == 0 for v in ctx.runtime_stats().values())
    chain.close(); ctx.close()
