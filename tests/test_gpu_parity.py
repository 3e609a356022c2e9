"""GPU parity tests: the HIP path, called through the C ABI, against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): correspondence / nearest indices BIT-EXACT; coefficients, transition densities
and likelihood values within 1e-5 relative (observed ~1e-12; the tighter bound used here is the regression guard).
"""
import numpy as np
import pytest

from conftest import make_theta, open_patch_target

pytestmark = pytest.mark.gpu

REL = 1e-9  # far inside the 1e-5 contract


@pytest.fixture(scope="module")
def ctx50(pkg, femur50):
    model, target = femur50
    c = pkg.IcpContext(model, target, device=0)
    yield c
    c.close()


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_transformed_mesh_bit_exact(ctx50, femur50, femur50_oracle):
    model, _ = femur50
    om, _ = femur50_oracle
    for seed in range(3):
        theta = make_theta(model, seed)
        assert np.array_equal(ctx50.transformedMesh(theta), om.instance(theta))


def test_pose_move_reposes_the_kept_deformations_bit_exact(ctx50, femur50, femur50_oracle):
    """A state that differs from a cached one in its pose only is not instantiated from the basis again: the kept deformations
    mean + Q·c of the cached state are posed anew (icp_ctx::state).  Same points as the oracle's instance, bit for bit — for pose
    moves off a state, for chains of them (every slot recycled) and for a shape move in between."""
    model, _ = femur50
    om, _ = femur50_oracle
    rng = np.random.default_rng(5)
    theta = make_theta(model, 4)
    assert np.array_equal(ctx50.transformedMesh(theta), om.instance(theta))
    for k in range(12):
        t = theta.copy()
        t[0] = 1.0 + 0.01 * rng.normal() if k % 4 == 3 else t[0]
        t[1:4] += rng.normal(size=3)
        t[4:7] += 0.02 * rng.normal(size=3)
        t[7:10] += rng.normal(size=3) if k % 5 == 4 else 0.0
        if k == 6:
            t[10:] += 0.05 * rng.normal(size=model.rank)   # a shape move: from the basis again
        assert np.array_equal(ctx50.transformedMesh(t), om.instance(t)), k
        theta = t


def test_vertex_normals(ctx50, femur50, femur50_oracle):
    model, _ = femur50
    om, _ = femur50_oracle
    theta = make_theta(model, 3)
    n = ctx50.vertexNormals(theta)
    no = om.vertex_normals(om.instance(theta))
    assert np.abs(n - no).max() < 1e-12


def test_closest_target_vertex_bit_exact(ctx50, femur50, oracle):
    model, target = femur50
    rng = np.random.default_rng(11)
    q = model.ref_points[rng.integers(0, model.n_points, 300)] + rng.normal(size=(300, 3)) * 3.0
    idx, d2 = ctx50.closestTargetVertex(q)
    io, d2o = oracle.nearest_vertex(q, target.points)
    assert np.array_equal(idx, io)
    assert np.array_equal(d2, d2o)
    # queries exactly ON vertices and exactly between two vertices (ties -> lowest index)
    q2 = np.concatenate([target.points[:5], 0.5 * (target.points[target.cells[:5, 0]] + target.points[target.cells[:5, 1]])])
    idx, d2 = ctx50.closestTargetVertex(q2)
    io, d2o = oracle.nearest_vertex(q2, target.points)
    assert np.array_equal(idx, io) and np.array_equal(d2, d2o)


def test_closest_point_on_target_bit_exact(ctx50, femur50, oracle):
    model, target = femur50
    rng = np.random.default_rng(12)
    q = model.ref_points[rng.integers(0, model.n_points, 200)] + rng.normal(size=(200, 3)) * 4.0
    cp, tri, d2 = ctx50.closestPointOnTarget(q)
    cpo, trio, d2o = oracle.closest_point_on_surface(q, target.points, target.cells)
    assert np.array_equal(tri, trio)
    assert np.array_equal(d2, d2o)
    assert np.array_equal(cp, cpo)
    # degenerate placements: on vertices, on edge midpoints, at triangle centroids (exact ties between triangles)
    c = target.cells[:40]
    q2 = np.concatenate([target.points[c[:, 0]], 0.5 * (target.points[c[:, 0]] + target.points[c[:, 1]]),
                         target.points[c].mean(axis=1)])
    cp, tri, d2 = ctx50.closestPointOnTarget(q2)
    cpo, trio, d2o = oracle.closest_point_on_surface(q2, target.points, target.cells)
    assert np.array_equal(tri, trio) and np.array_equal(d2, d2o) and np.array_equal(cp, cpo)


def test_queries_against_current_model_bit_exact(ctx50, femur50, femur50_oracle, oracle, pkg):
    model, target = femur50
    om, _ = femur50_oracle
    theta = make_theta(model, 5)
    x = om.instance(theta)
    tp = pkg.data.decimated_point_subset(target, 204)
    idx, d2 = ctx50.closestModelVertex(theta, tp)
    io, d2o = oracle.nearest_vertex(tp, x)
    assert np.array_equal(idx, io) and np.array_equal(d2, d2o)
    cp, tri, d2 = ctx50.closestPointOnModel(theta, tp)
    cpo, trio, d2o = oracle.closest_point_on_surface(tp, x, model.cells)
    assert np.array_equal(tri, trio) and np.array_equal(d2, d2o) and np.array_equal(cp, cpo)


def test_empty_and_single_queries(ctx50, femur50, oracle):
    _, target = femur50
    cp, tri, d2 = ctx50.closestPointOnTarget(np.zeros((0, 3)))
    assert cp.shape == (0, 3) and tri.shape == (0,)
    q = np.array([[1.0, 2.0, 3.0]])
    cp, tri, d2 = ctx50.closestPointOnTarget(q)
    cpo, trio, d2o = oracle.closest_point_on_surface(q, target.points, target.cells)
    assert tri[0] == trio[0] and d2[0] == d2o[0]


@pytest.mark.parametrize("direction", ["ModelSampling", "TargetSampling"])
@pytest.mark.parametrize("seed", [0, 1])
def test_posterior_propose_transition(pkg, ctx50, femur50, femur50_oracle, oracle, direction, seed):
    """a4/a5/a7/a8/a9 for the femur configuration (K = 2·rank, σt = 10, σn = 5, step 0.1)."""
    model, target = femur50
    om, ot = femur50_oracle
    r = model.rank
    theta = make_theta(model, 100 + seed)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    if direction == "ModelSampling":
        pp = oracle.proposal_params(0.1, 10.0, 5.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    else:
        pp = oracle.proposal_params(0.1, 10.0, 5.0, oracle.TARGET_SAMPLING, True, target_pts=tp)
    prop = pkg.NonRigidIcpProposal(ctx50, 0.1, 10.0, 5.0, 2 * r, direction, True, decimatedTargetPoints=tp)
    post = prop.icpPosterior(theta)
    po = oracle.icp_posterior(om, ot, pp, theta)
    assert np.array_equal(post.corr_id, po.corr_id)          # correspondence indices: bit-exact
    assert np.array_equal(post.keep, po.keep)
    assert np.array_equal(post.corr_aux, po.corr_aux)
    assert np.array_equal(post.corr_point, po.corr_pt)
    assert rel_err(post.M, po.M) < REL
    assert rel_err(post.alpha, po.alpha) < REL
    assert rel_err(post.S, po.S) < REL
    assert np.abs(post.V - po.V).max() < 1e-7                # eigenvectors: conditioning ~ eps / relative gap
    rng = np.random.default_rng(200 + seed)
    for z in (np.zeros(r), rng.normal(size=r)):
        got, corr = prop.propose(theta, z, return_correspondences=True)
        want = oracle.propose(om, ot, pp, theta, z)
        assert np.array_equal(got[:10], theta[:10])
        assert rel_err(got[10:], want[10:]) < 1e-7
        assert np.array_equal(corr, np.where(po.keep == 1, po.corr_id, -1))
        lt = prop.logTransitionProbability(theta, got)
        lo = oracle.log_transition(om, ot, pp, theta, want)
        assert abs(lt - lo) <= 1e-8 * abs(lo)
        lb = prop.logTransitionProbability(got, theta)
        lbo = oracle.log_transition(om, ot, pp, want, theta)
        assert abs(lb - lbo) <= 1e-8 * abs(lbo)
    # z = 0: the proposal moves the coefficients towards the posterior mean by exactly stepLength
    got0 = prop.propose(theta, np.zeros(r))
    assert rel_err(got0[10:], theta[10:] + 0.1 * (po.alpha - theta[10:])) < 1e-6
    # anything but the shape differs -> -inf (NonRigidIcpProposal.scala:72-74)
    other = got.copy()
    other[1] += 0.1
    assert prop.logTransitionProbability(theta, other) == -np.inf
    prop.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_independent_point_distance_evaluator(pkg, ctx50, femur50, femur50_oracle, oracle, mode):
    model, target = femur50
    om, ot = femur50_oracle
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 4 * r)
    ev = pkg.IndependentPointDistanceEvaluator(ctx50, 0.0, 2.0, mode, 4 * r, decimatedTargetPoints=tp)
    ep = oracle.evaluator_params(oracle.EVAL_INDEPENDENT, mode, n_model_ids=4 * r, target_pts=tp, p0=0.0, p1=2.0)
    for seed in (0, 1):
        theta = make_theta(model, 300 + seed)
        want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
        got = ev.logValue(theta)
        assert rc == 0 and abs(got - want) <= 1e-11 * abs(want)
        assert ev.logValue(theta) == got  # memoised (EvaluationCaching.scala:32-36)
    ev.close()


def test_hausdorff_evaluator(pkg, ctx50, femur50, femur50_oracle, oracle):
    model, _ = femur50
    om, ot = femur50_oracle
    ev = pkg.HausdorffDistanceEvaluator(ctx50, 1.0)
    ep = oracle.evaluator_params(oracle.EVAL_HAUSDORFF, 2, p0=1.0)
    for seed in (0, 1):
        theta = make_theta(model, 400 + seed)
        want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
        got, aux = ev.logValue(theta, return_aux=True)
        assert rc == 0 and abs(got - want) <= 1e-12 * abs(want)
    ev.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_boundary_aware_paths_on_open_target(pkg, femur50, oracle, mode):
    """Collective boundary-aware evaluator + boundary-aware ModelSampling proposal on a target with a hole."""
    model, target = femur50
    pts, cells = open_patch_target(target)
    tgt = pkg.data.TriangleMesh(pts, cells)
    assert pkg.data.boundary_vertex_flags(tgt).sum() > 0
    ctx = pkg.IcpContext(model, tgt, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(pts, cells)
    r = model.rank
    tp = pkg.data.decimated_point_subset(tgt, 4 * r)
    ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, mode, 4 * r, decimatedTargetPoints=tp)
    ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, mode, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
    theta = make_theta(model, 500)
    want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
    got, aux = ev.logValue(theta, return_aux=True)
    assert rc == 0 and abs(got - want) <= 1e-11 * abs(want)
    if mode == 0:
        pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        assert po.keep.sum() < po.keep.size, "test target should drop some correspondences"
        assert np.array_equal(post.keep, po.keep) and np.array_equal(post.corr_aux, po.corr_aux)
        assert rel_err(post.alpha, po.alpha) < REL
        z = np.random.default_rng(5).normal(size=r)
        assert rel_err(prop.propose(theta, z)[10:], oracle.propose(om, ot, pp, theta, z)[10:]) < 1e-7
        prop.close()
    ev.close()
    ctx.close()


def test_chain_eval_step_matches_separate_calls(pkg, ctx50, femur50):
    model, target = femur50
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    props = [pkg.NonRigidIcpProposal(ctx50, 0.1, 10.0, 5.0, 2 * r, d, True, decimatedTargetPoints=tp)
             for d in ("TargetSampling", "ModelSampling")]
    ev = pkg.IndependentPointDistanceEvaluator(ctx50, 0.0, 2.0, 0, 4 * r)
    cur = make_theta(model, 600, pose=False)
    prop_theta = props[0].propose(cur, np.random.default_rng(1).normal(size=r))
    val, fwd, bwd = pkg.chain_eval_step(ev, props, cur, prop_theta)
    assert val == ev.logValue(prop_theta)
    for i, p in enumerate(props):
        assert fwd[i] == p.logTransitionProbability(cur, prop_theta)
        assert bwd[i] == p.logTransitionProbability(prop_theta, cur)
    for p in props:
        p.close()
    ev.close()


@pytest.mark.parametrize("kind", ["independent", "collective"])
def test_chain_step_matches_separate_calls(pkg, femur50, kind):
    """icp_chain_step (five merged launches) == propose + logValue + 4 x logTransitionProbability.

    Same device code on both sides; bit-identical whenever no eigen-decomposition is involved.  The KL basis of a
    posterior is computed by a warm-started Jacobi iteration, so it depends (at rounding level, ~1e-15) on which
    decompositions ran before — and the merged path also prefetches the basis of the proposal that is NOT generating.
    Values downstream of a basis are therefore compared to 1e-10 relative, correspondence indices exactly."""
    model, target = femur50
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    mk_ctx = lambda: pkg.IcpContext(model, target, device=0)
    def mk(ctx):
        props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, d, True, decimatedTargetPoints=tp)
                 for d in ("TargetSampling", "ModelSampling")]
        if kind == "independent":
            ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 0, 4 * r)
        else:
            ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, 0, 4 * r)
        return props, ev
    ctx_a, ctx_b = mk_ctx(), mk_ctx()   # separate contexts: separate caches and search hints
    (props_a, ev_a), (props_b, ev_b) = mk(ctx_a), mk(ctx_b)
    cur = make_theta(model, 600, pose=False)
    rng = np.random.default_rng(3)
    close = lambda a, b: np.allclose(a, b, rtol=1e-10, atol=1e-11)   # observed: ~4e-13 (n·eps·cond of the warm-start transform)
    for step in range(6):
        gen = step % 3 - 1 if step else 0          # 0, 0, 1, -1, 0, 1
        z = rng.normal(size=r)
        if gen >= 0:
            prop_b = props_b[gen].propose(cur, z)
            prop_a, val, fwd, bwd = pkg.chain_step(ev_a, props_a, cur, generator=gen, z=z)
            assert close(prop_a, prop_b)
            # evaluate both paths at the SAME state from here on
            val, fwd, bwd = pkg.chain_eval_step(ev_a, props_a, cur, prop_b)
        else:
            prop_b = cur.copy()
            prop_b[10:] += 0.1 * z
            prop_a, val, fwd, bwd = pkg.chain_step(ev_a, props_a, cur, generator=-1, theta_prop=prop_b)
            assert np.array_equal(prop_a, prop_b)
        # no eigen-decomposition behind these: bit-identical
        assert val == ev_b.logValue(prop_b)
        for i, p in enumerate(props_b):
            assert fwd[i] == p.logTransitionProbability(cur, prop_b)
            assert bwd[i] == p.logTransitionProbability(prop_b, cur)
        # the caches filled by the merged launches serve the per-method entry points
        assert ev_a.logValue(prop_b) == val
        assert props_a[0].logTransitionProbability(cur, prop_b) == fwd[0]
        pa, pb = props_a[1].icpPosterior(prop_b, with_aux=False), props_b[1].icpPosterior(prop_b, with_aux=False)
        assert np.array_equal(pa.corr_id, pb.corr_id) and np.array_equal(pa.M, pb.M) and np.array_equal(pa.alpha, pb.alpha)
        assert close(pa.S, pb.S)
        if step % 2 == 0:
            cur = prop_b   # "accept"
    for p in props_a + props_b:
        p.close()
    ev_a.close(); ev_b.close(); ctx_a.close(); ctx_b.close()


def test_error_behaviour(pkg, ctx50, femur50):
    model, _ = femur50
    theta = make_theta(model, 1)
    bad = theta.copy()
    bad[12] = np.nan
    with pytest.raises(pkg._native.IcpNativeError) as e:
        ctx50.transformedMesh(bad)
    assert e.value.status == -3
    with pytest.raises(pkg._native.IcpNativeError) as e:
        pkg.NonRigidIcpProposal(ctx50, 0.1, -1.0, 5.0, 10, "ModelSampling")
    assert e.value.status == -1


def test_prelaunched_half_step_never_changes_results(pkg, femur50):
    """icp_chain_step_prelaunch issues launches 1-4 of a step ahead of the icp_chain_step that asks for it.  Whether that
    call then finds them (same arguments), finds others (different arguments: dropped), or none: bit-identical outputs —
    two contexts run the same sequence of steps, one with pre-launches (matching, mismatching, dropped), one without."""
    model, target = femur50
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    def mk():
        ctx = pkg.IcpContext(model, target, device=0)
        props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, d, True, decimatedTargetPoints=tp)
                 for d in ("TargetSampling", "ModelSampling")]
        return ctx, props, pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 0, 4 * r)
    (ctx_a, props_a, ev_a), (ctx_b, props_b, ev_b) = mk(), mk()
    cur = make_theta(model, 601, pose=False)
    rng = np.random.default_rng(11)
    zs = rng.normal(size=(8, r))
    # warm both sides up identically (posterior of `cur` on record, one basis computed)
    first_a = pkg.chain_step(ev_a, props_a, cur, generator=0, z=zs[0])
    first_b = pkg.chain_step(ev_b, props_b, cur, generator=0, z=zs[0])
    assert all(np.array_equal(x, y) for x, y in zip(first_a, first_b))
    for step in range(1, 8):
        gen = step % 2
        kind = step % 4
        if kind == 1:
            pkg.chain_step_prelaunch(ev_a, props_a, cur, generator=gen, z=zs[step])          # matches the call below
        elif kind == 2:
            pkg.chain_step_prelaunch(ev_a, props_a, cur, generator=gen, z=zs[step] + 1.0)    # does not: dropped
        elif kind == 3:
            pkg.chain_step_prelaunch(ev_a, props_a, cur, generator=1 - gen, z=zs[step])
            pkg.chain_step_prelaunch(ev_a, [], cur)                                          # dropped explicitly
        out_a = pkg.chain_step(ev_a, props_a, cur, generator=gen, z=zs[step])
        out_b = pkg.chain_step(ev_b, props_b, cur, generator=gen, z=zs[step])
        for x, y in zip(out_a, out_b):
            assert np.array_equal(np.asarray(x), np.asarray(y)), "step %d" % step
        if step % 3 == 0:   # "accept": continue from the proposed state on both sides
            cur = np.asarray(out_a[0]).copy()
    for o in props_a + props_b + [ev_a, ev_b, ctx_a, ctx_b]:
        o.close()


def test_hinted_searches_on_a_shuffled_distant_target(pkg, femur50, oracle):
    """The triangle filter's patch test (two levels) under conditions that are hard on it: triangles listed in
    random order (the sphere list is re-ordered inside the library: candidates must come back under their own ids), the
    whole scene 5 m from the origin (f32 spacing there: 0.5 µm — the slack of the patch test has to cover it), and a
    sequence of states so that the searches run with tight bounds from the previous winners.  Correspondence ids bit-exact,
    likelihood to 1e-9."""
    model, target = femur50
    rng = np.random.default_rng(21)
    shift = np.array([5000.0, -3000.0, 4000.0])
    cells = target.cells[rng.permutation(target.n_cells)]
    far = type(target)(target.points + shift, cells)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(far.points, far.cells)
    ctx = pkg.IcpContext(model, far, device=0)
    r = model.rank
    tp = pkg.data.decimated_point_subset(far, 2 * r)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, pkg.ModelSampling, True, decimatedTargetPoints=tp)
    pp = oracle.proposal_params(0.1, 10.0, 5.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r, decimatedTargetPoints=tp)
    ep = oracle.evaluator_params(oracle.EVAL_INDEPENDENT, oracle.MODEL_TO_TARGET, n_model_ids=4 * r, target_pts=tp, p0=0.0, p1=2.0)
    theta = make_theta(model, 3, shape_scale=0.3, pose=False)
    theta[1:4] = shift
    for step in range(5):
        post = oracle.icp_posterior(om, ot, pp, theta)
        got, corr = prop.propose(theta, rng.normal(size=r), return_correspondences=True)
        assert np.array_equal(corr, np.where(post.keep == 1, post.corr_id, -1)), f"step {step}: correspondence ids differ"
        lv, (lvo, _) = ev.logValue(theta), oracle.evaluator_log_value(om, ot, ep, theta)
        assert abs(lv - lvo) <= 1e-9 * abs(lvo), (step, lv, lvo)
        theta = theta.copy()
        theta[10:] += 0.05 * rng.normal(size=r)   # a nearby state: the next searches start from this one's winners
    prop.close(); ev.close(); ctx.close()


def test_caller_supplied_rotation_convention(pkg, femur50):
    """SURVEY.md §8b "pose across the boundary": the host may hand over Scalismo's own rotation matrix for a theta's Euler angles
    (icp_ctx_set_rotation), so the convention of Rotation(phi, theta, psi, centre) stays Scalismo's.  With the library's own
    Rz·Ry·Rx registered nothing changes; with another convention the mesh is the one that matrix gives, and the proposal's
    inverse pose (NonRigidIcpProposal.scala:142) uses the same matrix (z = 0 proposal moves towards the posterior mean)."""
    model, target = femur50
    ctx = pkg.IcpContext(model, target, device=0)
    theta = make_theta(model, 5, pose=True)
    theta[4:7] = [0.3, -0.2, 0.15]
    phi, th, psi = theta[4:7]
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(psi), -s(psi)], [0, s(psi), c(psi)]])
    Ry = np.array([[c(th), 0, s(th)], [0, 1, 0], [-s(th), 0, c(th)]])
    Rz = np.array([[c(phi), -s(phi), 0], [s(phi), c(phi), 0], [0, 0, 1]])
    x_native = ctx.transformedMesh(theta)
    ctx.close()

    def posed(R):
        shape = model.instance(theta[10:])
        ctr = theta[7:10]
        return theta[0] * ((shape - ctr) @ R.T + ctr + theta[1:4])

    assert np.abs(x_native - posed(Rz @ Ry @ Rx)).max() < 1e-10   # the library's documented convention
    for R, differs in ((Rz @ Ry @ Rx, False), (Rx @ Ry @ Rz, True)):
        ctx = pkg.IcpContext(model, target, device=0)
        ctx.setRotation(theta[4:7], R)
        x = ctx.transformedMesh(theta)
        assert np.abs(x - posed(R)).max() < 1e-10
        assert (np.abs(x - x_native).max() > 1.0) == differs
        other = theta.copy(); other[4] += 0.01       # a triple without an entry: native convention
        xo = ctx.transformedMesh(other)
        assert np.all(np.isfinite(xo))
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * model.rank, "ModelSampling", True)
        post = prop.icpPosterior(theta, with_aux=False)
        got = prop.propose(theta, np.zeros(model.rank))
        assert np.abs(got[10:] - (theta[10:] + 0.1 * (post.alpha - theta[10:]))).max() < 1e-6 * max(1.0, np.abs(post.alpha).max())
        # the correspondences' observations are the target points taken back through THIS rotation: R^T((q - t) - ctr) + ctr
        q = post.corr_point[0]
        back = ((q / 1.0 - theta[1:4]) - theta[7:10]) @ R + theta[7:10]
        assert np.all(np.isfinite(back))
        prop.close()
        ctx.setRotation(theta[4:7], None)
        with pytest.raises(pkg._native.IcpNativeError):
            ctx.setRotation(theta[4:7], 2.0 * R)     # not a rotation
        ctx.close()


def test_rotation_change_drops_what_was_cached_under_the_old_matrix(pkg, femur50):
    """icp_ctx_set_rotation on a triple that is already in use — replacement, withdrawal — must not leave instances, posteriors or
    likelihood values computed with the previous matrix behind: a second context that only ever saw the final matrix
    gives the reference values."""
    model, target = femur50
    r = model.rank
    theta = make_theta(model, 9, pose=True)
    theta[4:7] = [0.25, -0.1, 0.2]
    phi, th, psi = theta[4:7]
    c, s = np.cos, np.sin
    Rx = np.array([[1, 0, 0], [0, c(psi), -s(psi)], [0, s(psi), c(psi)]])
    Ry = np.array([[c(th), 0, s(th)], [0, 1, 0], [-s(th), 0, c(th)]])
    Rz = np.array([[c(phi), -s(phi), 0], [s(phi), c(phi), 0], [0, 0, 1]])
    R1, R2 = Rx @ Ry @ Rz, Ry @ Rx @ Rz
    tp = pkg.data.decimated_point_subset(target, 4 * r)

    def values(ctx, prop, ev):
        post = prop.icpPosterior(theta, with_aux=False)
        return ctx.transformedMesh(theta), post.alpha.copy(), post.corr_point.copy(), ev.logValue(theta)

    def fresh(R):
        ctx = pkg.IcpContext(model, target, device=0)
        if R is not None:
            ctx.setRotation(theta[4:7], R)
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "ModelSampling", True)
        ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 0, 4 * r, decimatedTargetPoints=tp)
        out = values(ctx, prop, ev)
        prop.close(); ev.close(); ctx.close()
        return out

    want1, want2, want_native = fresh(R1), fresh(R2), fresh(None)
    assert np.abs(want1[0] - want2[0]).max() > 1.0 and abs(want1[3] - want2[3]) > 1e-3
    ctx = pkg.IcpContext(model, target, device=0)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "ModelSampling", True)
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 0, 4 * r, decimatedTargetPoints=tp)
    for R, want in ((None, want_native), (R1, want1), (R2, want2), (R1, want1), (None, want_native)):   # first use, insert, replace, replace, withdraw
        if R is not None or want is want_native:
            ctx.setRotation(theta[4:7], R)
        got = values(ctx, prop, ev)
        assert np.array_equal(got[0], want[0]) and np.array_equal(got[2], want[2])
        assert np.array_equal(got[1], want[1]) and got[3] == want[3]
    prop.close(); ev.close(); ctx.close()


def test_cholesky_root_sampler_has_the_posterior_covariance(pkg, femur50):
    """Opt-in sampler (icp_proposal_set_sampler, NOT the reference's arithmetic): z multiplies W = D·L⁻ᵀ (M = L·Lᵀ) instead of the KL
    basis V√S.  W·Wᵀ must equal V·S·Vᵀ (= D·M⁻¹·D) to 1e-12 — same proposal distribution —, the proposal must be the closed form
    with that root, and logTransitionProbability, which does not depend on the root, must not change at all."""
    model, target = femur50
    r = model.rank
    ctx = pkg.IcpContext(model, target, device=0)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    rng = np.random.default_rng(3)
    Q = model.basis * np.sqrt(model.variance)[None, :]
    G = Q.T @ Q
    sigma2 = 1e-5
    P = np.linalg.inv(G + sigma2 * np.eye(r))
    sl = np.sqrt(model.variance)
    for direction in ("ModelSampling", "TargetSampling"):
        pe = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, direction, True, decimatedTargetPoints=tp)
        pr = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, direction, True, decimatedTargetPoints=tp).setSampler("cholesky-root")
        for seed in (1, 2):
            theta = make_theta(model, seed, pose=True)
            a, b = pe.icpPosterior(theta), pr.icpPosterior(theta)
            assert np.array_equal(a.corr_id, b.corr_id) and np.array_equal(a.alpha, b.alpha) and np.array_equal(a.M, b.M)
            cov_e = (a.V * a.S[None, :]) @ a.V.T
            # root mode: the view returns the factor itself, V = L (M = L·Lᵀ, lower triangular) and S = 1/diag(L); W = D·L⁻ᵀ
            Lg = b.V
            assert np.allclose(np.tril(Lg), Lg) and np.abs(b.S * np.diag(Lg) - 1.0).max() <= 1e-14
            assert np.abs(Lg @ Lg.T - a.M).max() <= 1e-13 * np.abs(a.M).max()
            W = sl[:, None] * np.linalg.inv(Lg).T
            cov_r = W @ W.T
            assert np.abs(cov_r - cov_e).max() <= 1e-12 * np.abs(cov_e).max()
            want_cov = (sl[:, None] * np.linalg.inv(a.M)) * sl[None, :]
            assert np.abs(cov_r - want_cov).max() <= 1e-12 * np.abs(want_cov).max()
            z = rng.normal(size=r)
            got = pr.propose(theta, z)
            L = np.linalg.cholesky(0.5 * (a.M + a.M.T))
            w = a.alpha + np.linalg.solve(L.T, z)
            cnew = w - sigma2 * (P @ w)
            want = theta[10:] + 0.1 * (cnew - theta[10:])
            assert np.abs(got[10:] - want).max() <= 1e-9 * np.abs(want).max()
            assert np.array_equal(got[:10], theta[:10])
            # root-free transition density: bit-identical under both samplers, and finite for a root-sampled proposal
            other = pe.propose(theta, z)
            for to in (got, other):
                assert pe.logTransitionProbability(theta, to) == pr.logTransitionProbability(theta, to)
            assert np.isfinite(pr.logTransitionProbability(theta, got))
        # switching back: the eigen form again, the same values as the proposal that never switched
        pr.setSampler("eigen")
        c = pr.icpPosterior(theta)
        assert np.abs(c.S - a.S).max() <= 1e-10 * a.S.max() and np.abs(c.V - a.V).max() <= 1e-7
        pe.close(); pr.close()
    ctx.close()


def test_cholesky_root_sampler_chain(pkg, femur50):
    """The metric chain with the opt-in sampler: a valid Metropolis–Hastings chain (same acceptance behaviour as the eigen form on
    average, different realisation), identical step by step through the merged step, the per-method calls and the batched step."""
    model, target = femur50
    n = 300
    recs = {}
    for fused in (2, 0):
        setup = pkg.femur_icp_proposal_registration(model, target, fused=fused)
        setup.sampler = "cholesky-root"
        ctx = pkg.IcpContext(model, target, device=0)
        chain = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024)
        recs[fused] = chain.run(n)
        assert all(v == 0 for v in ctx.runtime_stats().values())
        chain.close(); ctx.close()
    assert np.array_equal(recs[2][:, :3], recs[0][:, :3])
    assert np.abs(recs[2][:, 14:] - recs[0][:, 14:]).max() <= 1e-9 * np.abs(recs[0][:, 14:]).max()
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    ctx = pkg.IcpContext(model, target, device=0)
    ref = pkg.SamplingRegistration(ctx, setup, pkg.initial_parameters(model), seed=1024).run(n)
    ctx.close()
    acc_root, acc_eig = recs[2][:, 1].mean(), ref[:, 1].mean()
    assert 0.15 < acc_root < 0.9 and abs(acc_root - acc_eig) < 0.15
    assert recs[2][-1, 3] > recs[2][0, 3] + 100.0          # the chain climbs like the eigen form does
    assert abs(recs[2][-50:, 3].mean() - ref[-50:, 3].mean()) < 0.05 * abs(ref[-50:, 3].mean())
    # batched: chain by chain the records of the single chains
    setup = pkg.femur_icp_proposal_registration(model, target, fused=2)
    setup.sampler = "cholesky-root"
    B = 9
    def make():
        ctxs = [pkg.IcpContext(model, target, device=0) for _ in range(B)]
        return ctxs, [pkg.SamplingRegistration(ctxs[i], setup, pkg.random_initial_parameters(model, i), seed=50 + i) for i in range(B)]
    ctxs, chains = make()
    single = [c.run(40) for c in chains]
    [c.close() for c in chains]; [c.close() for c in ctxs]
    ctxs, chains = make()
    got = pkg.run_chains_batched(chains, 40)
    assert all(np.array_equal(a, b) for a, b in zip(got, single))
    [c.close() for c in chains]; [c.close() for c in ctxs]
