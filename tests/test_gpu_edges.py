"""GPU: edge cases of the reference's path that the main parity files do not reach, each against the oracle.

* boundaryAware = false on a target WITH boundary, both directions (NonRigidIcpProposal.scala:104,124: the boundary flag is
  computed and then ignored);
* TargetSampling against a model that HAS boundary vertices (:119 — the open face stand-in), boundaryAware true and false;
* an evaluator that drops every point: the reference's `.max` of an empty list throws
  (CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:51,63) -> ICP_ERR_EMPTY here, rc -2 in the oracle;
* a scale s != 1 through icpPosterior (:142 un-poses the target-side point WITHOUT the scale: the oracle follows that).
"""
import numpy as np
import pytest

from conftest import make_theta, open_patch_target

pytestmark = pytest.mark.gpu

REL = 1e-9
ICP_ERR_EMPTY = -5  # include/icp_proposal.h


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(np.asarray(b)).max(), 1e-300)


def check_posterior(post, po):
    assert np.array_equal(post.corr_id, po.corr_id)
    assert np.array_equal(post.keep, po.keep)
    assert np.array_equal(post.corr_aux, po.corr_aux)
    assert np.array_equal(post.corr_point, po.corr_pt)
    assert rel_err(post.M, po.M) < REL
    assert rel_err(post.alpha, po.alpha) < REL
    assert rel_err(post.S, po.S) < REL


@pytest.fixture(scope="module")
def open_femur(pkg, femur50, oracle):
    model, target = femur50
    pts, cells = open_patch_target(target)
    tgt = pkg.data.TriangleMesh(pts, cells)
    ctx = pkg.IcpContext(model, tgt, device=0)
    yield model, tgt, ctx, oracle.OracleModel.from_model(model), oracle.OracleMesh(pts, cells)
    ctx.close()


@pytest.mark.parametrize("direction", ["ModelSampling", "TargetSampling"])
def test_boundary_aware_false_on_open_target(pkg, oracle, open_femur, direction):
    model, tgt, ctx, om, ot = open_femur
    r = model.rank
    theta = make_theta(model, 510)
    tp = pkg.data.decimated_point_subset(tgt, 2 * r)
    dirn = oracle.MODEL_SAMPLING if direction == "ModelSampling" else oracle.TARGET_SAMPLING
    posts = {}
    for aware in (False, True):
        pp = oracle.proposal_params(0.1, 6.0, 3.0, dirn, aware, n_model_ids=2 * r, target_pts=tp)
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, direction, aware, decimatedTargetPoints=tp)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        check_posterior(post, po)
        z = np.random.default_rng(6).normal(size=r)
        got, want = prop.propose(theta, z), oracle.propose(om, ot, pp, theta, z)
        assert rel_err(got[10:], want[10:]) < 1e-7
        lt, lo = prop.logTransitionProbability(theta, got), oracle.log_transition(om, ot, pp, theta, want)
        assert abs(lt - lo) <= 1e-8 * abs(lo)
        posts[aware] = post
        prop.close()
    assert posts[False].keep.all(), "boundaryAware = false keeps every correspondence (:104,124)"
    if direction == "ModelSampling":
        assert not posts[True].keep.all(), "the open target should drop some correspondences when boundary-aware"
        assert np.abs(posts[True].alpha - posts[False].alpha).max() > 0.0
    # the whole step, through whichever path the library routes this configuration (merged launches or per stage): same numbers
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 0, 4 * r, decimatedTargetPoints=pkg.data.decimated_point_subset(tgt, 4 * r))
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, direction, False, decimatedTargetPoints=tp)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, dirn, False, n_model_ids=2 * r, target_pts=tp)
    z = np.random.default_rng(7).normal(size=r)
    prop_theta, lv, fwd, bwd = pkg.chain_step(ev, [prop], theta, generator=0, z=z)
    want = oracle.propose(om, ot, pp, theta, z)
    assert rel_err(prop_theta[10:], want[10:]) < 1e-7
    assert abs(fwd[0] - oracle.log_transition(om, ot, pp, theta, want)) <= 1e-7 * abs(fwd[0])
    assert abs(bwd[0] - oracle.log_transition(om, ot, pp, want, theta)) <= 1e-7 * abs(bwd[0])
    prop.close()
    ev.close()


@pytest.fixture(scope="module")
def small_face(pkg, oracle):
    model = pkg.data.synthetic_face_model(grid=41, rank=40)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    yield model, target, ctx, oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    ctx.close()


@pytest.mark.parametrize("aware", [True, False])
def test_target_sampling_against_a_model_with_boundary(pkg, oracle, small_face, aware):
    """NonRigidIcpProposal.scala:118-124: the nearest MODEL vertex of a target point may be a boundary vertex of the (open) model."""
    model, target, ctx, om, ot = small_face
    r = model.rank
    assert om.boundary().sum() > 0, "the face stand-in is an open sheet"
    # target points spread over the whole target, so that some land on the model's rim
    tp = pkg.data.decimated_point_subset(target, 6 * r)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.TARGET_SAMPLING, aware, target_pts=tp)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 6 * r, "TargetSampling", aware, decimatedTargetPoints=tp)
    dropped = 0
    for seed in (21, 22):
        theta = make_theta(model, seed, shape_scale=0.4)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        check_posterior(post, po)
        dropped += int((po.keep == 0).sum())
        z = np.random.default_rng(seed).normal(size=r)
        got, want = prop.propose(theta, z), oracle.propose(om, ot, pp, theta, z)
        assert rel_err(got[10:], want[10:]) < 1e-7
        lt, lo = prop.logTransitionProbability(theta, got), oracle.log_transition(om, ot, pp, theta, want)
        assert abs(lt - lo) <= 1e-7 * abs(lo)
    if aware:
        assert dropped > 0, "some target points should have a boundary vertex of the model as their nearest vertex"
    else:
        assert dropped == 0
    prop.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_evaluator_that_drops_every_point_is_empty(pkg, oracle, femur50, mode):
    """A target of three separate triangles: every vertex is a boundary vertex, so the boundary-aware evaluator keeps nothing in the
    model -> target direction (Collective…Evaluator.scala:44-51: `.max` of an empty list throws).  The target -> model direction tests
    the TARGET's flags with a MODEL-sample vertex id (:59): ids 0..8 are flagged, every other id is interior."""
    model, target = femur50
    rng = np.random.default_rng(3)
    tris = target.cells[rng.choice(target.cells.shape[0], 3, replace=False)]
    pts = target.points[tris.ravel()].copy()
    cells = np.arange(9, dtype=np.int32).reshape(3, 3)
    tgt = pkg.data.TriangleMesh(pts, cells)
    assert pkg.data.boundary_vertex_flags(tgt).all()
    ctx = pkg.IcpContext(model, tgt, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(pts, cells)
    r = model.rank
    tp = pts.copy()
    ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(ctx, 0.1, 0.3, 1.0, mode, 4 * r, decimatedTargetPoints=tp)
    ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, mode, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
    theta = make_theta(model, 520)
    want, rc = oracle.evaluator_log_value(om, ot, ep, theta)
    nat = pkg._native
    import ctypes as C
    out, aux = C.c_double(), np.zeros(4)
    th = np.ascontiguousarray(theta, dtype=np.float64)
    st = nat.lib().icp_evaluator_log_value(ev.h, th.ctypes.data_as(C.POINTER(C.c_double)), C.byref(out),
                                           aux.ctypes.data_as(C.POINTER(C.c_double)))
    if mode in (0, 2):
        assert rc == -2 and st == ICP_ERR_EMPTY
        with pytest.raises(nat.IcpNativeError) as ei:
            ev.logValue(theta)
        assert ei.value.status == ICP_ERR_EMPTY
        # the same through the whole-step entry point: the step's status is the evaluator's
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", False)
        z = np.random.default_rng(8).normal(size=r)
        with pytest.raises(nat.IcpNativeError) as ei:
            pkg.chain_step(ev, [prop], theta, generator=0, z=z)
        assert ei.value.status == ICP_ERR_EMPTY
        prop.close()
    else:
        assert rc == 0 and st == 0 and abs(out.value - want) <= 1e-11 * abs(want)
    ev.close()
    ctx.close()


@pytest.mark.parametrize("direction", ["ModelSampling", "TargetSampling"])
def test_posterior_with_a_scale(pkg, oracle, femur50, femur50_oracle, direction):
    """theta[0] = s = 1.02: the instance is scaled (ModelFittingParameters.scala:88-106), the target-side point is un-posed by the
    inverse RIGID transform only (NonRigidIcpProposal.scala:142)."""
    model, target = femur50
    om, ot = femur50_oracle
    ctx = pkg.IcpContext(model, target, device=0)
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    dirn = oracle.MODEL_SAMPLING if direction == "ModelSampling" else oracle.TARGET_SAMPLING
    pp = oracle.proposal_params(0.1, 10.0, 5.0, dirn, True, n_model_ids=2 * r, target_pts=tp)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, direction, True, decimatedTargetPoints=tp)
    theta = make_theta(model, 530)
    theta[0] = 1.02
    assert np.array_equal(ctx.transformedMesh(theta), om.instance(theta))
    post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
    check_posterior(post, po)
    one = theta.copy()
    one[0] = 1.0
    assert np.abs(post.alpha - prop.icpPosterior(one).alpha).max() > 1e-6, "the scale must matter"
    z = np.random.default_rng(9).normal(size=r)
    got, want = prop.propose(theta, z), oracle.propose(om, ot, pp, theta, z)
    assert got[0] == 1.02 and rel_err(got[10:], want[10:]) < 1e-7
    lt, lo = prop.logTransitionProbability(theta, got), oracle.log_transition(om, ot, pp, theta, want)
    assert abs(lt - lo) <= 1e-8 * abs(lo)
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, 2, 4 * r, decimatedTargetPoints=pkg.data.decimated_point_subset(target, 4 * r))
    ep = oracle.evaluator_params(oracle.EVAL_INDEPENDENT, 2, n_model_ids=4 * r, target_pts=pkg.data.decimated_point_subset(target, 4 * r),
                                 p0=0.0, p1=2.0)
    wv, rc = oracle.evaluator_log_value(om, ot, ep, theta)
    assert rc == 0 and abs(ev.logValue(theta) - wv) <= 1e-11 * abs(wv)
    ev.close()
    prop.close()
    ctx.close()


def test_context_handed_to_another_target(pkg, oracle, femur50):
    """icp_ctx_set_target: a batch registration's contexts go from target to target (model data, scratch and streams stay).  Everything
    computed afterwards is what a fresh context on the new target computes — searches, posterior, evaluator — and matches the oracle;
    with a proposal or an evaluator still alive the call is refused."""
    model, target = femur50
    pts, cells = open_patch_target(target)
    other = pkg.data.TriangleMesh(pts, cells)
    r = model.rank
    ctx = pkg.IcpContext(model, target, device=0)
    theta = make_theta(model, 540)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
    prop.icpPosterior(theta)  # (caches against the first target: state slot, hints, memo)
    import ctypes as C
    nat = pkg._native
    td = nat.MeshDesc(other.n_points, other.n_cells, other.points.ctypes.data_as(nat.c_double_p), other.cells.ctypes.data_as(C.POINTER(C.c_int32)))
    assert nat.lib().icp_ctx_set_target(ctx.h, C.byref(td)) == -1  # a proposal made for the old target is still alive
    ctx.setTarget(other)       # (closes it first)
    fresh = pkg.IcpContext(model, other, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(pts, cells)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    tp = pkg.data.decimated_point_subset(other, 4 * r)
    for c in (ctx, fresh):
        p = pkg.NonRigidIcpProposal(c, 0.1, 6.0, 3.0, 2 * r, "ModelSampling", True)
        post, po = p.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        check_posterior(post, po)
        assert not post.keep.all()
        ev = pkg.CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(c, 0.1, 0.3, 1.0, 2, 4 * r, decimatedTargetPoints=tp)
        ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, 2, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
        wv, rc = oracle.evaluator_log_value(om, ot, ep, theta)
        assert rc == 0 and abs(ev.logValue(theta) - wv) <= 1e-11 * abs(wv)
        z = np.random.default_rng(1).normal(size=r)
        got, lv, fwd, bwd = pkg.chain_step(ev, [p], theta, generator=0, z=z)
        want = oracle.propose(om, ot, pp, theta, z)
        assert rel_err(got[10:], want[10:]) < 1e-7
        ev.close(); p.close()
    fresh.close()
    ctx.close()
