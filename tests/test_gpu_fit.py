"""GPU: deterministic non-rigid ICP (api/other/IcpBasedSurfaceFitting.scala:46-126) against the oracle."""
import numpy as np
import pytest

from conftest import make_theta

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("direction", ["ModelSampling", "TargetSampling"])
def test_deterministic_icp_matches_oracle(pkg, oracle, femur50, femur50_oracle, direction):
    model, target = femur50
    om, ot = femur50_oracle
    rng = np.random.default_rng(17)
    ids = rng.integers(0, model.n_points, 300).astype(np.int32)            # stand-in for UniformMeshSampler3D + nearest vertex (:51-53)
    tps = target.points[rng.integers(0, target.n_points, 300)] + rng.normal(size=(300, 3)) * 0.01
    theta0 = pkg.initial_parameters(model)
    theta0[10:] = 0.3 * rng.normal(size=model.rank)
    ctx = pkg.IcpContext(model, target, device=0)
    fit = pkg.IcpBasedSurfaceFitting(ctx, stepLength=1.0, projectionDirection=direction, modelPointIds=ids, targetPointSamples=tps)
    n_it, seq = 4, (1.0, 0.1, 0.01)
    got = fit.runfitting(n_it, seq, theta0)
    want = oracle.fit_deterministic(om, ot, theta0, n_it, seq, direction=0 if direction == "ModelSampling" else 1, model_ids=ids,
                                    target_pts=tps, step_length=1.0)
    assert np.array_equal(got[:10], theta0[:10])
    assert np.abs(got[10:] - want[10:]).max() <= 1e-7 * np.abs(want[10:]).max()
    # the fit moved the model onto the target: the mean vertex-to-surface distance drops well below the start's
    def avg_dist(theta):
        x = ctx.transformedMesh(theta)
        _, _, d2 = ctx.closestPointOnTarget(x[::8])
        return np.sqrt(d2).mean()
    assert avg_dist(got) < 0.25 * avg_dist(theta0)
    # step length 0.5, zero extra iterations: one recursion per sigma (numIterations + 1 = 1), still equal to the oracle
    fit2 = pkg.IcpBasedSurfaceFitting(ctx, stepLength=0.5, projectionDirection=direction, modelPointIds=ids, targetPointSamples=tps)
    g2 = fit2.runfitting(0, (1.0,), theta0)
    w2 = oracle.fit_deterministic(om, ot, theta0, 0, (1.0,), direction=0 if direction == "ModelSampling" else 1, model_ids=ids,
                                  target_pts=tps, step_length=0.5)
    assert np.abs(g2[10:] - w2[10:]).max() <= 1e-8 * np.abs(w2[10:]).max()
    ctx.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_posterior_variability_matches_oracle(pkg, oracle, femur50, femur50_oracle, mode):
    """SURVEY.md §8f next row 3 (apps/util/PosteriorVariability.scala:30-73) on 25 chain-like samples."""
    model, target = femur50
    om, _ = femur50_oracle
    thetas = np.stack([make_theta(model, 700 + s, shape_scale=0.3) for s in range(25)])
    ctx = pkg.IcpContext(model, target, device=0)
    got = pkg.posterior_variability(ctx, thetas, mode=mode, theta_ref=thetas[3])
    want = oracle.posterior_variability(om, thetas, mode=mode, theta_ref=thetas[3])
    assert got.shape == (model.n_points,) and np.all(got >= 0)
    assert np.abs(got - want).max() <= 1e-12 * np.abs(want).max()
    ctx.close()


def test_registration_metrics_match_oracle(pkg, oracle, femur50):
    """SURVEY.md §8f next row 4 (api/other/RegistrationComparison.scala:24-49), on a target WITH a boundary."""
    from conftest import open_patch_target
    model, target = femur50
    pts, cells = open_patch_target(target)
    tgt = pkg.data.TriangleMesh(pts, cells)
    ctx = pkg.IcpContext(model, tgt, device=0)
    om = oracle.OracleModel.from_model(model)
    theta = make_theta(model, 811)
    got = pkg.evaluate_reconstruction_to_ground_truth(ctx, theta)
    x = om.instance(theta)
    cp, _, d2 = oracle.closest_point_on_surface(x, pts, cells)
    _, _, d2r = oracle.closest_point_on_surface(pts, x, model.cells)
    nn, _ = oracle.nearest_vertex(cp, pts)
    keep = pkg.data.boundary_vertex_flags(tgt)[nn] == 0
    d = np.sqrt(d2)
    assert abs(got["average2surface"] - d.sum() / len(d)) <= 1e-12 * d.mean()
    assert got["hausdorff"] == max(d.max(), np.sqrt(d2r).max())
    assert got["kept"] == keep.sum() and 0 < keep.sum() < len(d)
    assert abs(got["average2surface_boundary_aware"] - d[keep].sum() / keep.sum()) <= 1e-12 * d.mean()
    assert got["max_boundary_aware"] == d[keep].max()
    ctx.close()
