"""GPU: the HIP path against the committed build-oracle golden vectors (tests/golden/oracle_vectors_femur50.npz),
i.e. without running the oracle at all, plus size-independent properties at BASELINE.json's full size."""
import os

import numpy as np
import pytest

from conftest import ROOT, make_theta

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors_femur50.npz"))


@pytest.fixture(scope="module")
def ctx50(pkg, femur50):
    model, target = femur50
    c = pkg.IcpContext(model, target, device=0)
    yield c
    c.close()


def test_geometry_golden(ctx50):
    assert np.array_equal(ctx50.transformedMesh(GOLD["thetas"][0]), GOLD["instance0"])
    idx, d2 = ctx50.closestTargetVertex(GOLD["queries"])
    assert np.array_equal(idx, GOLD["nn_idx"]) and np.array_equal(d2, GOLD["nn_d2"])
    cp, tri, d2 = ctx50.closestPointOnTarget(GOLD["queries"])
    assert np.array_equal(tri, GOLD["cp_tri"]) and np.array_equal(cp, GOLD["cp"]) and np.array_equal(d2, GOLD["cp_d2"])


@pytest.mark.parametrize("name", ["model", "target"])
def test_proposal_golden(pkg, ctx50, femur50, name):
    model, _ = femur50
    r = model.rank
    prop = pkg.NonRigidIcpProposal(ctx50, 0.1, 10.0, 5.0, 2 * r, "ModelSampling" if name == "model" else "TargetSampling",
                                   True, decimatedTargetPoints=GOLD["target_pts"])
    for i in range(3):
        theta = GOLD["thetas"][i]
        post = prop.icpPosterior(theta, with_aux=False)
        assert np.array_equal(post.corr_id, GOLD[f"{name}_corr_id"][i])
        assert np.allclose(post.alpha, GOLD[f"{name}_alpha"][i], rtol=1e-8, atol=1e-11)
        assert np.allclose(post.S, GOLD[f"{name}_S"][i], rtol=1e-9)
        got = prop.propose(theta, GOLD["zs"][i])
        want = GOLD[f"{name}_proposed"][i]
        assert np.abs(got - want).max() <= 1e-7 * np.abs(want[10:]).max()
        assert np.isclose(prop.logTransitionProbability(theta, want), GOLD[f"{name}_logT_fwd"][i], rtol=1e-8)
        assert np.isclose(prop.logTransitionProbability(want, theta), GOLD[f"{name}_logT_bwd"][i], rtol=1e-8)
    prop.close()


def test_evaluators_golden(pkg, ctx50, femur50):
    model, target = femur50
    r = model.rank
    tp4 = pkg.data.decimated_point_subset(target, 4 * r)
    for mode in (0, 1, 2):
        ev = pkg.IndependentPointDistanceEvaluator(ctx50, 0.0, 2.0, mode, 4 * r, decimatedTargetPoints=tp4)
        got = np.asarray([ev.logValue(t) for t in GOLD["thetas"]])
        assert np.allclose(got, GOLD[f"indep_mode{mode}"], rtol=1e-11)
        ev.close()
    ev = pkg.HausdorffDistanceEvaluator(ctx50, 1.0)
    assert np.allclose([ev.logValue(t) for t in GOLD["thetas"]], GOLD["hausdorff"], rtol=1e-12)
    ev.close()
    pr = pkg.ModelPriorEvaluator(r)
    assert np.allclose([pr.logValue(t) for t in GOLD["thetas"]], GOLD["prior"], rtol=1e-14)


def test_chain_golden_200_steps(pkg, ctx50, femur50):
    """a16: the committed 200-step femur-50 chain (seed 1024): identical decisions, states within 1e-5 relative"""
    model, target = femur50
    setup = pkg.femur_icp_proposal_registration(model, target)
    chain = pkg.SamplingRegistration(ctx50, setup, pkg.initial_parameters(model), 1024)
    rec = chain.run(200)
    assert np.array_equal(rec[:, 1].astype(np.uint8), GOLD["chain_accepted"])
    assert np.array_equal(rec[:, 2].astype(np.int32), GOLD["chain_component"])
    scale = np.abs(GOLD["chain_states"][:, 10:]).max()
    assert np.abs(rec[:, 14:] - GOLD["chain_states"][:, 10:]).max() <= 1e-5 * scale
    assert np.abs(rec[:, 3] - GOLD["chain_logp"]).max() <= 1e-6 * np.abs(GOLD["chain_logp"]).max()
    chain.close()


# ------------------------------------------------------------------ BASELINE.json full size (58,322-vertex target)

@pytest.fixture(scope="module")
def full(pkg):
    model, target = pkg.data.synthetic_femur_target()
    ctx = pkg.IcpContext(model, target, device=0)
    yield model, target, ctx
    ctx.close()


def test_full_size_search_properties(pkg, full, oracle):
    model, target, ctx = full
    theta = make_theta(model, 42)
    x = ctx.transformedMesh(theta)
    q = x[:408]
    cp, tri, d2 = ctx.closestPointOnTarget(q)
    # idempotence: the closest point of a surface point is itself; its distance is exactly 0 or rounding-small
    cp2, tri2, d22 = ctx.closestPointOnTarget(cp)
    assert np.abs(cp2 - cp).max() < 1e-9 and d22.max() < 1e-18 * 1e6
    # optimality against the nearest vertex (an upper bound) and the oracle on a small slice
    idx, dv2 = ctx.closestTargetVertex(q)
    assert np.all(d2 <= dv2)
    cpo, trio, d2o = oracle.closest_point_on_surface(q[:24], target.points, target.cells)
    assert np.array_equal(tri[:24], trio) and np.array_equal(d2[:24], d2o) and np.array_equal(cp[:24], cpo)
    io, dvo = oracle.nearest_vertex(q[:24], target.points)
    assert np.array_equal(idx[:24], io) and np.array_equal(dv2[:24], dvo)
    # hinted path == unhinted path: the evaluator/proposal caches must not change results
    cp3, tri3, d23 = ctx.closestPointOnTarget(q)
    assert np.array_equal(tri3, tri) and np.array_equal(d23, d2)


def test_full_size_proposal_properties(pkg, full):
    model, target, ctx = full
    r = model.rank
    theta = make_theta(model, 43, pose=False)
    for direction in ("ModelSampling", "TargetSampling"):
        prop = pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, direction, True)
        post = prop.icpPosterior(theta, with_aux=False)
        assert np.all(np.linalg.eigvalsh(0.5 * (post.M + post.M.T)) >= 1.0 - 1e-9)           # M ≽ I
        assert np.all(post.S > 0) and np.all(np.diff(post.S) <= 0) and post.S[0] <= model.variance.max() * (1 + 1e-9)
        assert np.allclose(post.V.T @ post.V, np.eye(r), atol=1e-10)
        got0 = prop.propose(theta, np.zeros(r))                                                # z = 0 -> step towards alpha
        assert np.abs(got0[10:] - (theta[10:] + 0.1 * (post.alpha - theta[10:]))).max() < 1e-6 * np.abs(post.alpha).max()
        z = np.random.default_rng(1).normal(size=r)
        a, b = prop.propose(theta, z), prop.propose(theta, -z)                                 # linear in z
        assert np.allclose(0.5 * (a + b), got0, rtol=1e-9, atol=1e-12)
        assert prop.logTransitionProbability(theta, a) == prop.logTransitionProbability(theta, a)  # memoised, deterministic
        prop.close()
