"""CPU: the oracle (oracle/icp_oracle.c) against (1) the committed build-oracle golden vectors and (2) independent
numpy/scipy formulations of the same mathematics (what pins results while Scalismo parity is unpinned)."""
import os

import numpy as np
import pytest
from hypothesis import given, settings, strategies as st
from scipy.spatial import cKDTree

from conftest import ROOT, make_theta

GOLD = np.load(os.path.join(ROOT, "tests", "golden", "oracle_vectors_femur50.npz"))


def scaled_basis(model):
    return model.basis * np.sqrt(model.variance)[None, :]


# ------------------------------------------------------------------ golden vectors (regression guard)

def test_golden_geometry(oracle, femur50, femur50_oracle):
    model, target = femur50
    om, _ = femur50_oracle
    assert np.array_equal(om.instance(GOLD["thetas"][0]), GOLD["instance0"])
    idx, d2 = oracle.nearest_vertex(GOLD["queries"], target.points)
    assert np.array_equal(idx, GOLD["nn_idx"]) and np.array_equal(d2, GOLD["nn_d2"])
    cp, tri, d2 = oracle.closest_point_on_surface(GOLD["queries"], target.points, target.cells)
    assert np.array_equal(tri, GOLD["cp_tri"]) and np.array_equal(cp, GOLD["cp"]) and np.array_equal(d2, GOLD["cp_d2"])


@pytest.mark.parametrize("name", ["model", "target"])
def test_golden_posterior_propose_transition(oracle, femur50, femur50_oracle, name):
    model, _ = femur50
    om, ot = femur50_oracle
    r = model.rank
    pp = (oracle.proposal_params(0.1, 10.0, 5.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r) if name == "model"
          else oracle.proposal_params(0.1, 10.0, 5.0, oracle.TARGET_SAMPLING, True, target_pts=GOLD["target_pts"]))
    i = 1
    theta = GOLD["thetas"][i]
    post = oracle.icp_posterior(om, ot, pp, theta)
    assert np.array_equal(post.corr_id, GOLD[f"{name}_corr_id"][i])
    assert np.allclose(post.alpha, GOLD[f"{name}_alpha"][i], rtol=1e-12, atol=1e-14)
    prop = oracle.propose(om, ot, pp, theta, GOLD["zs"][i])
    assert np.allclose(prop, GOLD[f"{name}_proposed"][i], rtol=1e-11, atol=1e-13)
    assert np.isclose(oracle.log_transition(om, ot, pp, theta, prop), GOLD[f"{name}_logT_fwd"][i], rtol=1e-11)


def test_golden_evaluators(oracle, femur50, femur50_oracle, pkg):
    model, target = femur50
    om, ot = femur50_oracle
    r = model.rank
    tp4 = pkg.data.decimated_point_subset(target, 4 * r)
    theta = GOLD["thetas"][2]
    for mode in (0, 1, 2):
        ep = oracle.evaluator_params(oracle.EVAL_INDEPENDENT, mode, n_model_ids=4 * r, target_pts=tp4, p0=0.0, p1=2.0)
        assert np.isclose(oracle.evaluator_log_value(om, ot, ep, theta)[0], GOLD[f"indep_mode{mode}"][2], rtol=1e-13)
    assert np.isclose(oracle.prior_log_value(r, theta), GOLD["prior"][2], rtol=1e-14)


# ------------------------------------------------------------------ independent formulations

def test_instance_vs_numpy(oracle, femur50, femur50_oracle):
    model, _ = femur50
    om, _ = femur50_oracle
    theta = make_theta(model, 1)
    u = model.ref_points + model.mean_def + (scaled_basis(model) @ theta[10:]).reshape(-1, 3)
    R, ctr = oracle.rotation_matrix(*theta[4:7]), theta[7:10]
    assert np.allclose(R @ R.T, np.eye(3), atol=1e-15) and np.isclose(np.linalg.det(R), 1.0)
    x = theta[0] * ((u - ctr) @ R.T + ctr + theta[1:4])
    assert np.abs(om.instance(theta) - x).max() < 1e-12


def test_rotation_is_zyx_euler(oracle):
    phi, theta, psi = 0.3, -0.2, 0.5
    cz, sz, cy, sy, cx, sx = np.cos(phi), np.sin(phi), np.cos(theta), np.sin(theta), np.cos(psi), np.sin(psi)
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    assert np.allclose(oracle.rotation_matrix(phi, theta, psi), Rz @ Ry @ Rx, atol=1e-15)


def test_nearest_vertex_vs_kdtree(oracle, femur50, femur50_oracle):
    model, target = femur50
    om, _ = femur50_oracle
    x = om.instance(make_theta(model, 2))
    rng = np.random.default_rng(3)
    q = target.points[rng.integers(0, target.n_points, 500)] + rng.normal(size=(500, 3))
    idx, d2 = oracle.nearest_vertex(q, x)
    dd, ii = cKDTree(x).query(q)
    assert np.array_equal(idx, ii) and np.abs(np.sqrt(d2) - dd).max() < 1e-12
    # optimality: no vertex is closer than the reported one
    assert all(((x - q[k]) ** 2).sum(1).min() >= d2[k] - 1e-9 for k in range(0, 500, 50))


def _tri_samples(a, b, c, n=40):
    u, v = np.meshgrid(np.linspace(0, 1, n), np.linspace(0, 1, n))
    m = (u + v) <= 1.0
    u, v = u[m], v[m]
    return a + u[:, None] * (b - a) + v[:, None] * (c - a)


@settings(max_examples=200, deadline=None)
@given(st.lists(st.floats(-10, 10, allow_nan=False), min_size=12, max_size=12))
def test_point_triangle_properties(vals):
    from oracle import oracle as O
    v = np.asarray(vals).reshape(4, 3)
    p, a, b, c = v
    if np.linalg.norm(np.cross(b - a, c - a)) < 1e-6:
        return
    cp, tri, d2 = O.closest_point_on_surface(p[None], np.stack([a, b, c]), np.array([[0, 1, 2]], dtype=np.int32))
    cp = cp[0]
    # feasibility: on the triangle (barycentrics in [0,1]) ...
    T = np.stack([b - a, c - a], axis=1)
    uv, res, *_ = np.linalg.lstsq(T, cp - a, rcond=None)
    assert np.linalg.norm(T @ uv - (cp - a)) < 1e-8
    assert uv.min() > -1e-9 and uv.sum() < 1 + 1e-9
    # ... and optimal: no sampled point of the triangle is closer
    s = _tri_samples(a, b, c)
    assert ((s - p) ** 2).sum(1).min() >= d2[0] - 1e-9
    assert np.isclose(d2[0], ((cp - p) ** 2).sum(), rtol=1e-12, atol=1e-300)


def test_closest_point_on_surface_vs_vertices(oracle, femur50, femur50_oracle):
    model, target = femur50
    om, _ = femur50_oracle
    x = om.instance(make_theta(model, 4))
    cp, tri, d2 = oracle.closest_point_on_surface(x[:200], target.points, target.cells)
    dv, _ = cKDTree(target.points).query(x[:200])
    assert np.all(np.sqrt(d2) <= dv + 1e-12)           # the surface is never farther than the nearest vertex
    # the reported point lies on the reported triangle
    a, b, c = (target.points[target.cells[tri, i]] for i in range(3))
    n = np.cross(b - a, c - a)
    n /= np.linalg.norm(n, axis=1)[:, None]
    assert np.abs(((cp - a) * n).sum(1)).max() < 1e-9


def test_surface_noise_matches_closed_form(oracle):
    rng = np.random.default_rng(5)
    for _ in range(20):
        n = rng.normal(size=3)
        cov = oracle.surface_noise_cov(n, 5.0, 10.0)
        nh = n / np.linalg.norm(n)
        assert np.allclose(cov, 100.0 * np.eye(3) + (25.0 - 100.0) * np.outer(nh, nh), atol=1e-12)
    # the inverted tangent selection (SurfaceNoiseHelpers.scala:46) still gives an orthonormal frame near e_x
    cov = oracle.surface_noise_cov(np.array([1.0, 1e-3, 0.0]), 5.0, 10.0)
    nh = np.array([1.0, 1e-3, 0.0]) / np.linalg.norm([1.0, 1e-3, 0.0])
    assert np.allclose(cov, 100.0 * np.eye(3) - 75.0 * np.outer(nh, nh), atol=1e-10)


def test_sym_eigen_vs_lapack(oracle):
    rng = np.random.default_rng(6)
    a = rng.normal(size=(51, 51))
    a = a @ a.T + np.eye(51)
    w, V = oracle.sym_eigen(a)
    wl = np.linalg.eigvalsh(a)[::-1]
    assert np.allclose(w, wl, rtol=1e-12) and np.all(np.diff(w) <= 0)
    assert np.allclose(V @ np.diag(w) @ V.T, a, atol=1e-10) and np.allclose(V.T @ V, np.eye(51), atol=1e-12)
    assert np.all(V[np.abs(V).argmax(0), np.arange(51)] > 0)  # canonical signs


@pytest.mark.parametrize("direction", [0, 1])
def test_posterior_and_closed_forms_vs_numpy(oracle, femur50, femur50_oracle, pkg, direction):
    """Regression (App. A.4) against numpy, and the r-space closed forms the device uses (App. A.5/A.6 + the
    γ-form of the transition density) against the oracle's long forms."""
    model, target = femur50
    om, ot = femur50_oracle
    r = model.rank
    Q = scaled_basis(model)
    theta = make_theta(model, 7 + direction)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    pp = (oracle.proposal_params(0.1, 10.0, 5.0, 0, True, n_model_ids=2 * r) if direction == 0
          else oracle.proposal_params(0.1, 10.0, 5.0, 1, True, target_pts=tp))
    post = oracle.icp_posterior(om, ot, pp, theta)
    x = om.instance(theta)
    nrm = om.vertex_normals(x)
    R, ctr = oracle.rotation_matrix(*theta[4:7]), theta[7:10]
    y = (post.corr_pt - theta[1:4] - ctr) @ R + ctr - model.ref_points[post.corr_id]
    M, b = np.eye(r), np.zeros(r)
    for k, i in enumerate(post.corr_id):
        W = np.eye(3) / 100.0 + (1 / 25.0 - 1 / 100.0) * np.outer(nrm[i], nrm[i])
        Qi = Q[3 * i:3 * i + 3]
        M += Qi.T @ W @ Qi
        b += Qi.T @ W @ (y[k] - model.mean_def[i])
    assert np.abs(M - post.M).max() < 1e-10 * np.abs(M).max()
    alpha = np.linalg.solve(M, b)
    assert np.abs(alpha - post.alpha).max() < 1e-10 * np.abs(alpha).max()
    D = np.sqrt(model.variance)
    Sig = D[:, None] * np.linalg.inv(M) * D[None, :]
    assert np.allclose(post.V @ np.diag(post.S) @ post.V.T, Sig, atol=1e-9)
    assert np.all(np.linalg.eigvalsh(M) >= 1 - 1e-9)                      # M ≽ I
    assert np.all(post.S <= model.variance.max() + 1e-9)                  # posterior ≼ prior
    if direction == 1:
        dd, ii = cKDTree(x).query(tp)
        assert np.array_equal(post.corr_id, ii)
    # closed-form proposal
    z = np.random.default_rng(9).normal(size=r)
    G, s2 = Q.T @ Q, 1e-5
    cnew = np.linalg.solve(G + s2 * np.eye(r), G @ (post.alpha + (post.V @ (np.sqrt(post.S) * z)) / D))
    want = theta.copy()
    want[10:] = theta[10:] + (cnew - theta[10:]) * 0.1
    got = oracle.propose(om, ot, pp, theta, z)
    assert np.abs(got - want).max() < 1e-10 * np.abs(want[10:]).max()
    # closed-form transition density (valid for any square root of D M^-1 D)
    ct = theta[10:] + (got[10:] - theta[10:]) / 0.1
    gamma = np.linalg.solve(G + s2 * M, G @ (ct - post.alpha))
    lt = -0.5 * gamma @ M @ gamma - 0.5 * r * np.log(2 * np.pi)
    assert np.isclose(lt, oracle.log_transition(om, ot, pp, theta, got), rtol=1e-10)
    moved = got.copy()
    moved[2] += 1.0
    assert oracle.log_transition(om, ot, pp, theta, moved) == -np.inf


def test_evaluators_vs_numpy(oracle, femur50, femur50_oracle, pkg):
    model, target = femur50
    om, ot = femur50_oracle
    r = model.rank
    theta = make_theta(model, 11)
    x = om.instance(theta)
    tp = pkg.data.decimated_point_subset(target, 4 * r)
    _, _, d2a = oracle.closest_point_on_surface(x[:4 * r], target.points, target.cells)
    _, _, d2b = oracle.closest_point_on_surface(tp, x, model.cells)

    def lg(d, mu, s):
        return -((d - mu) / s) ** 2 / 2 - np.log(s * np.sqrt(2 * np.pi))
    m2t, t2m = lg(np.sqrt(d2a), 0, 2.0).sum(), lg(np.sqrt(d2b), 0, 2.0).sum()
    for mode, want in ((0, m2t), (1, t2m), (2, 0.5 * m2t + 0.5 * t2m)):
        ep = oracle.evaluator_params(oracle.EVAL_INDEPENDENT, mode, n_model_ids=4 * r, target_pts=tp, p0=0.0, p1=2.0)
        assert np.isclose(oracle.evaluator_log_value(om, ot, ep, theta)[0], want, rtol=1e-12)
    _, _, d2all = oracle.closest_point_on_surface(x, target.points, target.cells)
    _, _, d2rev = oracle.closest_point_on_surface(target.points, x, model.cells)
    hd = np.sqrt(max(d2all.max(), d2rev.max()))
    ep = oracle.evaluator_params(oracle.EVAL_HAUSDORFF, 2, p0=0.7)
    assert np.isclose(oracle.evaluator_log_value(om, ot, ep, theta)[0], np.log(0.7) - 0.7 * hd, rtol=1e-12)
    assert np.isclose(oracle.prior_log_value(r, theta), -0.5 * theta[10:] @ theta[10:] - 0.5 * r * np.log(2 * np.pi))


def test_boundary_flags(oracle, femur50, pkg):
    from conftest import open_patch_target
    model, target = femur50
    assert oracle.OracleMesh(target.points, target.cells).boundary().sum() == 0  # closed femur
    pts, cells = open_patch_target(target)
    b = oracle.OracleMesh(pts, cells).boundary()
    assert b.sum() > 0 and np.array_equal(b, pkg.data.boundary_vertex_flags(pkg.data.TriangleMesh(pts, cells)))


def test_rng_stream(oracle):
    """the counter-based generator shared with the C++ host harness: restated here in Python"""
    M64 = (1 << 64) - 1

    def sm(x):
        x = (x + 0x9E3779B97F4A7C15) & M64
        x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & M64
        x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & M64
        return x ^ (x >> 31)

    def uni(seed, step, lane):
        h = sm(sm(sm(seed) ^ ((step * 0xD1342543DE82EF95) & M64)) ^ ((lane * 0x2545F4914F6CDD1D) & M64))
        return ((h >> 11) + 0.5) / 9007199254740992.0
    for seed, step, lane in ((1024, 0, 0), (1024, 17, 2), (7, 123456, 1001)):
        assert oracle.lib().orc_rng_uniform(seed, step, lane) == uni(seed, step, lane)
    u = np.array([oracle.lib().orc_rng_uniform(1, s, 0) for s in range(4000)])
    z = np.array([oracle.lib().orc_rng_normal(1, s, 3) for s in range(4000)])
    assert abs(u.mean() - 0.5) < 0.02 and abs(z.mean()) < 0.06 and abs(z.std() - 1) < 0.05


def test_golden_chain_prefix(oracle, femur50, femur50_oracle, pkg):
    """the first steps of the committed 200-step femur-50 chain (seed 1024) are reproduced by the oracle"""
    from test_gpu_chain import oracle_chain_config
    model, target = femur50
    om, ot = femur50_oracle
    setup = pkg.femur_icp_proposal_registration(model, target)
    acc, comp, logp, states = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), pkg.initial_parameters(model), 1024, 25)
    assert np.array_equal(acc, GOLD["chain_accepted"][:25]) and np.array_equal(comp, GOLD["chain_component"][:25])
    assert np.allclose(states, GOLD["chain_states"][:25], rtol=1e-9, atol=1e-11)


def test_search_backends_agree(oracle, pkg):
    """the CPU baselines B1 (KD-tree + bounding-volume hierarchy) and B2 (OpenMP scans) are the restatement
    with another search back end (oracle/icp_spatial.c).  They must return what the single-thread scans return, bit for bit:
    indices (ties to the lowest index), squared distances and closest points — on the femur meshes, on a subdivided target
    with exactly coplanar/duplicated distances, and on queries that sit ON vertices and edges (ties by construction)."""
    model, target = pkg.data.load_femur_model_and_target(50)
    big = pkg.data.subdivide(target, 3)  # no jitter: shared edges and coplanar children give exact ties
    rng = np.random.default_rng(3)
    q = np.concatenate([model.ref_points[:200] + rng.normal(0, 2.0, size=(200, 3)),
                        big.points[:100],                                              # on vertices: several triangles at distance 0
                        0.5 * (big.points[big.cells[:50, 0]] + big.points[big.cells[:50, 1]]),  # on edges
                        rng.normal(0, 300.0, size=(20, 3))])                            # far away
    try:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
        idx0, d0 = oracle.nearest_vertex(q, big.points)
        cp0, tri0, s0 = oracle.closest_point_on_surface(q, big.points, big.cells)
        for backend in (oracle.SEARCH_TREES, oracle.SEARCH_BRUTE_OMP):
            oracle.set_search_backend(backend, 4)
            idx, d = oracle.nearest_vertex(q, big.points)
            cp, tri, s = oracle.closest_point_on_surface(q, big.points, big.cells)
            assert np.array_equal(idx, idx0) and np.array_equal(d, d0)
            assert np.array_equal(tri, tri0) and np.array_equal(s, s0) and np.array_equal(cp, cp0)
        # a changed mesh at the same address is a new mesh (B1 rebuilds its structures per state)
        oracle.set_search_backend(oracle.SEARCH_TREES)
        pts = np.ascontiguousarray(model.ref_points.copy())
        kd0, bvh0 = oracle.search_stats()
        a, _ = oracle.nearest_vertex(q[:50], pts)
        pts += 1.5
        b, _ = oracle.nearest_vertex(q[:50], pts)
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
        b0, _ = oracle.nearest_vertex(q[:50], pts)
        assert np.array_equal(b, b0)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)


def test_chain_identical_under_every_search_backend(oracle, pkg):
    """The whole chain (accept/reject sequence, components, states) is bit-identical under B1 and B2."""
    model, target = pkg.data.load_femur_model_and_target(50)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    setup = pkg.femur_icp_proposal_registration(model, target)
    icp = [oracle.proposal_params(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], p.get("boundary_aware", True),
                                  n_model_ids=p.get("n_model_ids", 0), target_pts=p.get("target_pts")) for p in setup.icp]
    e = setup.eval
    ep = oracle.evaluator_params(e["kind"], 2, n_model_ids=e["n_model_ids"], target_pts=e["target_pts"], p0=e["gauss_mean"], p1=e["gauss_sigma"],
                                 p2=e["exp_rate"])  # symmetric: the current model surface is searched, too
    cfg = oracle.chain_config(icp, [0.5, 0.5], setup.w_icp, setup.w_rw, setup.rw_sigma, ep)
    theta0 = pkg.initial_parameters(model)
    try:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
        ref = oracle.run_chain(om, ot, cfg, theta0, 7, 6)
        for backend in (oracle.SEARCH_TREES, oracle.SEARCH_BRUTE_OMP):
            oracle.set_search_backend(backend, 4)
            got = oracle.run_chain(om, ot, cfg, theta0, 7, 6)
            for a, b in zip(ref, got):
                assert np.array_equal(a, b)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
