"""CPU: the pose random walks (api/sampling/proposals/PoseProposals.scala:31-90) and their mixture
(api/sampling/MixedProposalDistributions.scala:29-39) in the C++ harness, against the oracle and against hand-computed values."""
import numpy as np
import pytest

ROT = (0.5, 0.02, 0.003)      # rotYaw, rotPitch, rotRoll  — unequal on purpose
TRANS = (0.7, 0.11, 0.013)    # transX, transY, transZ


def gauss_logpdf(x, sd):
    return -0.5 * (x / sd) ** 2 - (np.log(np.sqrt(2 * np.pi)) + np.log(sd))


def lse_mix(ts, ws):
    ts, ws = np.asarray(ts, float), np.asarray(ws, float)
    if np.all(np.isneginf(ts)):
        return -np.inf
    mx = ts.max()
    return np.log(np.sum(ws * np.exp(ts - mx))) + mx


def theta0(r=7, seed=0):
    rng = np.random.default_rng(seed)
    t = np.zeros(10 + r)
    t[0] = 1.0
    t[1:7] = rng.normal(size=6) * 0.1
    t[7:10] = (1.0, 2.0, 3.0)
    t[10:] = rng.normal(size=r)
    return t


def test_component_transition_follows_the_reference_reset_semantics(oracle):
    """PoseProposals.scala:47-49 resets the whole rotation triple of `to` before comparing (:78-80: the whole translation): -inf only
    for differences OUTSIDE the proposal's group; inside the group only its own axis enters the residual."""
    a = theta0()
    b = a.copy(); b[4] += 0.01                      # a Roll move (rotation._1)
    sd = 0.02
    # component order Yaw, Pitch, Roll, X, Y, Z
    assert oracle.pose_log_transition(2, sd, a, b) == pytest.approx(gauss_logpdf(0.01, sd), rel=1e-15)      # Roll: its own residual
    assert oracle.pose_log_transition(0, sd, a, b) == pytest.approx(gauss_logpdf(0.0, sd), rel=1e-15)       # Yaw on a Roll move: logPdf(0)
    assert oracle.pose_log_transition(1, sd, a, b) == pytest.approx(gauss_logpdf(0.0, sd), rel=1e-15)
    for comp in (3, 4, 5):                                                                               # translations: rotation differs
        assert oracle.pose_log_transition(comp, sd, a, b) == -np.inf
    c = a.copy(); c[2] -= 0.3                       # a TranslationY move
    assert oracle.pose_log_transition(4, sd, a, c) == pytest.approx(gauss_logpdf(-0.3, sd), rel=1e-15)
    assert oracle.pose_log_transition(3, sd, a, c) == pytest.approx(gauss_logpdf(0.0, sd), rel=1e-15)
    assert all(oracle.pose_log_transition(k, sd, a, c) == -np.inf for k in (0, 1, 2))
    d = a.copy(); d[12] += 1e-3                     # a shape move: outside both groups
    assert all(oracle.pose_log_transition(k, sd, a, d) == -np.inf for k in range(6))
    e = a.copy(); e[8] += 1e-3                      # the rotation centre belongs to neither group
    assert all(oracle.pose_log_transition(k, sd, a, e) == -np.inf for k in range(6))


def test_pose_mixture_transition_is_the_hand_computed_log_sum_exp(pkg, oracle):
    a = theta0(seed=3)
    cases = []
    b = a.copy(); b[4] += 0.004; cases.append(("roll", b, [gauss_logpdf(0, ROT[0]), gauss_logpdf(0, ROT[1]), gauss_logpdf(0.004, ROT[2])] + [-np.inf] * 3))
    b = a.copy(); b[6] -= 0.3; cases.append(("yaw", b, [gauss_logpdf(-0.3, ROT[0]), gauss_logpdf(0, ROT[1]), gauss_logpdf(0, ROT[2])] + [-np.inf] * 3))
    b = a.copy(); b[5] += 0.01; b[6] += 0.2; cases.append(("pitch+yaw", b, [gauss_logpdf(0.2, ROT[0]), gauss_logpdf(0.01, ROT[1]), gauss_logpdf(0, ROT[2])] + [-np.inf] * 3))
    b = a.copy(); b[3] += 0.02; cases.append(("z", b, [-np.inf] * 3 + [gauss_logpdf(0, TRANS[0]), gauss_logpdf(0, TRANS[1]), gauss_logpdf(0.02, TRANS[2])]))
    b = a.copy(); cases.append(("same", b, [gauss_logpdf(0, s) for s in ROT + TRANS]))
    b = a.copy(); b[1] += 0.1; b[5] += 0.01; cases.append(("x+pitch", b, [-np.inf] * 6))
    b = a.copy(); b[11] += 0.1; cases.append(("shape", b, [-np.inf] * 6))
    for name, b, ts in cases:
        for frm, to in ((a, b), (b, a)):
            if frm is b:   # the backward direction: residuals change sign, the Gaussians are symmetric
                pass
            want = lse_mix(ts, [1 / 6] * 6)
            got_o = oracle.pose_mixture_log_transition(ROT, TRANS, frm, to)
            got_h = pkg.sampling.pose_mixture_log_transition(ROT, TRANS, frm, to)
            if np.isneginf(want):
                assert got_o == -np.inf and got_h == -np.inf, name
            else:
                assert abs(got_o - want) <= 1e-14 * abs(want) and got_h == got_o, (name, got_o, got_h, want)


def test_pose_names_sigmas_and_parameters_belong_together(pkg, oracle):
    """mixedRandomPoseProposal hands rotYaw to YawAxis (MixedProposalDistributions.scala:31), and YawAxis perturbs rotation._3
    (PoseProposals.scala:41) = allParameters[6]; with unequal sigmas a swapped pairing shows."""
    a = theta0(seed=5)
    S = pkg.ChainSetup.scala_double
    names = ("RotationYaw", "RotationPitch", "RotationRoll", "TranslationX", "TranslationY", "TranslationZ")
    sig = ROT + TRANS
    seen = set()
    for step in range(200):
        out, leaf, name = pkg.sampling.pose_mixture_propose(ROT, TRANS, a, 99, step)
        k = leaf - 3
        seen.add(k)
        assert name == "%s-%s" % (names[k], S(sig[k]))
        idx = pkg.sampling.POSE_LEAF_PARAMETER[leaf]
        assert idx == oracle.POSE_PARAM_INDEX[k]
        changed = np.flatnonzero(out != a)
        assert list(changed) == [idx]
        assert out[idx] == a[idx] + sig[k] * oracle.lib().orc_rng_normal(99, step, 0)
        # the component is the one the mixture draw on lane 1 selects (six equal weights)
        u = oracle.lib().orc_rng_uniform(99, step, 1)
        acc, pick = 0.0, 5
        for i in range(6):
            acc += 0.5 / 3.0
            if acc >= u:
                pick = i
                break
        assert pick == k
    assert seen == set(range(6))
    setup = pkg.ChainSetup()
    setup.pose_rot_sigma, setup.pose_trans_sigma = ROT, TRANS
    ln = setup.leaf_names()
    assert [ln[3 + k] for k in range(6)] == ["%s-%s" % (names[k], S(sig[k])) for k in range(6)]


def test_scala_double_python_and_native_agree(pkg):
    """magnitudes in [1e-4, 1e-3) used to come out as '5.0000000000000001E-4' from the Python side."""
    S, N = pkg.ChainSetup.scala_double, pkg.sampling.scala_double_native
    assert S(0.0005) == "5.0E-4" and S(0.00015) == "1.5E-4" and S(1e-4) == "1.0E-4" and S(123456789.0) == "1.23456789E8"
    assert S(0.001) == "0.001" and S(9999999.0) == "9999999.0" and S(1e7) == "1.0E7" and S(-0.25) == "-0.25" and S(100.0) == "100.0"
    rng = np.random.default_rng(0)
    grid = [5e-4, 1.5e-4, 1e-3, 9.99e-4, 0.1, 0.01, 0.3, 2.0, 6.0, 1e-5, 3.3e-7, 1e7, 1.5e8, 12345.678, 1 / 3, 2 / 3e4]
    grid += list(10.0 ** rng.uniform(-8, 9, size=300) * rng.choice([1, -1], size=300))
    grid += [round(x, 3) for x in 10.0 ** rng.uniform(-5, 3, size=200)]
    for x in grid:
        assert S(x) == N(x), (x, S(x), N(x))
        assert float(S(x).replace("E", "e")) == x


def test_oracle_chain_with_pose_walks(pkg, oracle):
    """orc_run_chain with the (pose, ICP, shape walk) mixture of apps/bfm/BfmFittingPartial.scala:70 on a small open-target face:
    all three kinds are drawn, a pose step changes exactly its own parameter, and the reported log value is the state's."""
    model = pkg.data.synthetic_face_model(grid=21, rank=12)
    target = pkg.data.synthetic_partial_target(model, n_remove=30)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    r = model.rank
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * r)
    tp = pkg.data.decimated_point_subset(target, 4 * r)
    ep = oracle.evaluator_params(oracle.EVAL_COLLECTIVE, 2, n_model_ids=4 * r, target_pts=tp, p0=0.1, p1=0.3, p2=1.0)
    cfg = oracle.chain_config([pp], [0.5], 0.55, 0.05, 0.1, ep, w_pose=0.4, pose_rot_sigma=ROT[::-1], pose_trans_sigma=TRANS)
    th0 = oracle.initial_theta(model.ref_points, r)
    n = 60
    acc, comp, logp, states = oracle.run_chain(om, ot, cfg, th0, 4, n)
    assert set(comp) >= {0, 2} and len(set(comp) & {3, 4, 5, 6, 7, 8}) >= 3 and 0 < acc.sum() < n
    prev = th0
    for s in range(n):
        if acc[s] and comp[s] >= 3:
            assert list(np.flatnonzero(states[s] != prev)) == [oracle.POSE_PARAM_INDEX[comp[s] - 3]]
        if not acc[s]:
            assert np.array_equal(states[s], prev)
        prev = states[s]
    lv, rc = oracle.evaluator_log_value(om, ot, ep, states[-1])
    assert rc == 0 and abs(logp[-1] - (lv + oracle.prior_log_value(r, states[-1]))) <= 1e-12 * abs(logp[-1])
    # the mixture's density of an accepted pose move, from the whole-chain entry point, equals w_pose x the pose mixture's
    k = next(s for s in range(1, n) if acc[s] and comp[s] >= 3)
    got = oracle.chain_log_transition(om, ot, cfg, states[k - 1], states[k])
    want = np.log(0.4 / (0.4 + 0.55 + 0.05)) + oracle.pose_mixture_log_transition(ROT[::-1], TRANS, states[k - 1], states[k])
    assert abs(got - want) <= 1e-13 * abs(want)
