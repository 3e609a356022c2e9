"""GPU: the drop-in path as Scalismo drives it — the plug-in methods called ONE BY ONE (MetropolisHastings.next, SURVEY App. B1;
api/sampling/SamplingRegistration.scala:52-58; the mixture's log-sum-exp over every leaf, MixedProposalDistributions.scala:48-68) —
over a chain bound once with icp_chain_bind: the first call of a step submits the whole step, the calls behind it find their values on
the host.  Compared with the unbound per-method calls (a second context), with icp_chain_step, and — whole chains through the C++
harness, fused = 3 — with the oracle's chain decision for decision."""
import numpy as np
import pytest

from test_gpu_chain import oracle_chain_config

pytestmark = pytest.mark.gpu


def _femur_parts(pkg, ctx, model, target):
    r = model.rank
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    props = [pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "ModelSampling", True),
             pkg.NonRigidIcpProposal(ctx, 0.1, 10.0, 5.0, 2 * r, "TargetSampling", True, decimatedTargetPoints=tp)]
    ev = pkg.IndependentPointDistanceEvaluator(ctx, 0.0, 2.0, pkg.ModelToTargetEvaluation, 4 * r,
                                               decimatedTargetPoints=pkg.data.decimated_point_subset(target, 4 * r))
    return props, ev


def _mh_calls(props, ev, cur, prop_from):
    """The per-method calls of one MetropolisHastings.next in Scalismo's order; prop_from(cur) makes the proposal."""
    out = {"cur_value": ev.logValue(cur)}
    prop = prop_from(cur)
    out["prop"] = prop
    out["prop_value"] = ev.logValue(prop)
    out["fwd"] = [p.logTransitionProbability(cur, prop) for p in props]
    out["bwd"] = [p.logTransitionProbability(prop, cur) for p in props]
    return out


def test_bound_per_method_calls_equal_the_unbound_ones(pkg, femur50):
    """ICP steps from either direction, a shape random walk made on the host and a pose walk, accepted and rejected alike: every
    number a bound chain hands out is the number the unbound entry points compute (ranks <= 64: the same bits), one whole step per
    MH step, four parked densities behind it."""
    model, target = femur50
    r = model.rank
    ca, cb = pkg.IcpContext(model, target, device=0), pkg.IcpContext(model, target, device=0)
    pa, ea = _femur_parts(pkg, ca, model, target)
    pb, eb = _femur_parts(pkg, cb, model, target)
    ea.bindChain(pa)
    rng = np.random.default_rng(11)
    cur = pkg.initial_parameters(model)
    cur[10:] = 0.3 * rng.normal(size=r)
    n_icp = n_host = 0
    for step in range(12):
        kind = step % 4
        z = rng.normal(size=r)
        if kind in (0, 1):   # an ICP proposal generates
            make = lambda ps: (lambda c: ps[kind].propose(c, z))
            n_icp += 1
        elif kind == 2:      # RandomShapeUpdateProposal (RandomShapeUpdateProposal.scala:31-35): host arithmetic
            make = lambda ps: (lambda c: np.concatenate([c[:10], c[10:] + 0.1 * z]))
            n_host += 1
        else:                # a pose walk (PoseProposals.scala:39-41): every ICP density is -inf (NonRigidIcpProposal.scala:72-74)
            def make(ps):
                def f(c):
                    o = c.copy()
                    o[4] += 0.01 * z[0]
                    return o
                return f
            n_host += 1
        a = _mh_calls(pa, ea, cur, make(pa))
        # (a KL basis comes from a warm-started Jacobi iteration: at rounding level it depends on which decompositions ran before, so
        # the two contexts' samples agree to 1e-10, not to the bit — tests/test_gpu_parity.py::test_chain_step_matches_separate_calls;
        # everything behind the proposal is compared AT THE SAME STATE and has no decomposition behind it)
        assert np.allclose(make(pb)(cur), a["prop"], rtol=1e-10, atol=1e-11)
        b = _mh_calls(pb, eb, cur, lambda c: a["prop"])
        assert a["cur_value"] == b["cur_value"] and a["prop_value"] == b["prop_value"]
        assert a["fwd"] == b["fwd"] and a["bwd"] == b["bwd"], (step, a["fwd"], b["fwd"], a["bwd"], b["bwd"])
        if kind == 3:
            assert all(v == -np.inf for v in a["fwd"] + a["bwd"])
        if step % 3 != 2:  # "accept" two steps in three
            cur = a["prop"]
    st = ea.bindStats()
    assert st["steps_from_propose"] == n_icp and st["steps_from_log_value"] == n_host, st
    assert st["parked_transition_hits"] == 4 * (n_icp + n_host - 3), st  # (the pose walks' -inf never reach the parked values)
    paths = ca.step_paths()
    assert paths["merged"] + paths["wide"] + paths["per_stage"] == n_icp + n_host, paths
    # correspondence ids of a bound propose: the per-method path behind the bound step (same posterior, same z)
    z = rng.normal(size=r)
    got, ids = pa[0].propose(cur, z, return_correspondences=True)
    want, ids_b = pb[0].propose(cur, z, return_correspondences=True)
    assert np.allclose(got, want, rtol=1e-10, atol=1e-11) and np.array_equal(ids, ids_b)
    # destroying a member dissolves the binding; the evaluator then answers as an unbound one
    pa[1].close()
    other = cur.copy()
    other[10:] += 0.05
    assert ea.logValue(other) == eb.logValue(other)
    assert ea.bindStats()["steps_from_log_value"] == 0  # (a new, empty binding record)
    for o in pa[:1] + pb + [ea, eb]:
        o.close()
    ca.close()
    cb.close()


@pytest.mark.parametrize("config", ["femur50", "face"])
def test_chain_driven_method_by_method_matches_the_whole_step_chain(pkg, femur50, config):
    """The C++ harness with fused = 3 (every call of MetropolisHastings.next handed to the native side, chain bound once) against
    fused = 2 (icp_chain_step): the same records — on the merged step (femur-50) and on the wide step (open face target, pose walks,
    collective evaluator)."""
    if config == "femur50":
        model, target = femur50
        make = lambda f: pkg.femur_icp_proposal_registration(model, target, fused=f)
        theta0, n_steps = pkg.initial_parameters(model), 60
    else:
        model = pkg.data.synthetic_face_model(grid=41, rank=40)
        target = pkg.data.synthetic_partial_target(model, n_remove=90)
        make = lambda f: pkg.bfm_fitting_partial(model, target, evaluator="collective", fused=f)
        theta0, n_steps = pkg.random_initial_parameters(model, chain_index=1), 40
    recs, calls = {}, {}
    for fused in (2, 3):
        ctx = pkg.IcpContext(model, target, device=0)
        chain = pkg.SamplingRegistration(ctx, make(fused), theta0, 1024)
        recs[fused] = chain.run(n_steps)
        calls[fused] = chain.native_calls()
        chain.close()
        ctx.close()
    assert np.array_equal(recs[2][:, 1], recs[3][:, 1]) and np.array_equal(recs[2][:, 2], recs[3][:, 2])
    assert recs[2][:, 1].sum() >= 3
    if config == "femur50":
        assert np.array_equal(recs[2], recs[3])
    else:
        # (the wide step at ranks <= 64 decomposes ahead with a warm-started Jacobi iteration whose starting basis depends on which
        # decompositions happened to be complete — tests/test_gpu_wide_loop.py compares such chains the same way: decisions exactly,
        # states to rounding)
        assert np.allclose(recs[2][:, 3:], recs[3][:, 3:], rtol=1e-9, atol=1e-10)
    c = calls[3]
    assert c["bound_steps_from_propose"] + c["bound_steps_from_log_value"] == n_steps, c
    # per step: logValue(current) + logValue(proposal) at the boundary, propose (ICP steps) + 4 densities (shape moves)
    assert c["log_value_calls"] == 2 * n_steps + 1, c


def test_method_by_method_chain_matches_oracle(pkg, femur50, femur50_oracle, oracle):
    model, target = femur50
    om, ot = femur50_oracle
    n_steps, seed = 60, 77
    setup = pkg.femur_icp_proposal_registration(model, target, fused=3)
    theta0 = pkg.initial_parameters(model)
    acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n_steps)
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n_steps)
    assert np.array_equal(rec[:, 1].astype(np.uint8), acc_o), "accept/reject sequences differ"
    assert np.array_equal(rec[:, 2].astype(np.int32), comp_o), "mixture components differ"
    assert np.abs(rec[:, 4 + 10:] - states_o[:, 10:]).max() <= 1e-5 * np.abs(states_o[:, 10:]).max()
    assert np.abs(rec[:, 3] - logp_o).max() <= 1e-6 * np.abs(logp_o).max()
    chain.close()
    ctx.close()
