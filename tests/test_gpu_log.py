"""GPU: SURVEY.md §8f row 2 end to end on REAL records — a chain with pose moves -> the records' gather -> the reference's JSON log ->
reload -> best sample -> LogHelper.samplesFromLog -> posterior variability, every stage against the oracle on the same states."""
import json

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_chain_to_json_log_to_variability(pkg, oracle, tmp_path):
    from test_gpu_chain import oracle_chain_config
    model = pkg.data.synthetic_face_model(grid=41, rank=40)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    r = model.rank
    setup = pkg.bfm_fitting_partial(model, target, evaluator="collective")       # apps/bfm/BfmFittingPartial.scala:62-83
    setup.pose_rot_sigma, setup.pose_trans_sigma = (0.02, 0.01, 0.004), (0.2, 0.1, 0.05)
    theta0, seed, n = pkg.initial_parameters(model), 31, 160
    ctx = pkg.IcpContext(model, target, device=0)
    chain = pkg.SamplingRegistration(ctx, setup, theta0, seed)
    rec = chain.run(n)
    # ---- the gather's record block (one rank: the same array with a leading rank axis), then the reference's log format
    allrec = pkg.sharding.gather_records(rec)
    assert allrec.shape == (1, n, 14 + r) and np.array_equal(allrec[0], rec)
    names = setup.leaf_names()
    path = tmp_path / "chain.json"
    lg = pkg.loggers.JSONAcceptRejectLogger(str(path)).add_records(allrec[0], names)
    lg.write_log()                                                               # JSONAcceptRejectLogger.scala:112-120
    raw = json.load(open(path))
    assert len(raw) == n and [e["index"] for e in raw] == list(range(n))
    # ---- against the oracle's chain: same decisions; accepted entries carry the oracle's states, rejected ones nothing
    try:
        oracle.set_search_backend(oracle.SEARCH_TREES)
        acc_o, comp_o, logp_o, states_o = oracle.run_chain(om, ot, oracle_chain_config(oracle, setup), theta0, seed, n)
    finally:
        oracle.set_search_backend(oracle.SEARCH_BRUTE)
    assert [e["status"] for e in raw] == [bool(a) for a in acc_o]
    assert [e["name"] for e in raw] == [names[int(c)] for c in comp_o]
    scale = np.abs(states_o[:, 10:]).max()
    prev = theta0
    for e, a, st in zip(raw, acc_o, states_o):
        if a:
            th = pkg.loggers.JSONAcceptRejectLogger.sample_to_model_parameters(e)                       # :133-140
            assert np.abs(th[10:] - st[10:]).max() <= 1e-5 * scale and np.abs(th[1:10] - st[1:10]).max() <= 1e-12
            if e["name"].startswith(("Rotation", "Translation")):
                # the NAME, the sigma inside it and the parameter that moved belong together (PoseProposals.scala:39-41)
                leaf = next(k for k, v in names.items() if v == e["name"])
                assert list(np.flatnonzero(th != prev)) == [pkg.sampling.POSE_LEAF_PARAMETER[leaf]]
            prev = th
        else:
            assert e["rigid"] == [] and e["coeff"] == []                                               # :104
    assert abs(lg.percent_accepted - acc_o.mean()) < 1e-12
    used = {e["name"] for e in raw if e["status"]}
    assert any(nm.startswith("Rotation") for nm in used) and any(nm.startswith("Translation") for nm in used) and names[0] in used
    # ---- best sample (getBestFittingParsFromJSON :142-146) = the accepted step with the largest product value
    best = lg.get_best_fitting_pars_from_json()
    k_best = int(np.argmax(np.where(acc_o != 0, logp_o, -np.inf)))
    assert np.abs(best[1:] - states_o[k_best, 1:]).max() <= 1e-5 * scale
    # ---- LogHelper.samplesFromLog (apps/util/LogHelper.scala:27-38) -> PosteriorVariability (apps/util/PosteriorVariability.scala:30-73)
    first_acc = int(np.flatnonzero(acc_o)[0])
    sub = pkg.loggers.samples_from_log(raw, take_every_n=10, total=n, burn_in=max(first_acc, 20))
    assert len(sub) >= 10 and all(s["status"] for s, _ in sub)
    thetas = np.stack([pkg.loggers.JSONAcceptRejectLogger.sample_to_model_parameters(s) for s, _ in sub])
    assert len({tuple(t) for t in thetas}) >= 5
    for mode in (0, 1, 2):
        got = pkg.posterior_variability(ctx, thetas, mode=mode, theta_ref=best)
        want = oracle.posterior_variability(om, thetas, mode=mode, theta_ref=best)
        assert np.abs(got - want).max() <= 1e-12 * max(np.abs(want).max(), 1e-30) + 1e-13, mode
    chain.close(); ctx.close()
