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
    return oracle.chain_config(icp, [p.get("weight", 0.5) for p in setup.icp], setup.w_icp, setup.w_rw, setup.rw_sigma, ep)


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
