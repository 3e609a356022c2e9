#!/usr/bin/env python3
"""Generate tests/golden/oracle_vectors_femur50.npz — BUILD-ORACLE golden vectors (NOT Scalismo outputs).

The reference cannot be run in this image (Scala/Scalismo, no JVM) and ships no golden files, so parity is
unpinned.  These vectors freeze the outputs of oracle/icp_oracle.c on the reference's bundled femur
data for seeded inputs; they guard the oracle against regressions and give the GPU tests a fixed target.
Run from the repo root:  python tests/golden/make_golden_vectors.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft  # noqa: E402
from conftest import make_theta  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    pkg = graft.load_package()
    model, target = pkg.data.load_femur_model_and_target(50)
    om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
    r = model.rank
    out = {}
    thetas = np.stack([make_theta(model, 900 + i) for i in range(3)])
    out["thetas"] = thetas
    out["instance0"] = om.instance(thetas[0])
    rng = np.random.default_rng(77)
    q = model.ref_points[rng.integers(0, model.n_points, 64)] + rng.normal(size=(64, 3)) * 3.0
    out["queries"] = q
    out["nn_idx"], out["nn_d2"] = O.nearest_vertex(q, target.points)
    out["cp"], out["cp_tri"], out["cp_d2"] = O.closest_point_on_surface(q, target.points, target.cells)
    tp = pkg.data.decimated_point_subset(target, 2 * r)
    out["target_pts"] = tp
    zs = rng.normal(size=(3, r))
    out["zs"] = zs
    for name, pp in (("model", O.proposal_params(0.1, 10.0, 5.0, O.MODEL_SAMPLING, True, n_model_ids=2 * r)),
                     ("target", O.proposal_params(0.1, 10.0, 5.0, O.TARGET_SAMPLING, True, target_pts=tp))):
        ids, alphas, Ss, props, lts, ltb = [], [], [], [], [], []
        for i in range(3):
            post = O.icp_posterior(om, ot, pp, thetas[i])
            ids.append(post.corr_id.copy()); alphas.append(post.alpha.copy()); Ss.append(post.S.copy())
            prop = O.propose(om, ot, pp, thetas[i], zs[i])
            props.append(prop)
            lts.append(O.log_transition(om, ot, pp, thetas[i], prop))
            ltb.append(O.log_transition(om, ot, pp, prop, thetas[i]))
        out[f"{name}_corr_id"] = np.stack(ids); out[f"{name}_alpha"] = np.stack(alphas); out[f"{name}_S"] = np.stack(Ss)
        out[f"{name}_proposed"] = np.stack(props); out[f"{name}_logT_fwd"] = np.asarray(lts); out[f"{name}_logT_bwd"] = np.asarray(ltb)
    tp4 = pkg.data.decimated_point_subset(target, 4 * r)
    for mode in (0, 1, 2):
        ep = O.evaluator_params(O.EVAL_INDEPENDENT, mode, n_model_ids=4 * r, target_pts=tp4, p0=0.0, p1=2.0)
        out[f"indep_mode{mode}"] = np.asarray([O.evaluator_log_value(om, ot, ep, t)[0] for t in thetas])
    ep = O.evaluator_params(O.EVAL_HAUSDORFF, 2, p0=1.0)
    out["hausdorff"] = np.asarray([O.evaluator_log_value(om, ot, ep, t)[0] for t in thetas])
    out["prior"] = np.asarray([O.prior_log_value(r, t) for t in thetas])
    # a16: 200-step femur-50 chain, configuration of apps/femur/IcpProposalRegistration.scala:59-85, seed 1024
    from test_gpu_chain import oracle_chain_config
    setup = pkg.femur_icp_proposal_registration(model, target)
    acc, comp, logp, states = O.run_chain(om, ot, oracle_chain_config(O, setup), pkg.initial_parameters(model), 1024, 200)
    out["chain_accepted"], out["chain_component"], out["chain_logp"], out["chain_states"] = acc, comp, logp, states
    path = os.path.join(ROOT, "tests", "golden", "oracle_vectors_femur50.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes; chain acceptance", acc.mean())


if __name__ == "__main__":
    main()
