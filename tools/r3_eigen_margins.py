import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tests"))
import numpy as np
import __graft_entry__ as g
pkg = g.load_package()
from oracle import oracle
def face_theta(model, seed):
    rng = np.random.default_rng(seed); th = pkg.initial_parameters(model); th[10:] = 0.5 * rng.normal(size=model.rank); th[1:4] = rng.normal(size=3); th[4:7] = 0.02 * rng.normal(size=3); return th
for rank in (100, 150, 193, 200):
    model = pkg.data.synthetic_face_model(grid=41, rank=rank)
    target = pkg.data.synthetic_partial_target(model, n_remove=90)
    ctx = pkg.IcpContext(model, target, device=0)
    om, ot = oracle.OracleModel.from_model(model), oracle.OracleMesh(target.points, target.cells)
    pp = oracle.proposal_params(0.1, 6.0, 3.0, oracle.MODEL_SAMPLING, True, n_model_ids=2 * rank)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, 2 * rank, "ModelSampling", True)
    for seed in (5, 6):
        theta = face_theta(model, seed)
        post, po = prop.icpPosterior(theta), oracle.icp_posterior(om, ot, pp, theta)
        S = np.sort(po.S); gap = np.min(np.diff(S)) / S[-1]
        print("rank %d seed %d: dS %.2e  orth %.2e  dV %.2e  (smallest relative gap of S %.1e)" % (rank, seed, np.abs(post.S - po.S).max() / np.abs(po.S).max(),
              np.abs(post.V.T @ post.V - np.eye(rank)).max(), np.abs(post.V - po.V).max(), gap), flush=True)
    prop.close(); ctx.close()
