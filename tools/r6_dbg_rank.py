import sys, numpy as np
sys.path.insert(0, "/root/repo")
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import oracle as O
for rank in (252, 253, 255, 256):
    model = pkg.data.synthetic_face_model(grid=41, rank=rank)
    target = pkg.data.synthetic_partial_target(model, n_remove=60)
    om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
    r = model.rank
    ctx = pkg.IcpContext(model, target, device=0)
    K = 2 * r
    pp = O.proposal_params(0.1, 6.0, 3.0, O.MODEL_SAMPLING, True, n_model_ids=K)
    prop = pkg.NonRigidIcpProposal(ctx, 0.1, 6.0, 3.0, K, "ModelSampling", True)
    theta = pkg.random_initial_parameters(model, 3)
    try:
        post = prop.icpPosterior(theta)
    except Exception as e:
        print(rank, "EXC", e); continue
    po = O.icp_posterior(om, ot, pp, theta)
    print(rank, "M", np.abs(post.M - po.M).max() / np.abs(po.M).max(), "alpha", np.abs(post.alpha - po.alpha).max() / np.abs(po.alpha).max(),
          "S", np.abs(post.S - po.S).max() / np.abs(po.S).max(), flush=True)
    prop.close(); ctx.close()
