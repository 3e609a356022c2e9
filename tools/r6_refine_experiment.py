#!/usr/bin/env python3
"""CPU experiment (numpy + the oracle; no GPU): can the KL basis of a PROPOSED state's posterior be had by iterative refinement
(Ogita-Aishima: the step k_tri_correction already takes once behind the back-transformation) from the CURRENT state's basis, instead
of a fresh tridiagonal reduction?  For a chain of the given model: pairs (theta, theta' = propose(theta)), N = D^-1 M D^-1 of both,
refinement of N(theta') from V(theta): off-diagonal residual and eigenvector error per iteration.
usage: r6_refine_experiment.py femur200|face200 [steps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as graft
pkg = graft.load_package(require_native=False) if "require_native" in graft.load_package.__code__.co_varnames else graft.load_package()
from oracle import oracle as O
O.lib()

which = sys.argv[1] if len(sys.argv) > 1 else "femur200"
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 30
if which == "femur200":
    model, target = pkg.data.load_femur_model_and_target(200)
    sd, lo, hi = 0.1, 10.0, 5.0
else:
    model = pkg.data.synthetic_face_model(grid=41, rank=200)
    target = pkg.data.synthetic_partial_target(model, n_remove=60)
    sd, lo, hi = 0.1, 6.0, 3.0
om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
r = model.rank
K = min(2 * r, model.n_points)
pp = O.proposal_params(sd, lo, hi, O.MODEL_SAMPLING, True, n_model_ids=K)
D = np.sqrt(model.variance)


def nmat(theta):
    po = O.icp_posterior(om, ot, pp, theta)
    M = 0.5 * (po.M + po.M.T)
    return M / D[:, None] / D[None, :]


def refine(N, V, iters=6):
    out = []
    for _ in range(iters):
        R = np.eye(r) - V.T @ V
        S = V.T @ (N @ V)
        d = np.diag(S) / (1.0 - np.diag(R))
        den = d[None, :] - d[:, None]
        big = np.abs(den) > 1e-11 * (np.abs(d)[None, :] + np.abs(d)[:, None])
        E = np.where(big, (S + d[None, :] * R) / np.where(big, den, 1.0), 0.5 * R)
        E[np.diag_indices(r)] = 0.5 * np.diag(R)
        off = S - np.diag(np.diag(S))
        out.append((np.abs(off).max() / np.abs(np.diag(S)).max(), np.abs(E).max()))
        V = V + V @ E
    return V, out


rng = np.random.default_rng(5)
theta = pkg.random_initial_parameters(model, 1)
for step in range(n_steps):
    N0 = nmat(theta)
    w0, V0 = np.linalg.eigh(N0)
    z = rng.normal(size=r)
    theta_p = O.propose(om, ot, pp, theta, z)
    N1 = nmat(theta_p)
    w1, V1 = np.linalg.eigh(N1)
    gaps = np.diff(w1) / w1[1:]
    V, hist = refine(N1, V0.copy())
    # eigenvector error against eigh (sign-free), weighted: what the proposal sees is sum_i sqrt(1/mu_i) z_i v_i
    Sg = np.sign(np.sum(V * V1, axis=0)); Sg[Sg == 0] = 1
    err = np.abs(V * Sg[None, :] - V1).max()
    samp = lambda VV, ww: (VV * (D[:, None] / np.sqrt(ww)[None, :])) @ z
    d = np.diag(V.T @ N1 @ V)
    order = np.argsort(d)
    e_s = np.abs(samp(V[:, order] * np.sign(np.sum(V[:, order] * V1, axis=0))[None, :], d[order]) - samp(V1, w1)).max() / np.abs(samp(V1, w1)).max()
    print("step %2d  |dtheta| %.3g  min rel gap %.2e  offdiag/iter %s  maxE %s  vec err %.1e  sample err %.1e" % (
        step, np.abs(theta_p[10:] - theta[10:]).max(), gaps.min(), " ".join("%.0e" % h[0] for h in hist), " ".join("%.0e" % h[1] for h in hist), err, e_s))
    theta = theta_p  # (accept everything: the walk the chain would take at its most mobile)
