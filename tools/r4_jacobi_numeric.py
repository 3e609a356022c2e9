#!/usr/bin/env python3
"""CPU experiment (numpy + the oracle): how far is the warm-started Jacobi iteration of consecutive posteriors from done after each
sweep, and what does a first- / second-order perturbation correction leave?  (Is 'one sweep fewer + a second-order correction' a lead
for the accepted step of the headline?  NOTES, round 4.)
usage: r4_jacobi_numeric.py [config 0|1|2] [steps]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import __graft_entry__ as graft
pkg = graft.load_package()
from oracle import oracle as O
O.lib()
config = int(sys.argv[1]) if len(sys.argv) > 1 else 1
n_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 26
if config == 0:
    model, target = pkg.data.load_femur_model_and_target(50)
    setup = pkg.femur_icp_proposal_registration(model, target)
elif config == 1:
    model, target = pkg.data.synthetic_femur_target(n_subdiv=6)
    setup = pkg.femur_icp_proposal_registration(model, target)
else:
    model, target = pkg.data.synthetic_femur_target(n_subdiv=6, n_components=100)
    setup = pkg.femur_random_init_comparison(model, target)
O.set_search_backend(1, 0)
om, ot = O.OracleModel.from_model(model), O.OracleMesh(target.points, target.cells)
r = model.rank
pps = [O.proposal_params(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], p.get("boundary_aware", True),
                         n_model_ids=p.get("n_model_ids", 0), target_pts=p.get("target_pts")) for p in setup.icp]
sl = np.sqrt(np.asarray(model.variance, dtype=np.float64))

def round_robin_sweep(A, V):
    """one sweep in parallel (round-robin) ordering: n/2 disjoint rotations per round, n-1 rounds — the ordering of the device kernel's
    family (the cyclic-by-rows ordering converges alike)"""
    n = A.shape[0]
    m = n + (n & 1)
    idx = list(range(m))
    for _ in range(m - 1):
        J = np.eye(n)
        for k in range(m // 2):
            p, q = idx[k], idx[m - 1 - k]
            if p >= n or q >= n: continue
            if p > q: p, q = q, p
            apq = A[p, q]
            if apq == 0.0: continue
            tau = (A[q, q] - A[p, p]) / (2.0 * apq)
            t = (1.0 if tau >= 0 else -1.0) / (abs(tau) + np.sqrt(1.0 + tau * tau))
            c = 1.0 / np.sqrt(1.0 + t * t); s = t * c
            J[p, p] = c; J[q, q] = c; J[p, q] = s; J[q, p] = -s
        A = J.T @ A @ J; V = V @ J
        idx = [idx[0]] + [idx[-1]] + idx[1:-1]
    return A, V

def ratio(A):
    d = np.diag(A); gap = np.abs(d[None, :] - d[:, None]) + np.eye(len(d))
    R = np.abs(A) / gap; np.fill_diagonal(R, 0.0)
    return R.max()

def corrections(A):
    d = np.diag(A).copy(); E = A - np.diag(d)
    den = d[None, :] - d[:, None]; np.fill_diagonal(den, 1.0)     # den[k, i] = d_i - d_k
    X1 = E / den; np.fill_diagonal(X1, 0.0)
    X2 = (E @ X1) / den
    np.fill_diagonal(X2, -0.5 * np.sum(X1 * X1, axis=0))
    lam2 = d + np.sum(E * X1, axis=0)                 # second order
    lam3 = lam2 + np.sum(E * X2, axis=0) - 0.0        # + third order (E_ii = 0)
    return X1, X2, d, lam2, lam3

def vec_err(Vapprox, Vexact):
    Va = Vapprox / np.linalg.norm(Vapprox, axis=0)
    sg = np.sign(np.sum(Va * Vexact, axis=0))
    return np.abs(Va * sg - Vexact).max()

theta = O.initial_theta(model.ref_points, r)
rng = np.random.default_rng(5)
Vprev = [None] * len(pps)
worst = {}
print("columns per sweep: max |A_ij|/gap, max|X2|, eigenvector error 1st order, 2nd order, eigenvalue rel. error 0th/2nd/3rd order")
for step in range(n_steps):
    for di, pp in enumerate(pps):
        post = O.icp_posterior(om, ot, pp, theta)
        N = post.M / np.outer(sl, sl); N = 0.5 * (N + N.T)
        w, Vex = np.linalg.eigh(N)
        if Vprev[di] is not None:
            A = Vprev[di].T @ N @ Vprev[di]; V = Vprev[di].copy()
            mingap = np.min(np.diff(w)) / w[-1]
            line = "%3d dir %d gap %.0e start %.0e |" % (step, di, mingap, ratio(A))
            for sweeps in (1, 2, 3):
                A, V = round_robin_sweep(A, V)
                order = np.argsort(np.diag(A))
                X1, X2, l0, l2, l3 = corrections(A)
                e1 = vec_err((V @ (np.eye(r) + X1))[:, order], Vex)
                e2 = vec_err((V @ (np.eye(r) + X1 + X2))[:, order], Vex)
                le = [np.max(np.abs(l[order] - w) / w) for l in (l0, l2, l3)]
                rt = ratio(A)
                line += " %.0e %.0e %.0e %.0e (%.0e %.0e %.0e) |" % (rt, np.abs(X2 - np.diag(np.diag(X2))).max(), e1, e2, le[0], le[1], le[2])
                for lo, hi in ((1e-4, 2e-4), (2e-4, 4e-4), (4e-4, 8e-4), (8e-4, 1.6e-3)):
                    if lo < rt <= hi: worst[hi] = max(worst.get(hi, 0.0), e2)
            print(line, flush=True)
        Vprev[di] = Vex
    di = int(rng.integers(len(pps)))
    theta = O.propose(om, ot, pps[di], theta, rng.normal(size=r))
print("worst second-order eigenvector error by ratio bucket (upper edge):", {k: "%.1e" % v for k, v in sorted(worst.items())})
