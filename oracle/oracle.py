"""ctypes wrapper around oracle/libicp_oracle.so (the CPU restatement; see icp_oracle.c header).

TEST INFRASTRUCTURE ONLY — importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
PARITY UNPINNED — see the header of icp_oracle.c.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

c_double_p = C.POINTER(C.c_double)
c_int_p = C.POINTER(C.c_int)
c_ubyte_p = C.POINTER(C.c_ubyte)

MODEL_SAMPLING, TARGET_SAMPLING = 0, 1
MODEL_TO_TARGET, TARGET_TO_MODEL, SYMMETRIC = 0, 1, 2
EVAL_INDEPENDENT, EVAL_HAUSDORFF, EVAL_COLLECTIVE = 0, 1, 2


class ProposalParams(C.Structure):
    _fields_ = [("step_length", C.c_double), ("tangential_noise", C.c_double), ("noise_along_normal", C.c_double),
                ("direction", C.c_int), ("boundary_aware", C.c_int), ("n_model_ids", C.c_int),
                ("n_target_pts", C.c_int), ("target_pts", c_double_p)]


class Posterior(C.Structure):
    _fields_ = [("K", C.c_int), ("corr_id", c_int_p), ("corr_aux", c_int_p), ("corr_pt", c_double_p),
                ("keep", c_ubyte_p), ("alpha", c_double_p), ("M", c_double_p), ("Minv", c_double_p),
                ("V", c_double_p), ("S", c_double_p)]


class EvaluatorParams(C.Structure):
    _fields_ = [("kind", C.c_int), ("mode", C.c_int), ("n_model_ids", C.c_int), ("n_target_pts", C.c_int),
                ("target_pts", c_double_p), ("p0", C.c_double), ("p1", C.c_double), ("p2", C.c_double)]


class FitParams(C.Structure):
    _fields_ = [("direction", C.c_int), ("n_model_ids", C.c_int), ("model_ids", c_int_p), ("n_target_pts", C.c_int),
                ("target_pts", c_double_p), ("step_length", C.c_double)]


class ChainConfig(C.Structure):
    _fields_ = [("n_icp", C.c_int), ("icp", ProposalParams * 2), ("icp_weight", C.c_double * 2),
                ("w_icp", C.c_double), ("w_rw", C.c_double), ("rw_sigma", C.c_double), ("eval", EvaluatorParams),
                ("w_pose", C.c_double), ("pose_rot_sigma", C.c_double * 3), ("pose_trans_sigma", C.c_double * 3)]


def build(force: bool = False) -> str:
    so = os.path.join(_HERE, "libicp_oracle.so")
    src = os.path.join(_HERE, "icp_oracle.c")
    newest = max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("icp_oracle.c", "icp_spatial.c", "icp_spatial.h"))
    if force or not os.path.exists(so) or os.path.getmtime(so) < newest:
        subprocess.run(["make", "-C", _HERE, "-B", "libicp_oracle.so"], check=True, stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.orc_model_create.restype = C.c_void_p
        L.orc_model_create.argtypes = [C.c_int, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_int_p]
        L.orc_model_destroy.argtypes = [C.c_void_p]
        L.orc_mesh_create.restype = C.c_void_p
        L.orc_mesh_create.argtypes = [C.c_int, C.c_int, c_double_p, c_int_p]
        L.orc_mesh_destroy.argtypes = [C.c_void_p]
        L.orc_model_boundary.argtypes = [C.c_void_p, c_ubyte_p]
        L.orc_mesh_boundary.argtypes = [C.c_void_p, c_ubyte_p]
        L.orc_rotation_matrix.argtypes = [C.c_double, C.c_double, C.c_double, c_double_p]
        L.orc_instance.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.orc_vertex_normals.argtypes = [C.c_void_p, c_double_p, c_double_p]
        L.orc_nearest_vertex.argtypes = [C.c_int, c_double_p, C.c_int, c_double_p, c_int_p, c_double_p]
        L.orc_closest_point_on_surface.argtypes = [C.c_int, c_double_p, c_double_p, C.c_int, c_int_p, c_double_p,
                                                   c_int_p, c_double_p]
        L.orc_surface_noise_cov.argtypes = [c_double_p, C.c_double, C.c_double, c_double_p]
        L.orc_sym_eigen.argtypes = [C.c_int, c_double_p, c_double_p, c_double_p]
        L.orc_icp_posterior.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ProposalParams), c_double_p, C.POINTER(Posterior)]
        L.orc_propose_from_posterior.argtypes = [C.c_void_p, C.POINTER(ProposalParams), C.POINTER(Posterior),
                                                 c_double_p, c_double_p, c_double_p]
        L.orc_propose.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ProposalParams), c_double_p, c_double_p, c_double_p]
        L.orc_log_transition_from_posterior.argtypes = [C.c_void_p, C.POINTER(ProposalParams), C.POINTER(Posterior),
                                                        c_double_p, c_double_p, c_double_p]
        L.orc_log_transition.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ProposalParams), c_double_p, c_double_p, c_double_p]
        L.orc_prior_log_value.restype = C.c_double
        L.orc_prior_log_value.argtypes = [C.c_int, c_double_p]
        L.orc_evaluator_log_value.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(EvaluatorParams), c_double_p, c_double_p]
        L.orc_rng_uniform.restype = C.c_double
        L.orc_rng_uniform.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_rng_normal.restype = C.c_double
        L.orc_rng_normal.argtypes = [C.c_uint64, C.c_uint64, C.c_uint64]
        L.orc_fit_deterministic.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(FitParams), c_double_p, C.c_int, C.c_int, c_double_p, c_double_p]
        L.orc_posterior_variability.argtypes = [C.c_void_p, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p]
        L.orc_run_chain.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ChainConfig), c_double_p, C.c_uint64, C.c_int,
                                    c_ubyte_p, c_int_p, c_double_p, c_double_p]
        L.orc_pose_log_transition.restype = C.c_double
        L.orc_pose_log_transition.argtypes = [C.c_int, C.c_int, C.c_double, c_double_p, c_double_p]
        L.orc_pose_mixture_log_transition.restype = C.c_double
        L.orc_pose_mixture_log_transition.argtypes = [C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]
        L.orc_chain_log_transition.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(ChainConfig), c_double_p, c_double_p, c_double_p]
        _LIB = L
    return _LIB


def _d(a):
    return a.ctypes.data_as(c_double_p)


def _i(a):
    return a.ctypes.data_as(c_int_p)


def _u(a):
    return a.ctypes.data_as(c_ubyte_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


class OracleModel:
    def __init__(self, ref_points, cells, mean_def, basis, variance):
        self.ref = _f64(ref_points)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.mean = _f64(mean_def)
        self.basis = _f64(basis)
        self.lam = _f64(variance)
        self.N, self.T, self.r = self.ref.shape[0], self.cells.shape[0], self.lam.shape[0]
        self.h = lib().orc_model_create(self.N, self.T, self.r, _d(self.ref), _d(self.mean), _d(self.basis),
                                        _d(self.lam), _i(self.cells))

    @classmethod
    def from_model(cls, m):
        return cls(m.ref_points, m.cells, m.mean_def, m.basis, m.variance)

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_model_destroy(self.h)
            self.h = None

    def boundary(self):
        out = np.zeros(self.N, dtype=np.uint8)
        lib().orc_model_boundary(self.h, _u(out))
        return out

    def instance(self, theta):
        theta = _f64(theta)
        x = np.empty((self.N, 3))
        lib().orc_instance(self.h, _d(theta), _d(x))
        return x

    def vertex_normals(self, x):
        x = _f64(x)
        n = np.empty((self.N, 3))
        lib().orc_vertex_normals(self.h, _d(x), _d(n))
        return n


class OracleMesh:
    def __init__(self, points, cells):
        self.pts = _f64(points)
        self.cells = np.ascontiguousarray(cells, dtype=np.int32)
        self.M, self.T = self.pts.shape[0], self.cells.shape[0]
        self.h = lib().orc_mesh_create(self.M, self.T, _d(self.pts), _i(self.cells))

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_mesh_destroy(self.h)
            self.h = None

    def boundary(self):
        out = np.zeros(self.M, dtype=np.uint8)
        lib().orc_mesh_boundary(self.h, _u(out))
        return out


def rotation_matrix(phi, theta, psi):
    R = np.empty((3, 3))
    lib().orc_rotation_matrix(phi, theta, psi, _d(R))
    return R


SEARCH_BRUTE, SEARCH_TREES, SEARCH_BRUTE_OMP = 0, 1, 2


def set_search_backend(backend: int, n_threads: int = 0):
    """CPU-baseline variants of BASELINE.md §3 (icp_spatial.h): 0 the restatement's single-thread scans, 1 = B1 (KD-tree +
    bounding-volume hierarchy, rebuilt for every new mesh), 2 = B2 (the scans over n_threads cores, 0 = all).  Same results."""
    L = lib()
    L.orc_set_search_backend.argtypes = [C.c_int, C.c_int]
    L.orc_set_search_backend.restype = None
    L.orc_set_search_backend(int(backend), int(n_threads))


def search_stats():
    L = lib()
    kd, bvh = C.c_long(), C.c_long()
    L.orc_search_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
    L.orc_search_stats.restype = None
    L.orc_search_stats(C.byref(kd), C.byref(bvh))
    return kd.value, bvh.value


def nearest_vertex(q, pts):
    q, pts = _f64(q).reshape(-1, 3), _f64(pts)
    idx = np.empty(q.shape[0], dtype=np.int32)
    d2 = np.empty(q.shape[0])
    lib().orc_nearest_vertex(q.shape[0], _d(q), pts.shape[0], _d(pts), _i(idx), _d(d2))
    return idx, d2


def closest_point_on_surface(q, pts, cells):
    q, pts = _f64(q).reshape(-1, 3), _f64(pts)
    cells = np.ascontiguousarray(cells, dtype=np.int32)
    cp = np.empty_like(q)
    tri = np.empty(q.shape[0], dtype=np.int32)
    d2 = np.empty(q.shape[0])
    lib().orc_closest_point_on_surface(q.shape[0], _d(q), _d(pts), cells.shape[0], _i(cells), _d(cp), _i(tri), _d(d2))
    return cp, tri, d2


def surface_noise_cov(normal, sd_normal, sd_tangent):
    n = _f64(normal)
    cov = np.empty((3, 3))
    lib().orc_surface_noise_cov(_d(n), sd_normal, sd_tangent, _d(cov))
    return cov


def sym_eigen(a):
    a = _f64(a)
    n = a.shape[0]
    w, V = np.empty(n), np.empty((n, n))
    lib().orc_sym_eigen(n, _d(a), _d(w), _d(V))
    return w, V


def proposal_params(step_length, tangential_noise, noise_along_normal, direction, boundary_aware=True,
                    n_model_ids=0, target_pts=None):
    tp = _f64(target_pts).reshape(-1, 3) if target_pts is not None else np.zeros((0, 3))
    p = ProposalParams(step_length, tangential_noise, noise_along_normal, direction, int(boundary_aware),
                       int(n_model_ids), tp.shape[0], _d(tp))
    p._keep = tp
    return p


def evaluator_params(kind, mode, n_model_ids=0, target_pts=None, p0=0.0, p1=1.0, p2=1.0):
    tp = _f64(target_pts).reshape(-1, 3) if target_pts is not None else np.zeros((0, 3))
    e = EvaluatorParams(kind, mode, int(n_model_ids), tp.shape[0], _d(tp), p0, p1, p2)
    e._keep = tp
    return e


class PosteriorResult:
    def __init__(self, K, r):
        self.corr_id = np.full(max(K, 1), -1, dtype=np.int32)
        self.corr_aux = np.full(max(K, 1), -1, dtype=np.int32)
        self.corr_pt = np.zeros((max(K, 1), 3))
        self.keep = np.zeros(max(K, 1), dtype=np.uint8)
        self.alpha = np.zeros(r)
        self.M = np.zeros((r, r))
        self.Minv = np.zeros((r, r))
        self.V = np.zeros((r, r))
        self.S = np.zeros(r)
        self.K = K
        self.c = Posterior(K, _i(self.corr_id), _i(self.corr_aux), _d(self.corr_pt), _u(self.keep), _d(self.alpha),
                           _d(self.M), _d(self.Minv), _d(self.V), _d(self.S))


def icp_posterior(model: OracleModel, target: OracleMesh, pp: ProposalParams, theta) -> PosteriorResult:
    theta = _f64(theta)
    K = pp.n_target_pts if pp.direction == TARGET_SAMPLING else pp.n_model_ids
    res = PosteriorResult(K, model.r)
    rc = lib().orc_icp_posterior(model.h, target.h, C.byref(pp), _d(theta), C.byref(res.c))
    if rc != 0:
        raise RuntimeError(f"orc_icp_posterior rc={rc}")
    for name in ("corr_id", "corr_aux", "corr_pt", "keep"):
        setattr(res, name, getattr(res, name)[:K])
    return res


def propose(model, target, pp, theta, z):
    theta, z = _f64(theta), _f64(z)
    out = np.empty_like(theta)
    rc = lib().orc_propose(model.h, target.h, C.byref(pp), _d(theta), _d(z), _d(out))
    if rc != 0:
        raise RuntimeError(f"orc_propose rc={rc}")
    return out


def log_transition(model, target, pp, theta_from, theta_to):
    a, b = _f64(theta_from), _f64(theta_to)
    out = C.c_double()
    rc = lib().orc_log_transition(model.h, target.h, C.byref(pp), _d(a), _d(b), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_log_transition rc={rc}")
    return out.value


def prior_log_value(r, theta):
    theta = _f64(theta)
    return lib().orc_prior_log_value(r, _d(theta))


def evaluator_log_value(model, target, ep, theta):
    theta = _f64(theta)
    out = C.c_double()
    rc = lib().orc_evaluator_log_value(model.h, target.h, C.byref(ep), _d(theta), C.byref(out))
    return out.value, rc


def chain_config(icp_params, icp_weights, w_icp, w_rw, rw_sigma, ep, w_pose=0.0, pose_rot_sigma=(0.01, 0.01, 0.01),
                 pose_trans_sigma=(0.1, 0.1, 0.1)):
    """Outer mixture (pose, ICP, shape walk) as apps/bfm/BfmFittingPartial.scala:70 builds it; w_pose = 0 leaves the pose walks
    out (apps/femur/IcpProposalRegistration.scala:72).  pose_rot_sigma = (rotYaw, rotPitch, rotRoll), pose_trans_sigma = (x, y, z):
    the argument order of MixedProposalDistributions.mixedRandomPoseProposal (:29)."""
    cfg = ChainConfig()
    cfg.n_icp = len(icp_params)
    for i, p in enumerate(icp_params):
        cfg.icp[i] = p
        cfg.icp_weight[i] = icp_weights[i]
    cfg.w_icp, cfg.w_rw, cfg.rw_sigma, cfg.eval = w_icp, w_rw, rw_sigma, ep
    cfg.w_pose = w_pose
    for a in range(3):
        cfg.pose_rot_sigma[a] = pose_rot_sigma[a]
        cfg.pose_trans_sigma[a] = pose_trans_sigma[a]
    cfg._keep = (icp_params, ep)
    return cfg


POSE_PARAM_INDEX = (6, 5, 4, 1, 2, 3)  # Yaw, Pitch, Roll, X, Y, Z -> index into allParameters (PoseProposals.scala:39-41)


def pose_log_transition(component, sd, theta_from, theta_to):
    """GaussianAxisRotationProposal / GaussianAxisTranslationProposal.logTransitionProbability (PoseProposals.scala:46-60, :77-88);
    component 0..5 = Yaw, Pitch, Roll, X, Y, Z."""
    a, b = _f64(theta_from), _f64(theta_to)
    return lib().orc_pose_log_transition(a.shape[0], int(component), float(sd), _d(a), _d(b))


def pose_mixture_log_transition(rot_sigma, trans_sigma, theta_from, theta_to):
    a, b = _f64(theta_from), _f64(theta_to)
    rs, ts = _f64(rot_sigma), _f64(trans_sigma)
    return lib().orc_pose_mixture_log_transition(a.shape[0], _d(rs), _d(ts), _d(a), _d(b))


def chain_log_transition(model, target, cfg, theta_from, theta_to):
    """The whole proposal mixture's logTransitionProbability(from, to) (Scalismo MixtureProposal: log-sum-exp over all leaves)."""
    a, b = _f64(theta_from), _f64(theta_to)
    out = C.c_double()
    rc = lib().orc_chain_log_transition(model.h, target.h, C.byref(cfg), _d(a), _d(b), C.byref(out))
    if rc != 0:
        raise RuntimeError(f"orc_chain_log_transition rc={rc}")
    return out.value


def run_chain(model, target, cfg, theta0, seed, n_steps):
    theta0 = _f64(theta0)
    P = theta0.shape[0]
    acc = np.zeros(n_steps, dtype=np.uint8)
    comp = np.zeros(n_steps, dtype=np.int32)
    logp = np.zeros(n_steps)
    states = np.zeros((n_steps, P))
    rc = lib().orc_run_chain(model.h, target.h, C.byref(cfg), _d(theta0), seed, n_steps, _u(acc), _i(comp), _d(logp), _d(states))
    if rc != 0:
        raise RuntimeError(f"orc_run_chain rc={rc}")
    return acc, comp, logp, states


def initial_theta(model_ref_points, rank):
    """ref: api/sampling/SamplingRegistration.scala:40-43 — zero pose/shape, rotation centre = mean reference point."""
    ctr = np.asarray(model_ref_points, dtype=np.float64).sum(axis=0) * 1.0 / model_ref_points.shape[0]
    theta = np.zeros(10 + rank)
    theta[0] = 1.0
    theta[7:10] = ctr
    return theta


def fit_deterministic(model, mesh, theta_init, n_iterations, sigma2_seq=(1.0, 0.1, 0.01), direction=MODEL_SAMPLING, model_ids=None,
                      target_pts=None, step_length=1.0):
    """IcpBasedSurfaceFitting.runfitting (api/other/IcpBasedSurfaceFitting.scala:46-126); returns the final parameter vector."""
    ids = np.ascontiguousarray(model_ids if model_ids is not None else np.zeros(0), dtype=np.int32)
    tp = _f64(target_pts if target_pts is not None else np.zeros((0, 3))).reshape(-1, 3)
    fp = FitParams(int(direction), ids.shape[0], _i(ids), tp.shape[0], _d(tp), float(step_length))
    th = _f64(theta_init)
    sig = _f64(sigma2_seq)
    out = np.zeros_like(th)
    rc = lib().orc_fit_deterministic(model.h, mesh.h, C.byref(fp), _d(th), int(n_iterations), sig.shape[0], _d(sig), _d(out))
    if rc != 0:
        raise RuntimeError(f"orc_fit_deterministic failed ({rc})")
    return out


def posterior_variability(model, thetas, mode=0, theta_ref=None):
    """apps/util/PosteriorVariability.scala:30-73 (mode 0 total, 1 along the normals of theta_ref's mesh, 2 along the mean sample normal)."""
    th = _f64(thetas)
    ref = _f64(theta_ref if theta_ref is not None else th[0])
    out = np.zeros(model.N)
    rc = lib().orc_posterior_variability(model.h, th.shape[0], _d(th), int(mode), _d(ref), _d(out))
    if rc != 0:
        raise RuntimeError("orc_posterior_variability failed")
    return out

