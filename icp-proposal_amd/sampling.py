"""Chain driver: Python face of libicp_host.so (C++ mirror of SamplingRegistration.runfitting,
api/sampling/SamplingRegistration.scala:45-93) plus the reference's experiment configurations.

The per-step loop runs in C++ (icp-proposal_amd/host/); Python only builds the configuration and receives the
fixed-size per-step records (layout: host/icp_host.h).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _native as nat
from . import data as _data

_HERE = os.path.dirname(os.path.abspath(__file__))
HOST_LIB_PATH = os.path.join(_HERE, "libicp_host.so")
RECORD_HEADER = 4
_HOST = None


class HostChainConfig(C.Structure):
    _fields_ = [("n_icp", C.c_int32), ("icp", nat.ProposalParams * 2), ("icp_weight", C.c_double * 2),
                ("w_icp", C.c_double), ("w_rw", C.c_double), ("w_pose", C.c_double), ("rw_sigma", C.c_double),
                ("pose_rot_sigma", C.c_double * 3), ("pose_trans_sigma", C.c_double * 3),
                ("eval", nat.EvaluatorParams), ("fused", C.c_int32), ("sampler", C.c_int32)]


def host_lib():
    global _HOST
    if _HOST is None:
        nat.lib()  # the C ABI library first (no fallback if it is missing)
        if not os.path.exists(HOST_LIB_PATH):
            raise nat.IcpNativeError(-2, "load", f"{HOST_LIB_PATH} not built")
        L = C.CDLL(HOST_LIB_PATH)
        L.icp_host_chain_create.restype = C.c_int
        L.icp_host_chain_create.argtypes = [C.c_void_p, C.POINTER(HostChainConfig), nat.c_double_p, C.c_uint64, C.POINTER(C.c_void_p)]
        L.icp_host_chain_run.restype = C.c_int
        L.icp_host_chain_run.argtypes = [C.c_void_p, C.c_int32, nat.c_double_p]
        L.icp_host_chains_run_batched.restype = C.c_int
        L.icp_host_chains_run_batched.argtypes = [C.POINTER(C.c_void_p), C.c_int32, C.c_int32, C.POINTER(nat.c_double_p)]
        L.icp_host_chain_state.restype = C.c_int
        L.icp_host_chain_state.argtypes = [C.c_void_p, nat.c_double_p, nat.c_double_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
        L.icp_host_chain_native_calls.restype = C.c_int
        L.icp_host_chain_native_calls.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        L.icp_host_chain_destroy.restype = None
        L.icp_host_chain_destroy.argtypes = [C.c_void_p]
        L.icp_host_last_error.restype = C.c_char_p
        L.icp_host_chain_log_transition.restype = C.c_int
        L.icp_host_chain_log_transition.argtypes = [C.c_void_p, nat.c_double_p, nat.c_double_p, nat.c_double_p]
        L.icp_host_pose_mixture_log_transition.restype = C.c_int
        L.icp_host_pose_mixture_log_transition.argtypes = [C.c_int32, nat.c_double_p, nat.c_double_p, nat.c_double_p, nat.c_double_p,
                                                           nat.c_double_p]
        L.icp_host_pose_mixture_propose.restype = C.c_int
        L.icp_host_pose_mixture_propose.argtypes = [C.c_int32, nat.c_double_p, nat.c_double_p, nat.c_double_p, C.c_uint64, C.c_uint64,
                                                    nat.c_double_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]
        L.icp_host_scala_double.restype = C.c_int
        L.icp_host_scala_double.argtypes = [C.c_double, C.c_char_p, C.c_int32]
        _HOST = L
    return _HOST


def _dp(a):
    return a.ctypes.data_as(nat.c_double_p)


# leaf id of a pose walk -> index of the parameter it perturbs in allParameters = [s | t(3) | rotation._1,_2,_3 | centre(3) | c]
POSE_LEAF_PARAMETER = {3: 6, 4: 5, 5: 4, 6: 1, 7: 2, 8: 3}


def scala_double_native(x: float) -> str:
    """The C++ harness' java.lang.Double.toString (what the native chain writes into generatedBy)."""
    buf = C.create_string_buffer(64)
    host_lib().icp_host_scala_double(float(x), buf, 64)
    return buf.value.decode()


def pose_mixture_log_transition(rot_sigma, trans_sigma, theta_from, theta_to) -> float:
    """mixedRandomPoseProposal(...).logTransitionProbability(from, to) (MixedProposalDistributions.scala:29-39) — host arithmetic."""
    a = np.ascontiguousarray(theta_from, dtype=np.float64)
    b = np.ascontiguousarray(theta_to, dtype=np.float64)
    rs, ts = np.ascontiguousarray(rot_sigma, dtype=np.float64), np.ascontiguousarray(trans_sigma, dtype=np.float64)
    out = C.c_double()
    st = host_lib().icp_host_pose_mixture_log_transition(a.shape[0], _dp(rs), _dp(ts), _dp(a), _dp(b), C.byref(out))
    if st != 0:
        raise nat.IcpNativeError(st, "icp_host_pose_mixture_log_transition", (host_lib().icp_host_last_error() or b"").decode())
    return out.value


def pose_mixture_propose(rot_sigma, trans_sigma, theta, seed: int, step: int):
    """One propose() of mixedRandomPoseProposal with the chain's random numbers of (seed, step): (theta', leaf id, generatedBy)."""
    a = np.ascontiguousarray(theta, dtype=np.float64)
    rs, ts = np.ascontiguousarray(rot_sigma, dtype=np.float64), np.ascontiguousarray(trans_sigma, dtype=np.float64)
    out = np.zeros_like(a)
    leaf = C.c_int32()
    name = C.create_string_buffer(96)
    st = host_lib().icp_host_pose_mixture_propose(a.shape[0], _dp(rs), _dp(ts), _dp(a), seed, step, _dp(out), C.byref(leaf), name, 96)
    if st != 0:
        raise nat.IcpNativeError(st, "icp_host_pose_mixture_propose", (host_lib().icp_host_last_error() or b"").decode())
    return out, leaf.value, name.value.decode()


class ChainSetup:
    """Plain description of one experiment: proposals + evaluator (everything the reference sets in a `main`)."""

    def __init__(self):
        self.icp = []          # list of dicts: direction, step, sigma_t, sigma_n, boundary_aware, n_model_ids, target_pts, weight
        self.w_icp, self.w_rw, self.w_pose = 0.9, 0.1, 0.0
        self.rw_sigma = 0.1
        self.pose_rot_sigma = (0.01, 0.01, 0.01)
        self.pose_trans_sigma = (0.1, 0.1, 0.1)
        self.eval = dict(kind=0, mode=0, n_model_ids=0, target_pts=np.zeros((0, 3)), gauss_mean=0.0, gauss_sigma=1.0, exp_rate=1.0)
        # 0 per-method calls, 1 icp_chain_eval_step prefetch, 2 whole step in one icp_chain_step submission, 3 per-method calls as
        # Scalismo's MetropolisHastings.next makes them over a chain bound with icp_chain_bind (the drop-in path)
        self.fused = 2
        self.sampler = "eigen"  # or "cholesky-root" (opt-in, not the reference's arithmetic: NonRigidIcpProposal.setSampler)

    @staticmethod
    def scala_double(x: float) -> str:
        """java.lang.Double.toString, which Scala's string interpolation uses for the proposal names
        (api/sampling/MixedProposalDistributions.scala:31-54: s"RandomShape-$sd" -> "RandomShape-0.1"): the shortest decimal that
        round-trips, always with a fractional part, computerised scientific notation below 1e-3 and from 1e7 ("1.0E-4")."""
        x = float(x)
        if x != x:
            return "NaN"
        if x in (float("inf"), float("-inf")):
            return "Infinity" if x > 0 else "-Infinity"
        if x == 0.0:
            return "-0.0" if str(x).startswith("-") else "0.0"
        # repr() gives the shortest digits that round-trip (what Double.toString prints); only the layout differs
        import decimal
        sign, digits, exp10 = decimal.Decimal(repr(x)).as_tuple()
        digits = list(digits)
        while len(digits) > 1 and digits[-1] == 0:
            digits.pop()
            exp10 += 1
        ex = exp10 + len(digits) - 1          # decimal exponent of the first digit
        ds = "".join(str(d) for d in digits)
        neg = "-" if sign else ""
        if 1e-3 <= abs(x) < 1e7:
            if ex >= 0:
                ds = ds.ljust(ex + 2, "0")
                return neg + ds[:ex + 1] + "." + ds[ex + 1:]
            return neg + "0." + "0" * (-ex - 1) + ds
        return "%s%s.%sE%d" % (neg, ds[0], ds[1:] or "0", ex)

    def leaf_names(self):
        """generatedBy strings of the leaf proposals, indexed by the leaf id of the per-step records (host/icp_host.cpp).
        Leaves 3..8 are the pose walks in the order of MixedProposalDistributions.scala:31-36 — Yaw, Pitch, Roll, X, Y, Z — with
        pose_rot_sigma = (rotYaw, rotPitch, rotRoll); the parameter each one perturbs is POSE_LEAF_PARAMETER[leaf]
        (YawAxis -> rotation._3 = theta[6], PitchAxis -> theta[5], RollAxis -> theta[4]: PoseProposals.scala:39-41)."""
        names = {}
        for i, p in enumerate(self.icp):
            names[i] = "IcpProposal-%s-%sStep" % ("TargetSampling" if p["direction"] == 1 else "ModelSampling", self.scala_double(p["step"]))
        names[2] = "RandomShape-%s" % self.scala_double(self.rw_sigma)
        for a, nm in enumerate(("RotationYaw", "RotationPitch", "RotationRoll", "TranslationX", "TranslationY", "TranslationZ")):
            names[3 + a] = "%s-%s" % (nm, self.scala_double(self.pose_rot_sigma[a] if a < 3 else self.pose_trans_sigma[a - 3]))
        return names

    def to_c(self):
        cfg = HostChainConfig()
        keep = []
        cfg.n_icp = len(self.icp)
        for i, p in enumerate(self.icp):
            tp = np.ascontiguousarray(p.get("target_pts", np.zeros((0, 3))), dtype=np.float64).reshape(-1, 3)
            keep.append(tp)
            cfg.icp[i] = nat.ProposalParams(p["step"], p["sigma_t"], p["sigma_n"], p["direction"], int(p.get("boundary_aware", True)),
                                            int(p.get("n_model_ids", 0)), tp.shape[0], _dp(tp))
            cfg.icp_weight[i] = p.get("weight", 0.5)
        cfg.w_icp, cfg.w_rw, cfg.w_pose, cfg.rw_sigma = self.w_icp, self.w_rw, self.w_pose, self.rw_sigma
        for a in range(3):
            cfg.pose_rot_sigma[a] = self.pose_rot_sigma[a]
            cfg.pose_trans_sigma[a] = self.pose_trans_sigma[a]
        e = self.eval
        tp = np.ascontiguousarray(e["target_pts"], dtype=np.float64).reshape(-1, 3)
        keep.append(tp)
        cfg.eval = nat.EvaluatorParams(e["kind"], e["mode"], int(e["n_model_ids"]), tp.shape[0], _dp(tp), e["gauss_mean"],
                                       e["gauss_sigma"], e["exp_rate"])
        cfg.fused = int(self.fused)
        cfg.sampler = {"eigen": 0, "cholesky-root": 1}[self.sampler]
        cfg._keep = keep
        return cfg


def femur_icp_proposal_registration(model, target, n_icp_points=None, n_eval_points=None, direction="ModelAndTargetSampling",
                                    eval_mode=0, fused=2) -> ChainSetup:
    """The configuration of apps/femur/IcpProposalRegistration.scala:59-85: K = 2·rank proposal points, 4·rank evaluator
    points, 0.9 ICP(Model+Target, σt=10, σn=5, step 0.1) + 0.1 random walk(0.1), prior × independent Gaussian(0, 2)."""
    r = model.rank
    k_icp = 2 * r if n_icp_points is None else n_icp_points
    k_ev = 4 * r if n_eval_points is None else n_eval_points
    s = ChainSetup()
    tp = _data.decimated_point_subset(target, k_icp)
    if direction in ("TargetSampling", "ModelAndTargetSampling"):  # MixedProposalDistributions.scala:56-65 order
        s.icp.append(dict(direction=1, step=0.1, sigma_t=10.0, sigma_n=5.0, boundary_aware=True, target_pts=tp, weight=0.5))
    if direction in ("ModelSampling", "ModelAndTargetSampling"):
        s.icp.append(dict(direction=0, step=0.1, sigma_t=10.0, sigma_n=5.0, boundary_aware=True, n_model_ids=min(k_icp, model.n_points), weight=0.5))
    s.eval = dict(kind=0, mode=eval_mode, n_model_ids=min(k_ev, model.n_points), target_pts=_data.decimated_point_subset(target, k_ev),
                  gauss_mean=0.0, gauss_sigma=2.0, exp_rate=1.0)
    s.fused = fused
    return s


def femur_random_init_comparison(model, target, fused=2) -> ChainSetup:
    """The ICP chain of apps/femur/RunMHRandomInitComparison.scala:54-61 (BASELINE.json configs[2]): every model point is a
    sample point of the proposal and of the evaluator (:54-55), ICP mixture with ModelSampling only (:59, no random walk in
    this chain), prior × independent Gaussian(0, 2) with SymmetricEvaluation (:61)."""
    n = model.n_points
    s = ChainSetup()
    s.icp.append(dict(direction=0, step=0.1, sigma_t=10.0, sigma_n=5.0, boundary_aware=True, n_model_ids=n, weight=0.5))
    s.w_icp, s.w_rw, s.w_pose = 1.0, 0.0, 0.0
    s.eval = dict(kind=0, mode=2, n_model_ids=n, target_pts=_data.decimated_point_subset(target, n),
                  gauss_mean=0.0, gauss_sigma=2.0, exp_rate=1.0)
    s.fused = fused
    return s


def bfm_fitting_partial(model, target, evaluator: str = "collective", fused=2) -> ChainSetup:
    """The configuration of apps/bfm/BfmFittingPartial.scala:62-83 (BASELINE.json configs[3], configs[4]): 2·rank ICP points,
    4·rank evaluator points (:65,78); 0.4 pose + 0.55 ICP(ModelSampling, σt = 6, σn = 3, step 0.1) + 0.05 random walk (:66-70);
    evaluator: the boundary-aware collective average/Hausdorff likelihood with symmetric evaluation (σ_avg 0.3, rate 1.0,
    mean 0.1, :80), or the full-mesh Hausdorff evaluator (evaluators/HausdorffDistanceEvaluator.scala, Exponential(1))."""
    r = model.rank
    s = ChainSetup()
    s.icp.append(dict(direction=0, step=0.1, sigma_t=6.0, sigma_n=3.0, boundary_aware=True, n_model_ids=min(2 * r, model.n_points),
                      weight=0.5))
    s.w_pose, s.w_icp, s.w_rw = 0.4, 0.55, 0.05
    s.rw_sigma = 0.1
    if evaluator == "hausdorff":
        s.eval = dict(kind=1, mode=2, n_model_ids=0, target_pts=np.zeros((0, 3)), gauss_mean=0.0, gauss_sigma=1.0, exp_rate=1.0)
    else:
        s.eval = dict(kind=2, mode=2, n_model_ids=min(4 * r, model.n_points), target_pts=_data.decimated_point_subset(target, 4 * r),
                      gauss_mean=0.1, gauss_sigma=0.3, exp_rate=1.0)
    s.fused = fused
    return s


def random_initial_parameters(model, chain_index: int, seed: int = 1024) -> np.ndarray:
    """Chain i > 0 starts from shape coefficients c ~ N(0, 0.1·I) (apps/femur/RandomSamplesFromModel.scala:26-35 draws the
    stored start shapes that way); chain 0 from the mean."""
    from .api import initial_parameters as _init
    theta = _init(model)
    if chain_index > 0:
        theta[10:] = np.random.default_rng(seed + chain_index).normal(size=model.rank) * np.sqrt(0.1)
    return theta


class SamplingRegistration:
    """SamplingRegistration (api/sampling/SamplingRegistration.scala:37-93) over one IcpContext."""

    def __init__(self, ctx, setup: ChainSetup, initial_parameters=None, seed: int = 1024):
        from .api import initial_parameters as _init
        self.ctx, self.setup = ctx, setup
        self.P = 10 + ctx.rank
        theta0 = np.ascontiguousarray(initial_parameters if initial_parameters is not None else _init(ctx.model), dtype=np.float64)
        self._cfg = setup.to_c()
        h = C.c_void_p()
        st = host_lib().icp_host_chain_create(ctx.h, C.byref(self._cfg), _dp(theta0), seed, C.byref(h))
        if st != 0:
            raise nat.IcpNativeError(st, "icp_host_chain_create", (host_lib().icp_host_last_error() or b"").decode())
        self.h = h
        if hasattr(ctx, "_adopt"):
            ctx._adopt(self)  # (IcpContext.close() closes the chain — it owns proposals and an evaluator of the context — first)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            host_lib().icp_host_chain_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def run(self, n_steps: int, want_records: bool = True):
        """n_steps more MH steps; returns records [n_steps, 4 + 10 + r] (index, accepted, leaf id, log value, theta)."""
        rec = np.zeros((n_steps, RECORD_HEADER + self.P)) if want_records else None
        st = host_lib().icp_host_chain_run(self.h, n_steps, _dp(rec) if want_records else None)
        if st != 0:
            raise nat.IcpNativeError(st, "icp_host_chain_run", (host_lib().icp_host_last_error() or b"").decode())
        return rec

    def logTransitionProbability(self, theta_from, theta_to) -> float:
        """The chain's whole proposal mixture (Scalismo MixtureProposal.logTransitionProbability: log-sum-exp over all leaves)."""
        a = np.ascontiguousarray(theta_from, dtype=np.float64)
        b = np.ascontiguousarray(theta_to, dtype=np.float64)
        out = C.c_double()
        st = host_lib().icp_host_chain_log_transition(self.h, _dp(a), _dp(b), C.byref(out))
        if st != 0:
            raise nat.IcpNativeError(st, "icp_host_chain_log_transition", (host_lib().icp_host_last_error() or b"").decode())
        return out.value

    def native_calls(self) -> dict:
        """Per-method native calls the chain's adapters have made (fused = 0 / 3: what a Scalismo-driven chain costs at the boundary)."""
        out = (C.c_int64 * 5)()
        host_lib().icp_host_chain_native_calls(self.h, out)
        return {"proposal_calls": int(out[0]), "log_value_calls": int(out[1]), "bound_steps_from_propose": int(out[2]),
                "bound_steps_from_log_value": int(out[3]), "parked_transition_hits": int(out[4])}

    def state(self):
        theta = np.zeros(self.P)
        logp = C.c_double()
        n, a = C.c_int64(), C.c_int64()
        host_lib().icp_host_chain_state(self.h, _dp(theta), C.byref(logp), C.byref(n), C.byref(a))
        return theta, logp.value, n.value, a.value


def run_chains_batched(chains, n_steps: int, want_records: bool = True):
    """n_steps more steps of several SamplingRegistration chains (one IcpContext each, same setup) in lockstep: every
    step is ONE icp_chain_step_batched submission for all of them (icp_host_chains_run_batched).  Returns the list of the
    chains' record arrays — chain by chain what chain.run(n_steps) returns."""
    B = len(chains)
    recs = [np.zeros((n_steps, RECORD_HEADER + c.P)) for c in chains] if want_records else None
    hs = (C.c_void_p * B)(*[c.h for c in chains])
    rp = (nat.c_double_p * B)(*[_dp(r) for r in recs]) if want_records else None
    st = host_lib().icp_host_chains_run_batched(hs, B, n_steps, rp)
    if st != 0:
        raise nat.IcpNativeError(st, "icp_host_chains_run_batched", (host_lib().icp_host_last_error() or b"").decode())
    return recs
