"""Host-side mirror of the reference's plug-in surface for the closest-point-proposal path.

Same names, argument meaning and error behaviour as the reference's Scala classes (paths relative to the
reference's src/main/scala/), implemented as thin wrappers over the C ABI (include/icp_proposal.h):

    ModelFittingParameters                 api/sampling/ModelFittingParameters.scala:47-66
    NonRigidIcpProposal                    api/sampling/proposals/NonRigidIcpProposal.scala:30-155
    IndependentPointDistanceEvaluator      api/sampling/evaluators/IndependentPointDistanceEvaluator.scala:27-67
    HausdorffDistanceEvaluator             api/sampling/evaluators/HausdorffDistanceEvaluator.scala:25-36
    CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator   …/CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:27-79
    ModelPriorEvaluator                    api/sampling/evaluators/ModelPriorEvaluator.scala:24-31
    AcceptAllEvaluator                     api/sampling/evaluators/AcceptAllEvaluator.scala:22-28
    ModelSampling / TargetSampling / ModelAndTargetSampling    api/other/IcpProjectionDirection.scala:19-25
    ModelToTargetEvaluation / …            api/sampling/evaluators/EvaluationModeType.scala:20-26

Differences forced by the boundary: the Scalismo mesh decimations inside the reference's
constructors stay with the caller (here: `data.decimated_point_subset`, a deterministic stand-in), and the
standard normals of `posterior.sample()` are passed in explicitly (`propose(theta, z)`).
"""
from __future__ import annotations

import ctypes as C
import itertools
import dataclasses
import math

import numpy as np

from . import _native as nat
from . import data as _data

# api/other/IcpProjectionDirection.scala:19-25
ModelSampling, TargetSampling, ModelAndTargetSampling = "ModelSampling", "TargetSampling", "ModelAndTargetSampling"
# api/sampling/evaluators/EvaluationModeType.scala:20-26
ModelToTargetEvaluation, TargetToModelEvaluation, SymmetricEvaluation = 0, 1, 2


def _d(a):
    return a.ctypes.data_as(nat.c_double_p)


def _i(a):
    return a.ctypes.data_as(nat.c_int_p)


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


@dataclasses.dataclass(frozen=True)
class ModelFittingParameters:
    """Chain state: scale, pose (translation, Euler rotation, rotation centre), shape coefficients."""
    scale: float
    translation: tuple
    rotation: tuple
    rotation_center: tuple
    shape: tuple
    generatedBy: str = "Anonymous"

    @property
    def allParameters(self) -> np.ndarray:
        """ModelFittingParameters.scala:64 — [s | t | (phi,theta,psi) | centre | c]."""
        return np.asarray([self.scale, *self.translation, *self.rotation, *self.rotation_center, *self.shape],
                          dtype=np.float64)

    @classmethod
    def from_vector(cls, v, generatedBy="Anonymous"):
        v = np.asarray(v, dtype=np.float64)
        return cls(float(v[0]), tuple(v[1:4]), tuple(v[4:7]), tuple(v[7:10]), tuple(v[10:]), generatedBy)

    def copy_shape(self, shape, generatedBy):
        return dataclasses.replace(self, shape=tuple(np.asarray(shape, dtype=np.float64)), generatedBy=generatedBy)


def _theta(x) -> np.ndarray:
    return _f64(x.allParameters if isinstance(x, ModelFittingParameters) else x)


def initial_parameters(model) -> np.ndarray:
    """api/sampling/SamplingRegistration.scala:40-43: zero pose and shape, rotation centre = mean reference point."""
    theta = np.zeros(10 + model.rank)
    theta[0] = 1.0
    theta[7:10] = model.ref_points.sum(axis=0) * 1.0 / model.n_points
    return theta


_model_keys = itertools.count(1)


def _model_key(model) -> int:
    """icp_ctx_create_keyed's model_key: taken ONCE per model object — a content hash of the basis (xxhash, ≈ 10 ms for the face
    model's 137 MB) where that is importable, the object's number otherwise — so that the contexts of a batch registration (one per
    chain) do not each hash the basis again (6.6 ms per context).  The model's arrays are made read-only when the key is taken: a
    changed model is a new StatisticalMeshModel object (and a new key)."""
    arrays = (model.basis, model.variance, model.ref_points, model.mean_def, model.cells)
    where = tuple(a.__array_interface__["data"][0] for a in arrays)
    cached = getattr(model, "_icp_model_key", None)
    # (the key belongs to THESE arrays, frozen: a copy of the model object — copy.deepcopy carries the attribute along — or an array
    # swapped or made writeable again is hashed anew)
    if cached is not None and cached[1] == where and not any(a.flags.writeable for a in arrays):
        return cached[0]
    try:
        import xxhash
        hx = xxhash.xxh3_64()
        for arr in arrays[:4]:
            hx.update(np.ascontiguousarray(arr).view(np.uint8).reshape(-1).data)
        key = (hx.intdigest() & 0xFFFFFFFFFFFFFFFF) or 1
    except Exception:
        key = (next(_model_keys) << 20) | 0x5A5A5
    try:
        # the key vouches for these arrays (icp_ctx_create_keyed: "equal keys mean equal arrays"): an in-place edit after this
        # point would silently meet the stale device copy, so the arrays are frozen — a changed model is a new object
        for arr in arrays:
            arr.flags.writeable = False
        model._icp_model_key = (key, where)
    except Exception:
        pass
    return key


def expect_contexts(device: int, n_contexts: int) -> None:
    """icp_ctx_expect: a host about to make n_contexts contexts on `device` (−1: LOCAL_RANK) lets their streams be made ahead."""
    nat.check(nat.lib().icp_ctx_expect(int(device), int(n_contexts)), "icp_ctx_expect")


class IcpContext:
    """One StatisticalMeshModel + one target TriangleMesh3D resident on one MI355X (icp_ctx)."""

    def __init__(self, model, target, device: int = -1):
        self.model, self.target = model, target
        L = nat.lib()
        md = nat.ModelDesc(model.n_points, model.cells.shape[0], model.rank, _d(model.ref_points), _d(model.mean_def),
                           _d(model.basis), _d(model.variance), _i(model.cells))
        td = nat.MeshDesc(target.n_points, target.n_cells, _d(target.points), _i(target.cells))
        h = C.c_void_p()
        nat.check(L.icp_ctx_create_keyed(C.byref(md), C.byref(td), device, _model_key(model), C.byref(h)), "icp_ctx_create_keyed")
        self.h = h
        self.rank, self.N = model.rank, model.n_points
        self._children = []  # weak references to the proposals / evaluators / chains created on this context

    def setTarget(self, target):
        """icp_ctx_set_target: the same context (model data, scratch, streams) against another target mesh; every proposal, evaluator
        and chain made on it for the old target is closed first."""
        for ref in reversed(getattr(self, "_children", [])):
            child = ref()
            if child is not None and getattr(child, "h", None):
                child.close()
        self._children = []
        td = nat.MeshDesc(target.n_points, target.n_cells, _d(target.points), _i(target.cells))
        nat.check(nat.lib().icp_ctx_set_target(self.h, C.byref(td)), "icp_ctx_set_target")
        self.target = target
        return self

    def _adopt(self, child):
        import weakref
        self._children.append(weakref.ref(child))

    def close(self):
        """Destroys the context — after every proposal, evaluator and chain that still lives on it (their native objects hold
        device buffers, pinned memory and events of this context: closing the context first would leak them)."""
        if getattr(self, "h", None):
            for ref in reversed(getattr(self, "_children", [])):
                child = ref()
                if child is not None and getattr(child, "h", None):
                    child.close()
            self._children = []
            nat.lib().icp_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def runtime_stats(self) -> dict:
        """Fall-back counters of this context (icp_ctx_runtime_stats): all zero in a normal run."""
        return nat.runtime_stats(self.h)

    def step_paths(self) -> dict:
        """How many chain steps of this context took which path (icp_ctx_step_paths)."""
        return nat.step_paths(self.h)

    def profile_start(self, max_launches: int = 200000, count_searches: bool = False):
        """count_searches: also count the tests the searches execute (rows "count.*" of profile_stop; slows the filter launches
        down — for a short leg of its own, not for timing)."""
        nat.check(nat.lib().icp_ctx_profile_search_counters(self.h, int(count_searches)), "icp_ctx_profile_search_counters")
        nat.check(nat.lib().icp_ctx_profile_start(self.h, max_launches), "icp_ctx_profile_start")

    def profile_stop(self):
        """-> {kernel name: dict(calls, total_ms, avg_us, min_us, max_us)} measured with HIP events on the context stream."""
        stats = (nat.KernelStat * 64)()
        n = C.c_int32()
        nat.check(nat.lib().icp_ctx_profile_stop(self.h, stats, 64, C.byref(n)), "icp_ctx_profile_stop")
        out = {}
        for i in range(n.value):
            s = stats[i]
            out[s.name.decode()] = dict(calls=s.calls, total_ms=s.total_ms, avg_us=1e3 * s.total_ms / max(s.calls, 1),
                                        min_us=1e3 * s.min_ms, max_us=1e3 * s.max_ms)
        return out

    def setRotation(self, angles, R=None):
        """icp_ctx_set_rotation: use the caller's rotation matrix (row-major 3x3, e.g. Scalismo's Rotation(phi, theta, psi)) for every
        theta whose Euler angles equal `angles`; R=None withdraws it."""
        a = _f64(angles).reshape(3)
        m = _f64(R).reshape(9) if R is not None else None
        nat.check(nat.lib().icp_ctx_set_rotation(self.h, _d(a), _d(m) if m is not None else None), "icp_ctx_set_rotation")

    def rotationConvention(self) -> dict:
        """icp_ctx_rotation_convention: how many matrices handed to setRotation agreed with the library's Rz·Ry·Rx (to rounding) and
        how many did not; with mismatched == 0 the on-device loop takes mixtures with pose walks for this context."""
        v, m = C.c_int64(0), C.c_int64(0)
        nat.check(nat.lib().icp_ctx_rotation_convention(self.h, C.byref(v), C.byref(m)), "icp_ctx_rotation_convention")
        return {"verified": v.value, "mismatched": m.value}

    def transformedMesh(self, theta) -> np.ndarray:
        """ModelFittingParameters.transformedMesh (ModelFittingParameters.scala:108-110) -> points [N,3]."""
        th = _theta(theta)
        out = np.empty((self.N, 3))
        nat.check(nat.lib().icp_transformed_mesh(self.h, _d(th), _d(out)), "icp_transformed_mesh")
        return out

    def vertexNormals(self, theta) -> np.ndarray:
        th = _theta(theta)
        out = np.empty((self.N, 3))
        nat.check(nat.lib().icp_vertex_normals(self.h, _d(th), _d(out)), "icp_vertex_normals")
        return out

    def closestPointOnTarget(self, pts):
        q = _f64(pts).reshape(-1, 3)
        n = q.shape[0]
        cp, tri, d2 = np.empty((n, 3)), np.empty(n, dtype=np.int32), np.empty(n)
        nat.check(nat.lib().icp_closest_point_on_target(self.h, n, _d(q), _d(cp), _i(tri), _d(d2)), "icp_closest_point_on_target")
        return cp, tri, d2

    def closestTargetVertex(self, pts):
        q = _f64(pts).reshape(-1, 3)
        n = q.shape[0]
        idx, d2 = np.empty(n, dtype=np.int32), np.empty(n)
        nat.check(nat.lib().icp_closest_target_vertex(self.h, n, _d(q), _i(idx), _d(d2)), "icp_closest_target_vertex")
        return idx, d2

    def closestModelVertex(self, theta, pts):
        th, q = _theta(theta), _f64(pts).reshape(-1, 3)
        n = q.shape[0]
        idx, d2 = np.empty(n, dtype=np.int32), np.empty(n)
        nat.check(nat.lib().icp_closest_model_vertex(self.h, _d(th), n, _d(q), _i(idx), _d(d2)), "icp_closest_model_vertex")
        return idx, d2

    def closestPointOnModel(self, theta, pts):
        th, q = _theta(theta), _f64(pts).reshape(-1, 3)
        n = q.shape[0]
        cp, tri, d2 = np.empty((n, 3)), np.empty(n, dtype=np.int32), np.empty(n)
        nat.check(nat.lib().icp_closest_point_on_model(self.h, _d(th), n, _d(q), _d(cp), _i(tri), _d(d2)), "icp_closest_point_on_model")
        return cp, tri, d2


@dataclasses.dataclass
class IcpPosterior:
    corr_id: np.ndarray
    corr_aux: np.ndarray
    corr_point: np.ndarray
    keep: np.ndarray
    alpha: np.ndarray
    M: np.ndarray
    V: np.ndarray
    S: np.ndarray


class NonRigidIcpProposal:
    """NonRigidIcpProposal.scala:30-41.  `projectionDirection` is ModelSampling or TargetSampling."""

    def __init__(self, ctx: IcpContext, stepLength: float, tangentialNoise: float, noiseAlongNormal: float,
                 numOfSamplePoints: int, projectionDirection=ModelSampling, boundaryAware: bool = True,
                 generatedBy: str = "ShapeIcpProposal", decimatedTargetPoints=None, numDecimatedModelPoints=None):
        self.ctx, self.stepLength, self.generatedBy = ctx, stepLength, generatedBy
        self.projectionDirection = projectionDirection
        if projectionDirection == TargetSampling:
            # :46 target.operations.decimate(numOfSamplePoints) — the caller's decimation outcome
            tp = _f64(decimatedTargetPoints if decimatedTargetPoints is not None
                      else _data.decimated_point_subset(ctx.target, numOfSamplePoints)).reshape(-1, 3)
            prm = nat.ProposalParams(stepLength, tangentialNoise, noiseAlongNormal, 1, int(boundaryAware), 0, tp.shape[0], _d(tp))
        elif projectionDirection == ModelSampling:
            # :45 model.decimate(numOfSamplePoints): only its point COUNT matters (:94-96)
            k = int(numDecimatedModelPoints if numDecimatedModelPoints is not None else min(numOfSamplePoints, ctx.N))
            prm = nat.ProposalParams(stepLength, tangentialNoise, noiseAlongNormal, 0, int(boundaryAware), k, 0, None)
        else:
            raise ValueError("a NonRigidIcpProposal samples one direction; mix two for ModelAndTargetSampling "
                             "(MixedProposalDistributions.scala:52-65)")
        h = C.c_void_p()
        nat.check(nat.lib().icp_proposal_create(ctx.h, C.byref(prm), C.byref(h)), "icp_proposal_create")
        self.h = h
        self.K = nat.lib().icp_proposal_num_candidates(h)
        ctx._adopt(self)

    def setSampler(self, sampler: str = "eigen"):
        """icp_proposal_set_sampler: "eigen" (default; Scalismo's posterior.sample(), parity with the reference for a given z) or
        "cholesky-root" (opt-in: the same distribution from W = D·L⁻ᵀ, no eigen-decomposition; ranks <= 64)."""
        kind = {"eigen": 0, "cholesky-root": 1}[sampler]
        nat.check(nat.lib().icp_proposal_set_sampler(self.h, kind), "icp_proposal_set_sampler")
        return self

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            nat.lib().icp_proposal_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def propose(self, theta, z, return_correspondences: bool = False):
        """:53-68.  z = the r standard normals `posterior.sample()` draws (:55)."""
        th, z = _theta(theta), _f64(z)
        out = np.empty_like(th)
        corr = np.empty(max(self.K, 1), dtype=np.int32)
        nat.check(nat.lib().icp_proposal_propose(self.h, _d(th), _d(z), _d(out), _i(corr)), "icp_proposal_propose")
        res = out
        if isinstance(theta, ModelFittingParameters):
            res = theta.copy_shape(out[10:], self.generatedBy)
        return (res, corr[:self.K]) if return_correspondences else res

    def logTransitionProbability(self, theta_from, theta_to) -> float:
        """:71-85 (−inf when anything but the shape differs)."""
        a, b = _theta(theta_from), _theta(theta_to)
        out = C.c_double()
        nat.check(nat.lib().icp_proposal_log_transition(self.h, _d(a), _d(b), C.byref(out)), "icp_proposal_log_transition")
        return out.value

    def icpPosterior(self, theta, with_aux: bool = True) -> IcpPosterior:
        """:88-153, diagnostic view."""
        th = _theta(theta)
        r, K = self.ctx.rank, max(self.K, 1)
        p = IcpPosterior(np.empty(K, dtype=np.int32), np.empty(K, dtype=np.int32), np.empty((K, 3)),
                         np.empty(K, dtype=np.uint8), np.empty(r), np.empty((r, r)), np.empty((r, r)), np.empty(r))
        view = nat.PosteriorView(0, _i(p.corr_id), _i(p.corr_aux) if with_aux else None, _d(p.corr_point),
                                 p.keep.ctypes.data_as(nat.c_ubyte_p), _d(p.alpha), _d(p.M), _d(p.V), _d(p.S))
        nat.check(nat.lib().icp_proposal_posterior(self.h, _d(th), C.byref(view)), "icp_proposal_posterior")
        k = view.n_candidates
        p.corr_id, p.corr_aux, p.corr_point, p.keep = p.corr_id[:k], p.corr_aux[:k], p.corr_point[:k], p.keep[:k]
        if not with_aux:
            p.corr_aux = np.full(k, -1, dtype=np.int32)
        return p


class _Evaluator:
    def __init__(self, ctx: IcpContext, kind, mode, n_model_ids, target_pts, gauss_mean, gauss_sigma, exp_rate):
        self.ctx = ctx
        tp = _f64(target_pts).reshape(-1, 3) if target_pts is not None else np.zeros((0, 3))
        prm = nat.EvaluatorParams(kind, mode, int(n_model_ids), tp.shape[0], _d(tp), gauss_mean, gauss_sigma, exp_rate)
        h = C.c_void_p()
        nat.check(nat.lib().icp_evaluator_create(ctx.h, C.byref(prm), C.byref(h)), "icp_evaluator_create")
        self.h = h
        ctx._adopt(self)

    def close(self):
        if getattr(self, "h", None) and getattr(self.ctx, "h", None):
            nat.lib().icp_evaluator_destroy(self.h)
        self.h = None

    def __del__(self):
        self.close()

    def logValue(self, sample, return_aux: bool = False):
        th = _theta(sample)
        out = C.c_double()
        aux = np.zeros(4)
        nat.check(nat.lib().icp_evaluator_log_value(self.h, _d(th), C.byref(out), _d(aux)), "icp_evaluator_log_value")
        return (out.value, aux) if return_aux else out.value

    def bindChain(self, proposals):
        """icp_chain_bind: this evaluator and the chain's ICP proposals (in the mixture's order) form ONE Metropolis–Hastings chain
        driven method by method (Scalismo's MetropolisHastings.next, api/sampling/SamplingRegistration.scala:52-58): the first call of
        a step submits the whole step, the calls behind it find their values parked.  An empty list unbinds."""
        n = len(proposals)
        arr = (C.c_void_p * max(n, 1))(*[p.h for p in proposals])
        nat.check(nat.lib().icp_chain_bind(self.h, n, arr), "icp_chain_bind")

    def bindStats(self) -> dict:
        out = (C.c_int64 * 3)()
        nat.check(nat.lib().icp_chain_bind_stats(self.h, out), "icp_chain_bind_stats")
        return {"steps_from_propose": int(out[0]), "steps_from_log_value": int(out[1]), "parked_transition_hits": int(out[2])}


class AcceptAllEvaluator:
    """api/sampling/evaluators/AcceptAllEvaluator.scala:22-28 — logValue is the constant 0.0 whatever the sample (every proposal
    whose transition ratio allows it is accepted); no native call."""

    def logValue(self, sample) -> float:
        return 0.0


def _sides(ctx, numberOfPointsForComparison, decimatedTargetPoints, numDecimatedModelPoints):
    tp = (decimatedTargetPoints if decimatedTargetPoints is not None
          else _data.decimated_point_subset(ctx.target, numberOfPointsForComparison))
    k = int(numDecimatedModelPoints if numDecimatedModelPoints is not None else min(numberOfPointsForComparison, ctx.N))
    return k, tp


class IndependentPointDistanceEvaluator(_Evaluator):
    """IndependentPointDistanceEvaluator.scala:27-31; likelihoodModel = breeze Gaussian(mean, sigma)
    (ProductEvaluators.scala:39 uses Gaussian(0, uncertainty))."""

    def __init__(self, ctx, likelihoodMean: float, likelihoodSigma: float, evaluationMode, numberOfPointsForComparison: int,
                 decimatedTargetPoints=None, numDecimatedModelPoints=None):
        k, tp = _sides(ctx, numberOfPointsForComparison, decimatedTargetPoints, numDecimatedModelPoints)
        super().__init__(ctx, 0, evaluationMode, k, tp, likelihoodMean, likelihoodSigma, 1.0)


class HausdorffDistanceEvaluator(_Evaluator):
    """HausdorffDistanceEvaluator.scala:25-28; likelihoodModel = breeze Exponential(rate) (ProductEvaluators.scala:58)."""

    def __init__(self, ctx, likelihoodRate: float):
        super().__init__(ctx, 1, 2, 0, None, 0.0, 1.0, likelihoodRate)


class CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator(_Evaluator):
    """CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:27-32; likelihoodModelAvg = Gaussian(mean, sigma),
    likelihoodModelMax = Exponential(rate) (ProductEvaluators.scala:77-78)."""

    def __init__(self, ctx, avgMean: float, avgSigma: float, maxRate: float, evaluationMode, numberOfPointsForComparison: int,
                 decimatedTargetPoints=None, numDecimatedModelPoints=None):
        k, tp = _sides(ctx, numberOfPointsForComparison, decimatedTargetPoints, numDecimatedModelPoints)
        super().__init__(ctx, 2, evaluationMode, k, tp, avgMean, avgSigma, maxRate)


class ModelPriorEvaluator:
    """ModelPriorEvaluator.scala:24-31 — MultivariateNormalDistribution(0, I_rank).logpdf(shape coefficients)."""

    def __init__(self, rank: int):
        self.rank = rank

    def logValue(self, theta) -> float:
        th = _theta(theta)
        out = C.c_double()
        nat.check(nat.lib().icp_prior_log_value(self.rank, _d(th), C.byref(out)), "icp_prior_log_value")
        return out.value


def chain_eval_step(evaluator: _Evaluator, proposals, theta_cur, theta_prop):
    """icp_chain_eval_step: likelihood of theta_prop + forward/backward transition log-densities of every proposal,
    one device submission, one synchronisation."""
    a, b = _theta(theta_cur), _theta(theta_prop)
    n = len(proposals)
    arr = (C.c_void_p * max(n, 1))(*[p.h for p in proposals])
    val = C.c_double()
    fwd, bwd = np.zeros(max(n, 1)), np.zeros(max(n, 1))
    nat.check(nat.lib().icp_chain_eval_step(evaluator.h, n, arr, _d(a), _d(b), C.byref(val), _d(fwd), _d(bwd)),
              "icp_chain_eval_step")
    return val.value, fwd[:n], bwd[:n]


def chain_step(evaluator: _Evaluator, proposals, theta_cur, generator: int = -1, z=None, theta_prop=None):
    """icp_chain_step: propose from proposals[generator] (or take theta_prop as given when generator < 0) AND evaluate the
    proposal — likelihood + forward/backward transition log-densities of every proposal — in one device submission.
    Returns (theta_prop, log_value, fwd, bwd)."""
    a = _theta(theta_cur)
    n = len(proposals)
    arr = (C.c_void_p * max(n, 1))(*[p.h for p in proposals])
    if generator >= 0:
        zz = np.ascontiguousarray(z, dtype=np.float64)
        b = np.zeros_like(a)
    else:
        zz = np.zeros(1)
        b = _theta(theta_prop).copy()
    val = C.c_double()
    fwd, bwd = np.zeros(max(n, 1)), np.zeros(max(n, 1))
    nat.check(nat.lib().icp_chain_step(evaluator.h, n, arr, int(generator), _d(a), _d(zz), _d(b), C.byref(val), _d(fwd), _d(bwd)),
              "icp_chain_step")
    return b, val.value, fwd[:n], bwd[:n]


def chain_step_batched(evaluators, proposals, theta_cur, generator, z=None, theta_prop=None):
    """icp_chain_step_batched: chain_step for B independent chains (one context each) in one sequence of launches.
    evaluators[b], proposals[b] (list of n_props), theta_cur[b], generator[b]; z[b] where generator[b] >= 0, theta_prop[b]
    where generator[b] < 0.  Returns (theta_prop [B, 10+r], log_value [B], fwd [B, n], bwd [B, n], status [B])."""
    B = len(evaluators)
    n = len(proposals[0])
    cur = [_theta(t) for t in theta_cur]
    P = cur[0].shape[0]
    out = np.zeros((B, P))
    zs = []
    for b in range(B):
        if generator[b] >= 0:
            zs.append(np.ascontiguousarray(z[b], dtype=np.float64))
        else:
            zs.append(np.zeros(1))
            out[b] = _theta(theta_prop[b])
    ev = (C.c_void_p * B)(*[e.h for e in evaluators])
    pr = (C.c_void_p * max(B * n, 1))(*[p.h for ps in proposals for p in ps])
    gen = (C.c_int32 * B)(*[int(g) for g in generator])
    dp = nat.c_double_p
    curp = (dp * B)(*[_d(t) for t in cur])
    zp = (dp * B)(*[_d(t) for t in zs])
    outp = (dp * B)(*[_d(out[b]) for b in range(B)])
    val, fwd, bwd = np.zeros(B), np.zeros((B, max(n, 1))), np.zeros((B, max(n, 1)))
    status = (C.c_int32 * B)()
    rc = nat.lib().icp_chain_step_batched(B, ev, n, pr, gen, curp, zp, outp, _d(val), _d(fwd), _d(bwd), status)
    nat.check(rc, "icp_chain_step_batched")
    return out, val, fwd[:, :n], bwd[:, :n], np.array(list(status), dtype=np.int32)


class BatchedStepTicket:
    """icp_chain_step_batched_issue: a batch in flight.  collect() waits for it and returns what chain_step_batched returns;
    abandon() waits for its launches and drops the step.  Until then the member contexts refuse every other call (ICP_ERR_BUSY)."""

    def __init__(self, evaluators, proposals, theta_cur, generator, z=None, theta_prop=None, launch_ctx=None):
        B = len(evaluators)
        n = len(proposals[0])
        self._n = n
        self._keep = cur = [_theta(t) for t in theta_cur]
        self.out = np.zeros((B, cur[0].shape[0]))
        zs = []
        for b in range(B):
            if generator[b] >= 0:
                zs.append(np.ascontiguousarray(z[b], dtype=np.float64))
            else:
                zs.append(np.zeros(1))
                self.out[b] = _theta(theta_prop[b])
        self._zs = zs
        ev = (C.c_void_p * B)(*[e.h for e in evaluators])
        pr = (C.c_void_p * max(B * n, 1))(*[p.h for ps in proposals for p in ps])
        gen = (C.c_int32 * B)(*[int(g) for g in generator])
        dp = nat.c_double_p
        curp = (dp * B)(*[_d(t) for t in cur])
        zp = (dp * B)(*[_d(t) for t in zs])
        outp = (dp * B)(*[_d(self.out[b]) for b in range(B)])
        self.val, self.fwd, self.bwd = np.zeros(B), np.zeros((B, max(n, 1))), np.zeros((B, max(n, 1)))
        self.status = (C.c_int32 * B)()
        self._h = C.c_void_p()
        rc = nat.lib().icp_chain_step_batched_issue(B, ev, n, pr, gen, curp, zp, outp, _d(self.val), _d(self.fwd), _d(self.bwd), self.status,
                                                    launch_ctx.h if launch_ctx is not None else None, C.byref(self._h))
        nat.check(rc, "icp_chain_step_batched_issue")

    def collect(self):
        h, self._h = self._h, None
        nat.check(nat.lib().icp_chain_step_batched_collect(h), "icp_chain_step_batched_collect")
        return self.out, self.val, self.fwd[:, :self._n], self.bwd[:, :self._n], np.array(list(self.status), dtype=np.int32)

    def abandon(self):
        h, self._h = self._h, None
        if h:
            nat.check(nat.lib().icp_chain_step_batched_abandon(h), "icp_chain_step_batched_abandon")


def chain_step_prelaunch(evaluator: _Evaluator, proposals, theta_cur, generator: int = -1, z=None, theta_prop=None):
    """icp_chain_step_prelaunch: issue the first launches of the step that a later chain_step with exactly these arguments
    will ask for (typically: the next step under the assumption that the step in flight is rejected).  Never changes
    results; proposals == [] drops a pending half step."""
    n = len(proposals)
    if n == 0:
        nat.check(nat.lib().icp_chain_step_prelaunch(evaluator.h, 0, None, -1, None, None), "icp_chain_step_prelaunch")
        return
    a = _theta(theta_cur)
    arr = (C.c_void_p * n)(*[p.h for p in proposals])
    key = np.ascontiguousarray(z, dtype=np.float64) if generator >= 0 else _theta(theta_prop)
    nat.check(nat.lib().icp_chain_step_prelaunch(evaluator.h, n, arr, int(generator), _d(a), _d(key)), "icp_chain_step_prelaunch")


class IcpBasedSurfaceFitting:
    """api/other/IcpBasedSurfaceFitting.scala:32 — the deterministic non-rigid ICP baseline (posterior MEAN, isotropic noise).
    `modelPointIds` / `targetPointSamples` stand for the UniformMeshSampler3D draws of :51-53 (made by the caller)."""

    def __init__(self, ctx: IcpContext, stepLength: float = 1.0, projectionDirection=ModelSampling, modelPointIds=None,
                 targetPointSamples=None):
        self.ctx, self.step = ctx, float(stepLength)
        self.direction = 1 if projectionDirection in (TargetSampling, "TargetSampling", 1) else 0
        self.ids = np.ascontiguousarray(modelPointIds if modelPointIds is not None else np.zeros(0), dtype=np.int32)
        self.tp = _f64(targetPointSamples if targetPointSamples is not None else np.zeros((0, 3))).reshape(-1, 3)

    def runfitting(self, numIterations: int, iterationSeq=(1.0, 0.1, 0.01), initialModelParameters=None) -> np.ndarray:
        """:46-126; returns the final parameter vector (the reference returns the corresponding mesh: ctx.transformedMesh)."""
        th = _theta(initialModelParameters if initialModelParameters is not None else initial_parameters(self.ctx.model))
        sig = _f64(iterationSeq)
        out = np.zeros_like(th)
        fp = nat.FitParams(self.direction, self.ids.shape[0], _i(self.ids), self.tp.shape[0], _d(self.tp), self.step)
        nat.check(nat.lib().icp_fit_deterministic(self.ctx.h, C.byref(fp), _d(th), int(numIterations), sig.shape[0], _d(sig), _d(out)),
                  "icp_fit_deterministic")
        return out


def posterior_variability(ctx: IcpContext, thetas, mode: int = 0, theta_ref=None) -> np.ndarray:
    """apps/util/PosteriorVariability.scala:30-73 over logged chain states: per-vertex total variance (mode 0), variance along the
    normals of theta_ref's mesh (1) or along the mean sample normal (2)."""
    th = _f64(thetas).reshape(-1, 10 + ctx.rank)
    ref = _theta(theta_ref if theta_ref is not None else th[0])
    out = np.zeros(ctx.model.n_points)
    nat.check(nat.lib().icp_posterior_variability(ctx.h, th.shape[0], _d(th), int(mode), _d(ref), _d(out)), "icp_posterior_variability")
    return out


def evaluate_reconstruction_to_ground_truth(ctx: IcpContext, theta) -> dict:
    """api/other/RegistrationComparison.scala:24-49 for the mesh of theta against the context's target."""
    out = np.zeros(5)
    nat.check(nat.lib().icp_mesh_metrics(ctx.h, _d(_theta(theta)), _d(out)), "icp_mesh_metrics")
    return {"average2surface": out[0], "hausdorff": out[1], "average2surface_boundary_aware": out[2], "max_boundary_aware": out[3],
            "kept": int(out[4])}

