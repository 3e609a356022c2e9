/*
 * icp_proposal.h — C ABI of the MI355X-native closest-point-proposal path (libicp_proposal_amd.so).
 *
 * This is the drop-in boundary for ONE hot path of unibas-gravis/icp-proposal: the three Scalismo plug-in
 * methods the Metropolis–Hastings chain calls per step,
 *
 *     ProposalGenerator[ModelFittingParameters].propose(current)
 *     TransitionProbability[ModelFittingParameters].logTransitionProbability(from, to)
 *     DistributionEvaluator[ModelFittingParameters].logValue(sample)
 *
 * as implemented by the reference classes cited at each entry point below (paths relative to the reference's
 * src/main/scala/).  The reference has no FFI of its own (pure Scala on Scalismo); INTEGRATION.md shows the
 * JNI stub + Scala adapters a maintainer would add to bind these symbols.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE double, all ids int32; arrays are row-major.
 *   - theta = ModelFittingParameters.allParameters (ModelFittingParameters.scala:64):
 *       [ s | tx ty tz | phi theta psi | cx cy cz | c_0 .. c_{r-1} ]      (10 + rank doubles)
 *     pose = translation ∘ rotation(phi,theta,psi about c) with R = Rz(phi)·Ry(theta)·Rx(psi)
 *     (ModelFittingParameters.scala:79-86), scale applied last (:88-106).
 *   - every function returns an icp_status (0 = ok, < 0 = error); outputs are written into caller-owned buffers;
 *     the library copies model/target to the GPU once at icp_ctx_create and keeps no pointer to caller memory.
 *   - -inf is a VALID value of icp_proposal_log_transition (NonRigidIcpProposal.scala:72-74).
 *   - entry points are thread-safe (calls on one context are serialised internally); the random numbers of
 *     posterior.sample() (NonRigidIcpProposal.scala:55) are drawn by the CALLER and passed in as z.
 *   - there is no CPU fallback: if no HIP device is usable, icp_ctx_create fails with ICP_ERR_DEVICE.
 */
#ifndef ICP_PROPOSAL_H
#define ICP_PROPOSAL_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ICP_API __attribute__((visibility("default")))

typedef struct icp_ctx icp_ctx;             /* one statistical mesh model + one target mesh, resident on one GPU */
typedef struct icp_proposal icp_proposal;   /* api/sampling/proposals/NonRigidIcpProposal.scala:30-41 */
typedef struct icp_evaluator icp_evaluator; /* the classes under api/sampling/evaluators/ */

typedef enum {
  ICP_OK = 0,
  ICP_ERR_INVALID_ARG = -1, /* null pointer, negative size, id out of range, unknown enum */
  ICP_ERR_DEVICE = -2,      /* HIP runtime / no device / out of memory (icp_last_error() has the HIP string) */
  ICP_ERR_NOT_FINITE = -3,  /* a result is NaN (the Scalismo chain throws on NaN transition probabilities) */
  ICP_ERR_NOT_SPD = -4,     /* a normal-equation matrix failed to factor */
  ICP_ERR_EMPTY = -5,       /* boundary-aware evaluator dropped every point (reference: empty .max throws) */
  ICP_ERR_BUSY = -6         /* the context is part of a batch between icp_chain_step_batched_issue and _collect / _abandon */
} icp_status;

/* api/other/IcpProjectionDirection.scala:19-25.  ModelAndTargetSampling is not a proposal direction: the
 * reference builds TWO proposals and mixes them (api/sampling/MixedProposalDistributions.scala:52-65). */
typedef enum { ICP_MODEL_SAMPLING = 0, ICP_TARGET_SAMPLING = 1 } icp_direction;

/* api/sampling/evaluators/EvaluationModeType.scala:20-26 */
typedef enum { ICP_MODEL_TO_TARGET = 0, ICP_TARGET_TO_MODEL = 1, ICP_SYMMETRIC = 2 } icp_eval_mode;

typedef enum {
  ICP_EVAL_INDEPENDENT_POINT_DISTANCE = 0, /* evaluators/IndependentPointDistanceEvaluator.scala:27-67 */
  ICP_EVAL_HAUSDORFF = 1,                  /* evaluators/HausdorffDistanceEvaluator.scala:25-36 */
  ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE = 2 /* evaluators/CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:27-79 */
} icp_eval_kind;

/* What crosses the boundary of a Scalismo StatisticalMeshModel (Statismo layout). */
typedef struct {
  int32_t n_points;               /* N */
  int32_t n_triangles;            /* T */
  int32_t rank;                   /* r */
  const double *ref_points;       /* [N*3]  reference mesh vertices */
  const double *mean_deformation; /* [N*3]  GP mean at the reference vertices (NULL = zero) */
  const double *basis;            /* [3N*r] UNSCALED eigenfunctions, row 3i+d = vertex i axis d */
  const double *variance;         /* [r]    eigenvalues */
  const int32_t *triangles;       /* [T*3] */
} icp_model_desc;

typedef struct {
  int32_t n_points;
  int32_t n_triangles;
  const double *points;     /* [M*3] */
  const int32_t *triangles; /* [Tt*3] */
} icp_mesh_desc;

/* Constructor arguments of NonRigidIcpProposal (NonRigidIcpProposal.scala:30-41).  The two decimations at
 * :45-46 stay with the caller (Scalismo): only their outcome crosses the boundary — the COUNT of points of the
 * decimated model (the reference uses ids 0 until K of the full mesh, :94-96) and the POINTS of the decimated
 * target (:117). */
typedef struct {
  double step_length;        /* :33 */
  double tangential_noise;   /* :34 stddev in the tangent plane */
  double noise_along_normal; /* :35 stddev along the vertex normal */
  int32_t direction;         /* :37 icp_direction */
  int32_t boundary_aware;    /* :38 */
  int32_t n_model_ids;       /* ModelSampling: number of points of model.decimate(numOfSamplePoints) */
  int32_t n_target_points;   /* TargetSampling: number of points of target.operations.decimate(numOfSamplePoints) */
  const double *target_points; /* [n_target_points*3] */
} icp_proposal_params;

/* Constructor arguments of the three likelihood evaluators; the likelihood distributions are the Breeze
 * objects built in api/sampling/ProductEvaluators.scala:39,58,77-78. */
typedef struct {
  int32_t kind;            /* icp_eval_kind */
  int32_t mode;            /* icp_eval_mode (ignored by ICP_EVAL_HAUSDORFF) */
  int32_t n_model_ids;     /* number of points of model.decimate(numberOfPointsForComparison); ids 0 until K */
  int32_t n_target_points; /* points of targetMesh.operations.decimate(numberOfPointsForComparison) */
  const double *target_points;
  double gauss_mean;       /* Gaussian(mean, sigma): kind 0 (per-point distance) and kind 2 (average distance) */
  double gauss_sigma;
  double exp_rate;         /* Exponential(rate): kind 1 (Hausdorff distance) and kind 2 (max distance) */
} icp_evaluator_params;

/* Diagnostic view of one ICP posterior (NonRigidIcpProposal.scala:88-153); any pointer may be NULL. */
typedef struct {
  int32_t n_candidates;   /* out: K (before the boundary filter) */
  int32_t *corr_id;       /* [K]   model vertex id of each correspondence — THE correspondence index (:118 / :94) */
  int32_t *corr_aux;      /* [K]   ModelSampling: target vertex nearest to the surface point (:98), else -1 */
  double *corr_point;     /* [K*3] target-side point (:97 / :117) */
  uint8_t *keep;          /* [K]   1 = survives the boundary filter (:104 / :124) */
  double *alpha;          /* [r]   posterior mean coefficients */
  double *M;              /* [r*r] I + sum_i Q_i^T Sigma_i^-1 Q_i */
  double *V;              /* [r*r] eigenvectors (columns) of D M^-1 D — the posterior KL basis is Phi·V */
  double *S;              /* [r]   its eigenvalues, descending */
} icp_posterior_view;

/* ---------------------------------------------------------------- context */

/* device: HIP ordinal, or -1 = use LOCAL_RANK from the environment (0 if unset). */
ICP_API int icp_ctx_create(const icp_model_desc *model, const icp_mesh_desc *target, int device, icp_ctx **out);
/* The same for a caller that makes MANY contexts from one model (one per chain of a batch registration: SURVEY.md §8e).  Contexts of a
 * device made from the same model share its device data; icp_ctx_create recognises the model by hashing its arrays — 137 MB of basis at
 * the face model's size, 6.6 ms per context.  model_key != 0: the caller vouches that equal keys mean equal model arrays (an object
 * id, a hash taken once); the library then hashes the key, the small arrays and a sample of the basis only (0.3 ms), and while the FIRST
 * context of a keyed model does its one-off host work, the streams of the contexts to come can be made by a helper thread
 * (icp_ctx_expect; hipStreamCreateWithPriority: 2.4 ms each; ICP_NO_STREAM_PREWARM=1: not).  0 = icp_ctx_create. */
ICP_API int icp_ctx_create_keyed(const icp_model_desc *model, const icp_mesh_desc *target, int device, uint64_t model_key, icp_ctx **out);
/* Optional hint of a host that is about to make n_contexts contexts on `device` (one per chain of a batch registration, one per chain
 * thread of a `.par` experiment): their streams — three quarters of what a further context costs — are made ahead by a helper thread
 * while the caller's first icp_ctx_create_keyed does the model's one-off host work.  Never changes results; at most 64 are made; a
 * host with one chain per GPU simply does not call it (until round 6 every first keyed context made two dozen unasked). */
ICP_API int icp_ctx_expect(int device, int32_t n_contexts);
ICP_API void icp_ctx_destroy(icp_ctx *ctx);
/* Gives the context ANOTHER target mesh and keeps everything that does not depend on the target — the model's device data, the
 * per-chain scratch, streams, pinned buffers: a batch registration (one statistical model against many targets:
 * apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala:106-163) makes its contexts once.  The context must have no proposal and no
 * evaluator at the time (they are made for one target: destroy them first, create new ones afterwards); what it had cached against the
 * old target is dropped.  Creating a context costs 20+ ms at the face model's size, this call a fraction of a millisecond once the
 * target's own device data exists. */
ICP_API int icp_ctx_set_target(icp_ctx *ctx, const icp_mesh_desc *target);
ICP_API const char *icp_status_string(int status);
ICP_API const char *icp_last_error(void); /* thread-local detail of the last failing call */
ICP_API int icp_ctx_rank(const icp_ctx *ctx);
ICP_API int icp_ctx_device(const icp_ctx *ctx);

/* Pose across the boundary.  theta carries the three Euler angles; the reference turns them into a rotation with
 * Scalismo's Rotation(phi, theta, psi, centre) (ModelFittingParameters.scala:79-86).  A caller that wants Scalismo's OWN matrix
 * used — whatever its convention — registers it for the triple before passing a theta with these angles: R = row-major 3x3 rotation
 * (checked: orthonormal, determinant +1); R == NULL withdraws the entry.  Up to 32 triples are remembered (least recently used
 * out); a theta whose angles have no entry is posed with the library's Rz(phi)·Ry(theta)·Rx(psi).  Whenever the matrix in force
 * for a triple changes — first registration, replacement by a different matrix, withdrawal, eviction — everything the context has
 * cached under thetas with these angles (instances, posteriors of every proposal, memoised likelihood values) is dropped, so a
 * later call recomputes it with the new matrix; registering before the first use of a triple costs nothing. */
ICP_API int icp_ctx_set_rotation(icp_ctx *ctx, const double angles[3], const double R[9]);
/* The convention check behind it: every matrix handed to icp_ctx_set_rotation is compared with the library's own
 * Rz(phi)·Ry(theta)·Rx(psi) for the same angles (include/icp_sincos.h).  *verified counts the matrices that agreed to rounding
 * (2e-15 per entry), *mismatched the ones that did not (either may be NULL).  A host with mismatched == 0 — the Scala adapter, if
 * Scalismo's Rotation(phi, theta, psi, centre) is Rz·Ry·Rx as SURVEY App. B8 assumes — may run mixtures WITH pose walks through
 * icp_chains_run_on_device, where the proposed pose's matrix is made on the device; one mismatch closes that path for the context
 * (ICP_ERR_INVALID_ARG there; every host-stepped entry point keeps using the caller's matrices as before). */
ICP_API int icp_ctx_rotation_convention(icp_ctx *ctx, int64_t *verified, int64_t *mismatched);

/* ModelFittingParameters.transformedMesh (ModelFittingParameters.scala:108-110): points_out [N*3]. */
ICP_API int icp_transformed_mesh(icp_ctx *ctx, const double *theta, double *points_out);
/* vertex normals of that mesh (Scalismo vertexNormals, used at NonRigidIcpProposal.scala:100,120): [N*3]. */
ICP_API int icp_vertex_normals(icp_ctx *ctx, const double *theta, double *normals_out);

/* The two brute-force searches, exposed for parity tests and micro-benchmarks.
 * icp_closest_point_on_target : target.operations.closestPointOnSurface(p).point (NonRigidIcpProposal.scala:97);
 * icp_closest_target_vertex   : target.pointSet.findClosestPoint(p).id            (:98);
 * icp_closest_model_vertex    : currentMesh.pointSet.findClosestPoint(p).id for the mesh of theta (:118);
 * icp_closest_point_on_model  : modelSample.operations.closestPointOnSurface(p) (IndependentPointDistanceEvaluator.scala:51).
 * Ties: lowest squared distance, then lowest index.  Any output pointer may be NULL. */
ICP_API int icp_closest_point_on_target(icp_ctx *ctx, int32_t n, const double *queries, double *points_out,
                                        int32_t *triangle_out, double *dist2_out);
ICP_API int icp_closest_target_vertex(icp_ctx *ctx, int32_t n, const double *queries, int32_t *id_out, double *dist2_out);
ICP_API int icp_closest_model_vertex(icp_ctx *ctx, const double *theta, int32_t n, const double *queries,
                                     int32_t *id_out, double *dist2_out);
ICP_API int icp_closest_point_on_model(icp_ctx *ctx, const double *theta, int32_t n, const double *queries,
                                       double *points_out, int32_t *triangle_out, double *dist2_out);

/* ---------------------------------------------------------------- NonRigidIcpProposal */

ICP_API int icp_proposal_create(icp_ctx *ctx, const icp_proposal_params *params, icp_proposal **out);
ICP_API void icp_proposal_destroy(icp_proposal *p);

/* propose (NonRigidIcpProposal.scala:53-68).  z[r] = the standard normals posterior.sample() draws (:55);
 * theta_out[10+r] = theta with the shape coefficients replaced (:61-66).  corr_id_out (optional, [K]) receives
 * the correspondence indices of the posterior that was used, -1 where the boundary filter dropped one. */
ICP_API int icp_proposal_propose(icp_proposal *p, const double *theta, const double *z, double *theta_out,
                                 int32_t *corr_id_out);

/* logTransitionProbability(from, to) (NonRigidIcpProposal.scala:71-85). */
ICP_API int icp_proposal_log_transition(icp_proposal *p, const double *theta_from, const double *theta_to, double *out);

/* Opt-in, NOT the reference's arithmetic: how propose() turns the caller's standard normals z into a sample of the posterior.
 *   ICP_SAMPLER_EIGEN (default)   z multiplies the posterior's KL basis, D M^-1 D = V S V^T (what Scalismo's posterior.sample() does,
 *                                 NonRigidIcpProposal.scala:55): parity with the reference for a given z; needs the eigen-decomposition
 *                                 of every accepted state's posterior, the longest link of an accepted step.
 *   ICP_SAMPLER_CHOLESKY_ROOT     z multiplies W = D L^-T (M = L L^T): W W^T = D M^-1 D as well, so the sample has the SAME distribution
 *                                 and logTransitionProbability (which does not depend on the root) is unchanged — the chain
 *                                 is a different realisation of the same Markov kernel.  No eigen-decomposition at all: D^-1 W z = L^-T z
 *                                 is one back substitution per proposal.  The diagnostic view then returns V = L (row-major lower
 *                                 triangle) and S = 1 / diag(L).  Ranks <= 256 (up to 64: a kernel of its own where the decomposition
 *                                 would run; above: the posterior's factorisation hands its factor out).
 * Call before the proposal's first use, or any time: posteriors already decomposed the other way are decomposed again. */
typedef enum { ICP_SAMPLER_EIGEN = 0, ICP_SAMPLER_CHOLESKY_ROOT = 1 } icp_sampler;
ICP_API int icp_proposal_set_sampler(icp_proposal *p, int32_t sampler);

/* icpPosterior(theta) (NonRigidIcpProposal.scala:88-153), diagnostic. */
ICP_API int icp_proposal_posterior(icp_proposal *p, const double *theta, icp_posterior_view *view);
ICP_API int icp_proposal_num_candidates(const icp_proposal *p);

/* ---------------------------------------------------------------- evaluators */

ICP_API int icp_evaluator_create(icp_ctx *ctx, const icp_evaluator_params *params, icp_evaluator **out);
ICP_API void icp_evaluator_destroy(icp_evaluator *e);

/* logValue(sample) of the likelihood evaluator (computeLogValue of the three evaluator classes).
 * aux (optional, [4]): kind 0 -> {modelToTarget sum, targetToModel sum, 0, 0}; kind 1 -> {hausdorff, m2t max, t2m max, 0};
 * kind 2 -> {avg, max, n kept model->target, n kept target->model}. */
ICP_API int icp_evaluator_log_value(icp_evaluator *e, const double *theta, double *out, double *aux);

/* ModelPriorEvaluator.logValue (evaluators/ModelPriorEvaluator.scala:24-31): O(r) host arithmetic. */
ICP_API int icp_prior_log_value(int32_t rank, const double *theta, double *out);

/* ---------------------------------------------------------------- deterministic non-rigid ICP
 * IcpBasedSurfaceFitting.runfitting (api/other/IcpBasedSurfaceFitting.scala:46-126), the paper's comparison baseline: for every
 * sigma2 of the sequence (:36-40: 1, 0.1, 0.01) the recursion (:55-104) runs numIterations + 1 times: instance, correspondences
 * in one direction (:71-79), GP regression with ISOTROPIC noise sigma2 (:81), posterior MEAN (:82), its coefficients (:84), step
 * (:85).  The sample ids / sample points are drawn by Scalismo's UniformMeshSampler3D in the reference (:51-53) and are inputs
 * here; so is the per-iteration direction draw of ModelAndTargetSampling (:66-69: call once per iteration instead).  All
 * iterations run on the device without host round trips.  theta_out = theta_init with the fitted shape coefficients. */
typedef struct {
  int32_t direction;             /* icp_direction */
  int32_t n_model_ids;           /* ModelSampling: pointIds (:53), any ids, repeats allowed */
  const int32_t *model_ids;
  int32_t n_target_points;       /* TargetSampling: targetPointSamples (:51) */
  const double *target_points;
  double step_length;            /* :32 (default 1.0) */
} icp_fit_params;
ICP_API int icp_fit_deterministic(icp_ctx *ctx, const icp_fit_params *params, const double *theta_init, int32_t n_iterations,
                                  int32_t n_sigma, const double *sigma2_seq, double *theta_out);

/* ---------------------------------------------------------------- posterior variability maps
 * apps/util/PosteriorVariability.scala:30-73 over n_samples logged chain states (thetas [n_samples*(10+r)]):
 *   mode 0: trace of the per-vertex sample covariance (computeDistanceMapFromMeshesTotal :30-49);
 *   mode 1: variance along the unit vertex normals of the mesh of theta_ref (computeDistanceMapFromMeshesNormal, sumNormals = false);
 *   mode 2: variance along the mean (not renormalised) of the samples' unit vertex normals (sumNormals = true, :63-65).
 * out [N].  n_samples >= 2. */
ICP_API int icp_posterior_variability(icp_ctx *ctx, int32_t n_samples, const double *thetas, int32_t mode, const double *theta_ref,
                                      double *out);

/* ---------------------------------------------------------------- registration metrics
 * api/other/RegistrationComparison.scala:24-49 between the mesh of theta (reconstruction) and the target (ground truth):
 *   out[0] = MeshMetrics.avgDistance(reconstruction, target)       mean vertex-to-surface distance              (:25)
 *   out[1] = MeshMetrics.hausdorffDistance(reconstruction, target) max over both directions                     (:27)
 *   out[2], out[3] = boundary-aware average and maximum: reconstruction vertices whose closest target point's nearest target
 *                    vertex lies on the target's boundary are dropped                                           (:31-42)
 *   out[4] = number of vertices kept by that filter. */
ICP_API int icp_mesh_metrics(icp_ctx *ctx, const double *theta, double *out /* [5] */);

/* ---------------------------------------------------------------- fused chain step (measurement harness)
 * One call = all device work one Metropolis–Hastings step needs for a NEW state theta_prop proposed from
 * theta_cur, submitted as one stream sequence with a single synchronisation: the likelihood of theta_prop and,
 * for each of the n_props proposals, logT(cur -> prop) and logT(prop -> cur).  Results are identical to the
 * per-method entry points above (same kernels, same caches); only the number of host round trips differs.
 * fwd/bwd: [n_props]. */
ICP_API int icp_chain_eval_step(icp_evaluator *e, int32_t n_props, icp_proposal *const *props, const double *theta_cur,
                                const double *theta_prop, double *log_value_prop, double *fwd, double *bwd);

/* One call = propose (if generator >= 0) + icp_chain_eval_step, i.e. ALL device work of one Metropolis–Hastings step
 *  in one submission of five merged launches with a single synchronisation.
 *   generator >= 0 : theta_prop (out) = props[generator].propose(theta_cur) with the caller's standard normals z[r]
 *                    (NonRigidIcpProposal.scala:53-68);
 *   generator <  0 : theta_prop (in)  = a sample generated on the host (random-walk / pose proposals,
 *                    RandomShapeUpdateProposal.scala:31-35, PoseProposals.scala:31-90); z is ignored.
 * Then, as icp_chain_eval_step: log_value_prop = likelihood of theta_prop; fwd[i] / bwd[i] = logT(cur -> prop) /
 * logT(prop -> cur) of props[i].  Values are identical to the per-method entry points (same device code, same caches);
 * configurations the merged launches do not cover are routed through the per-stage kernels transparently. */
ICP_API int icp_chain_step(icp_evaluator *e, int32_t n_props, icp_proposal *const *props, int32_t generator,
                           const double *theta_cur, const double *z, double *theta_prop, double *log_value_prop, double *fwd,
                           double *bwd);

/* Optional companion of icp_chain_step: issue the first launches (proposal, instance, searches, correspondences) of the
 * step  theta_cur --generator, z--> proposal  (generator < 0: z_or_theta_prop is the proposed state itself) ahead of the
 * call that asks for it.  Meant to be called from the idle hook (below) with the NEXT step's arguments under the
 * assumption that the step in flight is rejected: an icp_chain_step with exactly these arguments then finds its first
 * half already on the device; any other call drops it.  Never changes results.  Does nothing when the posteriors of
 * theta_cur are not on record or the configuration is not covered by the merged launches.  n_props == 0 drops a
 * pending half step (a caller that stops stepping for a while should: the speculative work attached to it would
 * otherwise wait for its time-out). */
ICP_API int icp_chain_step_prelaunch(icp_evaluator *e, int32_t n_props, icp_proposal *const *props, int32_t generator,
                                     const double *theta_cur, const double *z_or_theta_prop);

/* B chains per launch (SURVEY.md §8b "*_batched variants", §8e "within a GPU, batch B chains per launch"; the
 * reference runs its chains from a ForkJoin pool, apps/femur/RunMHRandomInitComparison.scala:66): icp_chain_step for
 * n_chains independent chains in ONE sequence of five launches (chain = second grid dimension), the decompositions of
 * the chains that moved beside it.  Chain b = evaluators[b] with props[b*n_props .. b*n_props + n_props), generator[b],
 * theta_cur[b], z[b] (may be NULL where generator[b] < 0), theta_prop[b] (in or out as in icp_chain_step);
 * log_value_prop[b], fwd/bwd[b*n_props + i] and status[b] (ICP_OK / ICP_ERR_EMPTY / error of that chain) are written
 * per chain.  Every chain needs a context of its own (contexts hold the per-chain scratch; model and target are simply
 * given to each); chains on another device or of another rank than chain 0, a second chain on one context, and
 * configurations the merged launches do not cover take icp_chain_step one after the other.  Values are
 * bit-identical to icp_chain_step chain by chain.  Returns ICP_OK or the first failing chain's code.  Calls from
 * several threads must use disjoint sets of contexts. */
ICP_API int icp_chain_step_batched(int32_t n_chains, icp_evaluator *const *evaluators, int32_t n_props,
                                   icp_proposal *const *props, const int32_t *generator, const double *const *theta_cur,
                                   const double *const *z, double *const *theta_prop, double *log_value_prop, double *fwd,
                                   double *bwd, int32_t *status);

/* The same in two halves, for a caller that keeps two batches in flight (the decompositions of one run beside the
 * launches of the other; the C++ harness does): _issue returns when the batch's work is on the device, _collect waits for
 * it and writes the outputs named at _issue.  Everything passed to _issue by pointer (states, z, outputs) must stay valid
 * until _collect; the pointer ARRAYS themselves are copied.  A ticket is consumed by _collect or by _abandon (either may
 * come from any thread; no lock is held in between); until then every other entry point on a member context fails with
 * ICP_ERR_BUSY.  _abandon waits for the batch's launches and drops the step (nothing of it is recorded).  launch_ctx (may be NULL: the
 * first chain's context) names the context whose stream carries the launches: two batches given the SAME launch_ctx run
 * their launches one behind the other while the decompositions of the second run beside the launches of the first — at
 * most ICP_MAX_BATCHES_IN_FLIGHT tickets per launch_ctx at a time (one more: ICP_ERR_BUSY, nothing issued), collected in the
 * order they were issued. */
#define ICP_MAX_BATCHES_IN_FLIGHT 8
typedef struct icp_step_ticket icp_step_ticket;
ICP_API int icp_chain_step_batched_issue(int32_t n_chains, icp_evaluator *const *evaluators, int32_t n_props,
                                         icp_proposal *const *props, const int32_t *generator,
                                         const double *const *theta_cur, const double *const *z, double *const *theta_prop,
                                         double *log_value_prop, double *fwd, double *bwd, int32_t *status,
                                         icp_ctx *launch_ctx, icp_step_ticket **ticket);
ICP_API int icp_chain_step_batched_collect(icp_step_ticket *ticket);
ICP_API int icp_chain_step_batched_abandon(icp_step_ticket *ticket);

/* ---------------------------------------------------------------- the per-method entry points as ONE submission per step (round 6)
 * The drop-in contract is "Scalismo's chain, unchanged": MetropolisHastings.next (constructed at
 * api/sampling/SamplingRegistration.scala:52-58) calls  logValue(current) [memoised] → propose(current) → logValue(proposal) →
 * logTransitionRatio(current, proposal), and the MixtureProposal of api/sampling/MixedProposalDistributions.scala:48-68 turns the last
 * into logTransitionProbability(current, proposal) and (proposal, current) of EVERY ICP proposal (App. B2): with two ICP proposals six
 * device round trips per step where icp_chain_step has one.  A caller that cannot change that loop binds the chain's likelihood
 * evaluator and its ICP proposals (in the mixture's order) ONCE.  From then on
 *   - icp_proposal_propose of a bound proposal submits the WHOLE step — what icp_chain_step submits: the proposal, the proposed state's
 *     instance, its searches, every bound proposal's posterior there, the likelihood, all transition densities both ways — and returns
 *     when it is complete;
 *   - icp_evaluator_log_value of the bound evaluator for a state it has no value for (a random-walk or pose proposal made on the host)
 *     submits the same with the state of the PREVIOUS icp_evaluator_log_value call as the current state (MetropolisHastings.next evaluates
 *     the current state before it proposes; the Scala adapter passes every logValue through);
 *   - the calls that follow — logValue(proposal), logTransitionProbability(current, proposal), (proposal, current) — find their values
 *     on the host: the likelihood in the evaluator's Memoize(3), the densities parked under the exact (from, to) vectors of the two
 *     latest such steps.  Any other argument computes as before.
 * Values are those of the unbound calls, bit for bit (same device code, same caches; tests/test_gpu_dropin.py).  A bound step that
 * fails leaves the per-method call to compute — and report — on its own.  A proposal belongs to one binding at a time (binding it
 * again moves it); destroying a member dissolves the binding.  n_props == 0 unbinds.  Not for a caller that evaluates unrelated states
 * through the bound evaluator: every unseen state costs a whole step from the previous one. */
ICP_API int icp_chain_bind(icp_evaluator *e, int32_t n_props, icp_proposal *const *props);
/* out[0] whole steps submitted by icp_proposal_propose, out[1] by icp_evaluator_log_value, out[2] transition densities answered from
 * a parked step, since the binding was made. */
ICP_API int icp_chain_bind_stats(const icp_evaluator *e, int64_t out[3]);

/* ---------------------------------------------------------------- the whole Metropolis–Hastings loop on the device
 * n_steps steps of n_chains independent chains WITHOUT a host round trip per step: beside the five merged launches of
 * icp_chain_step_batched, a small kernel at the head of every step draws the mixture component and makes the step's proposal input
 * (Scalismo MixtureProposal.propose; api/sampling/proposals/RandomShapeUpdateProposal.scala:31-35), and one behind launch 5 is
 * MetropolisHastings.next (api/sampling/SamplingRegistration.scala:52-58): ModelPriorEvaluator × likelihood, the mixture's transition
 * ratio by log-sum-exp over all leaves, accept/reject, the step's record; the KL bases of accepted states follow in the same stream.
 * The host only enqueues launches and streams the standard normals in ahead.
 *   Mixture: (w_icp: the n_props ICP proposals with icp_weight) + (w_rw: shape random walk of rw_sigma) in the reference's order
 *   (apps/femur/IcpProposalRegistration.scala:70-72), optionally behind the six pose walks (w_pose > 0: apps/bfm/BfmFittingPartial.scala:70;
 *   the proposed pose's rotation matrix is made on the device with include/icp_sincos.h, the sines and cosines the host side and the
 *   oracle use — a context whose caller-supplied rotation matrices, icp_ctx_set_rotation, have all agreed with that convention is
 *   covered, one that ever supplied a different matrix is not: icp_ctx_rotation_convention); product evaluator = shape
 *   prior × `evaluator`.
 *   Random numbers: the counter-based generator of the C++ harness (host/icp_host.hpp StepRandom, shared bit for bit with the oracle):
 *   stream (seeds[b], step, lane), steps first_step[b] .. first_step[b] + n_steps − 1.
 *   theta[b] (in/out): the chain's current state; log_value[b] (in/out): its product log value (as MetropolisHastings carries it).
 *   records[b] (may be NULL): n_steps rows [index, accepted, leaf id (0/1 ICP proposal, 2 shape walk, 3..8 pose walks), log value, theta].
 * Covered (one context per chain, one device, one rank and sampler per run; otherwise ICP_ERR_INVALID_ARG and nothing has run):
 *   - what the five merged launches cover at ranks <= 64 (closed target or no boundary-aware branch): their launches, from
 *     device-resident records that change roles when a state is accepted;
 *   - (round 5) everything the WIDE step covers — targets with a boundary, the full-mesh Hausdorff evaluator, ranks up to 256 (round 6;
 *     200 until then: the reference's femur_gp_model_200-components.h5 has 201 components), the femur
 *     mixture at ranks 65..116 (apps/bfm/BfmFittingPartial.scala:62-96, apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala) — with
 *     the chains of a run sharing one model: the wide step's own launches replayed from device-resident records, an accepted state's
 *     posterior copied into the current state's entries; above rank 64 the proposed state is decomposed ahead of the decision, beside
 *     the factorisation and the evaluator's searches (KL-basis sampler; the Cholesky-root sampler up to rank 64).
 * Results are those of icp_chain_step_batched driven by the harness, chain by chain — bit for bit, except that the warm-started
 * Jacobi iteration of a wide step at ranks <= 64 starts from another basis than the host-stepped one's (states equal to 1e-11)
 * (tests/test_gpu_chain.py::test_device_loop_*, tests/test_gpu_wide_loop.py). */
typedef struct {
  /* = sizeof(icp_mh_mixture) of the header the caller was compiled against; anything else is refused with ICP_ERR_INVALID_ARG (the
   * structure grew in round 4: a caller built against the shorter one must not have its memory read past its end) */
  uint64_t struct_size;
  double icp_weight[2];
  double w_icp, w_rw;
  double rw_sigma;
  /* (round 4) the six pose walks of MixedProposalDistributions.mixedRandomPoseProposal (MixedProposalDistributions.scala:29-39;
   * PoseProposals.scala:31-90), first in the outer mixture as in apps/bfm/BfmFittingPartial.scala:70: w_pose = 0 -> none.
   * pose_rot_sigma = (yaw, pitch, roll), pose_trans_sigma = (x, y, z): the argument order of :29. */
  double w_pose;
  double pose_rot_sigma[3], pose_trans_sigma[3];
} icp_mh_mixture;
ICP_API int icp_chains_run_on_device(int32_t n_chains, icp_evaluator *const *evaluators, int32_t n_props, icp_proposal *const *props,
                                     const icp_mh_mixture *mixture, const uint64_t *seeds, const int64_t *first_step,
                                     double *const *theta, double *log_value, int32_t n_steps, double *const *records,
                                     int64_t *accepted);

/* ---------------------------------------------------------------- instrumentation (bench.py's roofline leg)
 * Between start and stop every kernel the context launches is bracketed by HIP events on the context stream;
 * stop returns one row per kernel name.  Off by default (adds nothing to the launch path). */
typedef struct {
  char name[40];
  int64_t calls;
  double total_ms, min_ms, max_ms;
} icp_kernel_stat;
ICP_API int icp_ctx_profile_start(icp_ctx *ctx, int32_t max_launches);
/* on != 0: the NEXT profile_start .. profile_stop interval also counts the tests the searches execute (per wave, with atomics on
 * a handful of device words — they slow the filter launches down, so this is for a short leg of its own, not for timing);
 * profile_stop then returns extra rows "count.surface_ball_tests", "count.surface_sphere_tests", "count.surface_exact_tests",
 * "count.vertex_filter_tests", "count.vertex_exact_tests" with the number in `calls`. */
ICP_API int icp_ctx_profile_search_counters(icp_ctx *ctx, int32_t on);
ICP_API int icp_ctx_profile_stop(icp_ctx *ctx, icp_kernel_stat *stats, int32_t capacity /* >= 32 */, int32_t *n_out);

/* ---- fall-back counters.  The step schedules above take their cross-stream order on the device (a launch waits for a word another
 * stream's launch raises) with time-outs behind them; every time-out ends in a slower but equivalent schedule, so a normal run must
 * show ZERO everywhere — anything else is a performance bug (or a tool that lets one kernel run at a time: rocprofv3 --pmc).
 * ctx == NULL: totals of the process since it started. */
typedef struct {
  int64_t wait_timeouts;        /* a step's first launch gave up waiting for its word (50 ms on the device) */
  int64_t speculation_giveups;  /* a KL basis started ahead never saw its input (5 ms) and the step that drew from it was repeated */
  int64_t pipeline_fallbacks;   /* contexts switched to the one-stream schedule for good after a time-out */
  int64_t step_redos;           /* steps computed twice because of any of the above */
  int64_t gate_timeouts;        /* batched steps: the launch sequence was not released because the batch's decompositions did not
                                   become resident in time (2 s) */
  int64_t reserved[3];
} icp_runtime_stats;
ICP_API int icp_ctx_runtime_stats(const icp_ctx *ctx, icp_runtime_stats *out);

/* ---- which path the chain steps took (diagnostic: the results do not depend on it).  Counts since the context was created or, with
 * ctx == NULL, of the process: out[0] the five merged launches (icp_chain_step[_batched]), out[1] the wide step (targets with a
 * boundary, the Hausdorff evaluator, ranks up to 256, pose moves: DESIGN.md), out[2] per-stage kernels, out[3] steps taken inside
 * icp_chains_run_on_device. */
ICP_API int icp_ctx_step_paths(const icp_ctx *ctx, int64_t out[4]);

/* Is the KL basis the proposal would draw from at `theta` ready?  2: the posterior of theta is on record and decomposed (or needs no
 * decomposition); 1: its decomposition is still on the device (started ahead by the step that proposed theta) — an ICP proposal from
 * theta would wait for it; 0: nothing on record (a step from theta computes it).  Never blocks.  A caller that steps many independent
 * chains uses it to let a chain whose basis is still under way sit out a round instead of holding the others back. */
ICP_API int icp_proposal_basis_state(icp_proposal *p, const double *theta);

/* Which path icp_chain_step[_batched] takes for this proposal set and evaluator (a property of the configuration, not of a state):
 * 0 the five merged launches, 1 the wide step, 2 per-stage kernels.  A caller that steps many chains uses it to size its batches:
 * the merged launches are short and want several groups of chains in flight, a wide step is long and wants one. */
ICP_API int icp_chain_step_path(icp_evaluator *e, int32_t n_props, icp_proposal *const *props);

/* ---- model cache.  Contexts made from the same model arrays share its derived device data (the scaled basis in two layouts, the
 * Gram matrix and its inverses: 0.35 s of host work and 2 x 137 MB of uploads at N = 28,561, rank 200).  The library keeps the two
 * most recently used models alive after their last context is destroyed, so that a job which builds one context per target over one
 * model (BASELINE.json configs[4]) pays for the model once; this call drops them (their device memory is freed as soon as no context
 * uses them).
 * Likewise the streams, pinned host blocks and device buffers of destroyed contexts, proposals and evaluators are kept for the next
 * ones (a job that makes its chains anew for every target spent a third of its time creating and destroying them): up to 96 streams
 * per device and priority class, 64 MiB of pinned memory, 6 GiB of device memory in blocks of at most 64 MiB.  This call gives all
 * of that back as well; the environment variable ICP_NO_POOL=1 (read when the library is loaded) switches the pools off. */
ICP_API void icp_release_cached_models(void);

/* ---- idle hook (optional).  icp_chain_step spends most of a step waiting for the device.  A caller that has host
 * work which does not depend on the step's outcome — drawing the random numbers of the NEXT step, say — registers it
 * here: `fn(arg)` is called once per icp_chain_step, on the calling thread, after the step's launches have been issued
 * and before the wait.  It may call icp_chain_step_prelaunch and nothing else of this library.  fn == NULL removes the hook.  Nothing in the
 * reference corresponds to it (its Breeze RNG is sequential); the C++ harness uses it with its counter-based RNG. */
typedef void (*icp_idle_fn)(void *arg);
ICP_API int icp_ctx_set_idle_hook(icp_ctx *ctx, icp_idle_fn fn, void *arg);

#ifdef __cplusplus
}
#endif
#endif /* ICP_PROPOSAL_H */
