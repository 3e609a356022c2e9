// icp_kernels.hpp — host-callable launchers of the HIP kernels (all asynchronous on the given stream).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "icp_device.hpp"

#include <cstdlib>
#include <vector>

namespace icp {

// Developer A/B switches and test hooks read the environment ONLY in builds made for that purpose (-DICP_DEV_SWITCHES for the
// tools/, -DICP_TEST_HOOKS for libicp_proposal_amd_testhooks.so, which tests/ load through ICP_LIBRARY_PATH); the shipped
// library answers nullptr.  Operational switches (ICP_NO_PIPELINE, ICP_SPECULATION, ICP_HOST_TIMING) use getenv directly.
inline const char* dev_env(const char* name) {
#if defined(ICP_DEV_SWITCHES) || defined(ICP_TEST_HOOKS)
  return std::getenv(name);
#else
  (void)name;
  return nullptr;
#endif
}

// ---- optional per-kernel timing with HIP events on the launch stream (bench.py's roofline leg).
// Off by default: when `g_prof` is null the launch wrappers add nothing.
enum KernelId {
  KID_INSTANCE = 0, KID_SURFACE_INIT, KID_SURFACE_FILTER, KID_SURFACE_RESOLVE,
  KID_VERTEX_INIT, KID_VERTEX_FILTER, KID_VERTEX_RESOLVE, KID_TRI_SPHERES, KID_CORRESPOND,
  KID_REGRESSION, KID_FACTOR, KID_TAIL, KID_EIGEN, KID_PROPOSE, KID_REDUCE,
  KID_STEP_BEGIN, KID_STEP_FILTER, KID_STEP_RESOLVE, KID_STEP_REGRESSION, KID_STEP_FINISH, KID_COUNT
};
extern const char* const kKernelNames[KID_COUNT];

constexpr int kSearchCounters = 8;
struct Profiler {
  struct Rec { hipEvent_t a, b; int id; };
  std::vector<Rec> pool;
  size_t used = 0;
  bool overflow = false;
  unsigned long long* counters = nullptr;  // device, kSearchCounters: executed tests of the searches (SurfaceTask::stats)
  void begin(hipStream_t st, int id);
  void end(hipStream_t st);
};
extern thread_local Profiler* g_prof;

struct ProfScope {  // RAII: events around one kernel launch
  hipStream_t st;
  bool on;
  ProfScope(hipStream_t st, int id) : st(st), on(g_prof != nullptr) { if (on) g_prof->begin(st, id); }
  ~ProfScope() { if (on) g_prof->end(st); }
};

// ---- geometry (kernels_geometry.hip)

// K1: x = s(R(x̄ + μ + Q c − ctr) + ctr + t).  Qp = scaled basis in planes [(j*3+d)*N + i]; coeffs on device.
void launch_instance(hipStream_t st, int N, int r, const double* Qp, const double* ref, const double* mean,
                     const Pose& pose, const double* coeffs, double* x);
// … keeping every point's deformation mean + Q·c, and a state that differs in its pose only from those (bit-identical points)
void launch_instance_keep(hipStream_t st, int N, int r, const double* Qp, const double* ref, const double* mean,
                          const Pose& pose, const double* coeffs, double* x, double* defo);
void launch_instance_pose(hipStream_t st, int N, const double* ref, const Pose& pose, const double* defo_in, double* x, double* defo_out);

// K2: all vertex normals (diagnostic entry point; the posterior kernels compute normals on demand)
void launch_vertex_normals(hipStream_t st, int N, const double* x, const int* tris, const int* adj_off,
                           const int* adj, double* normals);

// f32 bounding sphere (centroid, conservatively inflated max corner distance) of every triangle: spheres[t] = {cx,cy,cz,R}
// `spheres` holds sphere_floats4(T) float4: the T spheres, then the T triangle ids they belong to (position -> triangle)
void launch_tri_spheres(hipStream_t st, int T, const double* verts, const int* tris, const int* order /* device, or nullptr */,
                        float4* spheres);
inline size_t sphere_floats4(int T) { return (size_t)(T > 0 ? T : 0) + ((size_t)(T > 0 ? T : 0) + 3) / 4 + 1; }
std::vector<int> coherent_triangle_order(int V, int T, const double* verts /* host */, const int* tris /* host */);

// One brute-force query batch against a triangle mesh or a vertex set.  `hint` carries the previous winner of
// each query (any valid index gives a valid upper bound; kNoIndex/-1 = none) and receives the new winner.
struct QueryBuffers {   // scratch of one query batch; per-query arrays hold the batch size rounded up to a multiple of 4
  double* thr2;          // vertex queries: squared-distance bound
  float4* qrec;          // surface queries: f32 copy of the point
  float* thrA;           // surface queries: distance bound (inflated, f32)
  int* cnt;              // candidates per query
  int* cand;             // candidate lists, one row of cand_stride(n_elements) ints per query
  size_t cand_capacity;  // ints in `cand`; a batch takes floor(capacity / stride) queries
};

// Candidate lists hold at most this many entries per query.  With a hint (the previous winner) a query lists a few dozen; a list
// that overflows — a first search without hints — keeps COUNTING, and the resolve stage, seeing a count above the capacity,
// scans the whole searched set for that query instead (same lexicographic minimum, exact either way).  Lists sized for the worst
// case (every element a candidate of every query) were 97 MB per chain at the metric size and 1 GiB for a full-mesh Hausdorff pass.
constexpr int kCandStride = 512;      // at least (fewer elements than that: all of them)
constexpr int kCandStrideMax = 4096;  // at most, where the scratch has room: a query far from the searched surface (a partial target's
                                      // hole) has thousands of triangles within its bound
inline int cand_stride(int n_elems) { return n_elems < kCandStride ? (n_elems > 0 ? n_elems : 1) : kCandStride; }
// entries per query of a batch of K queries given `capacity` ints of scratch
inline int cand_stride_for(int n_elems, int K, size_t capacity) {
  size_t s = capacity / (size_t)(K + 4 > 0 ? K + 4 : 1);
  if (s > (size_t)kCandStrideMax) s = kCandStrideMax;
  if (s < (size_t)kCandStride) s = kCandStride;
  if (s > (size_t)(n_elems > 0 ? n_elems : 1)) s = n_elems > 0 ? n_elems : 1;
  return (int)s;
}

struct SurfaceTask {  // one batch of closest-point-on-surface queries against one triangle mesh
  int K, Kpad, T, stride;
  const double* P;        // [K*3] query points
  const double* verts;
  const int* tris;
  const float4* spheres;  // [T] f32 bounding spheres, followed by the triangle id of every list position; nullptr: the searched
                          // mesh has just been made (the current model instance inside a merged step) — the filter computes the spheres
                          // itself, in the order given by `order`
  const int* order;       // position in the sphere list -> triangle (used when spheres == nullptr)
  int* hint;              // [K] previous winner (in/out; may be null)
  float4* qrec;           // [Kpad] scratch
  float* thrA;            // [Kpad] scratch; nullptr: the bounds (distance to the hinted triangle) are taken by the filter itself,
                          // when the searched mesh is complete
  int* cnt;               // [Kpad] scratch
  int* cand;              // [Kpad*stride] scratch
  double* cp;             // outputs, any may be null
  double* d2;
  int* tri;
  int tblocks, ksplit, kchunk;  // filter decomposition: tblocks × ksplit workgroups
  // profiling only (nullptr otherwise): executed tests, counted per wave — [0] ball tests (one query against a wave's patch),
  // [1] sphere tests (one query against one triangle's bounding sphere), [2] exact point–triangle evaluations of the resolve stage
  unsigned long long* stats;
};

struct VertexTask {  // one batch of nearest-vertex queries against one vertex set
  int K, Kpad, V, stride;
  const double* P;
  const double* verts;
  int* hint;
  double* thr2;
  int* cnt;
  int* cand;
  double* d2;  // outputs, any may be null
  int* idx;
  int vblocks, ksplit, kchunk;
  unsigned long long* stats;  // profiling only: [3] exact point–vertex distances of the filter (every pair), [4] of the resolve stage
};

void split_queries(int n_elem_blocks, int Kpad, int* ksplit, int* kchunk);

// K4: closest point on surface.  Outputs (any may be null): cp [K*3], d2 [K], tri [K].
void launch_surface_query(hipStream_t st, int T, const double* verts, const int* tris, const float4* spheres,
                          int K, const double* P, int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri);

// K3: nearest vertex.  Outputs (any may be null): d2 [K], idx [K].
void launch_vertex_query(hipStream_t st, int V, const double* verts, int K, const double* P, int* hint,
                         const QueryBuffers& qb, double* d2, int* idx);

// ---- posterior assembly + r-space algebra (kernels_posterior.hip)

struct CorrBuffers {   // per-correspondence data of one ICP posterior (device)
  int* id;             // [K] model vertex id
  int* aux;            // [K] nearest target vertex of the surface point (ModelSampling) or -1
  double* pt;          // [K*3] target-side point
  unsigned char* keep; // [K]
  double* nhat;        // [K*3] unit vertex normal at id on the current mesh
  double* e;           // [K*3] observation minus mean: R^T((pt − t) − ctr) + ctr − x̄_id − μ_id
};

struct CorrTask {  // everything one correspondence needs besides its search result
  int K;
  CorrBuffers cb;
  const double* x;               // current instance
  const double* tpts;            // TargetSampling: the decimated-target points (else null)
  const unsigned char* boundary; // ModelSampling: target boundary flags (indexed by nnv); TargetSampling: model boundary flags
  const int* nnv;                // ModelSampling: nearest target vertex of the surface point (null = not needed)
  int boundary_aware;
  Pose pose;
  const double* ref;
  const double* mean;
  const int* tris;
  const int* adj_off;
  const int* adj;
  // (the wide step) non-null: this task hangs behind a NEAREST-VERTEX search of the surface points `cp` (NonRigidIcpProposal.scala:98):
  // the resolve wave of query k builds the ModelSampling correspondence of model id k from cp[k] and the vertex it has just found
  const double* cp = nullptr;
};

// optional extra of the correspondence launches: the memo entry's coefficient copy and cleared status words (coeffs_dst == nullptr: none)
struct EntryInit { const double* coeffs_src = nullptr; double* coeffs_dst = nullptr; int r = 0; int* status = nullptr; };
// NonRigidIcpProposal.scala:89-110 (ModelSampling): ids 0..K-1, surface points cp, optional nearest-vertex ids
void launch_correspond_model(hipStream_t st, int K, const double* x, const double* cp, const int* nnv,
                             const unsigned char* tgt_boundary, int boundary_aware, const Pose& pose,
                             const double* ref, const double* mean, const int* tris, const int* adj_off,
                             const int* adj, const CorrBuffers& cb, const EntryInit& init = EntryInit{});
// NonRigidIcpProposal.scala:112-131 (TargetSampling): target points + nearest model vertex ids
void launch_correspond_target(hipStream_t st, int K, const double* x, const double* tpts, const int* nn_id,
                              const unsigned char* model_boundary, int boundary_aware, const Pose& pose,
                              const double* ref, const double* mean, const int* tris, const int* adj_off,
                              const int* adj, const CorrBuffers& cb, const EntryInit& init = EntryInit{});

// K5a (f64 MFMA): partial sums Mpart[s][(r+1)x(r+1)] of Σ_kept [Q_i | e_i]^T Σ_i^-1 [Q_i | e_i]; *splits_out = number of partials.
// Mpart must hold regression_splits(K)·(r+1)² doubles.
int regression_splits(int K);
// How many of a posterior's regression_splits(K) leaves one wave folds into its partial: 1 (every leaf a wave and a partial of its own:
// a lone chain needs the parallelism) or all of them (one partial per posterior: no split-K traffic, no summing launch) — decided by
// how many output tiles the launch carries across all its chains.  Either way the summed matrix has the same bits.
int regression_fold(int K, int r, int n_posteriors_in_launch);
int regression_macro(int r, int fold);          // macro-tile edge of a folded posterior's units (1: single tiles)
int regression_units(int r, int leaves, int fold, int macro);  // work units (waves) of one posterior in a regression launch
void launch_regression(hipStream_t st, int K, int r, const double* Q, const CorrBuffers& cb, double w_tangent,
                       double kappa, double* Mpart, int* splits_out);

// K5b, up to kWideMaxChains (16) posteriors per launch: M = I + Σ partials, alpha = M^-1 b (Cholesky); status[0] != 0 if M is not SPD.
struct PosteriorFactorIO { const double* Mpart; int splits; double* M; double* alpha; int* status; double* scratch /* (r+1)·r, large ranks only */;
                           double* Lout = nullptr; double* Sout = nullptr; /* optional: the factor L (r × r, zero upper triangle) and 1/diag(L) */ };
void launch_posterior_factor(hipStream_t st, int r, int n_post, const PosteriorFactorIO* io);
int posterior_factor_max();  // posteriors per launch
// Σ of one posterior's split-K partials into its first partial, on many CUs (what the factor kernels do themselves otherwise; with
// it done, they take splits = 1), and M = I + that sum, both triangles, from the summed partial — the start of a decomposition
// that does not wait for the factorisation (icp_chain_eval_step)
void launch_sum_partials(hipStream_t st, int r, double* Mpart, int splits);
void launch_assemble_posterior_matrix(hipStream_t st, int r, const double* Mpart_summed, double* M);

// a9 tails, up to 2·kWideMaxChains (32) per launch: out = −½ γ^T M γ − (r/2) ln 2π with (G + σ²M) γ = G (c_from + (c_to − c_from)/step − α).
// Iterative (needs Ginv = G^-1); status[0] != 0 = did not contract -> use the direct kernel.
struct TransitionTailIO {
  const double* alpha; const double* M; const double* c_from; const double* c_to; double step; double* out; int* status;
  // (optional) the three status words of the posterior this tail belongs to, passed on to `relay_out` — next to the step's other
  // results, so that they reach the host in the step's ONE result copy instead of a copy of their own
  const int* relay_in = nullptr; int* relay_out = nullptr;
};
void launch_transition_tails(hipStream_t st, int r, int n, const TransitionTailIO* io, const double* Ginv, double sigma2);
void launch_transition_tail_direct(hipStream_t st, int r, const TransitionTailIO& io, const double* G, double sigma2, double* work /* r*r */);

// eigen-decomposition of D M^-1 D (posterior KL basis): V columns (and its transpose Vt), S descending, canonical signs
// Vwarm (optional): eigenvectors of a nearby posterior, used as the starting basis of the Jacobi iteration
size_t eigen_work_doubles(int r);  // size of `work`
// EigenSpec (optional, ranks <= 64 only): a decomposition started before it is known to be needed.
//   splits   > 0: `M` points at the split-K partials of the regression launch (splits × (r+1)² row-major; lower triangle
//            used, identity not yet added) and the kernel sums them itself, in the order the factorisation does — the same
//            values as the finished r×r M, available one launch earlier
//   cancel   pinned host word, polled once per sweep: the decomposition gives up (writes nothing) once *cancel == seq
//            (seq != 0)
//   ready    device word raised (to ready_seq or beyond) by the regression launch when the partials are complete: the
//            decomposition is enqueued without a stream dependency on that launch and waits for the word itself
constexpr int kEigenGaveUp = 3;  // pinned status of a speculative decomposition whose input never arrived
struct EigenSpec { int splits; const int* cancel; int seq; const int* ready; int ready_seq;
                   long long* wait_ticks = nullptr; /* profiling: 100 MHz ticks spent waiting for `ready` are added here */ };
bool eigen_speculation_supported(int r);
// up to two decompositions of the same rank in ONE launch (they run side by side); false: not available for this rank
// done_word (optional): set to done_value (release, agent scope) when THIS decomposition's outputs are complete — or when it
// gave up — so that a consumer on another stream can wait for one of the two without waiting for the whole launch
struct EigenRequest { const double* M; const double* Vwarm; double* V; double* Vt; double* S; double* work; int* status;
                      const EigenSpec* spec; int* host_status; int* done_word; int done_value;
                      const double* sqrt_lambda = nullptr; /* of the request's own model; launch_posterior_eigen_many needs it */
                      bool direct = false; /* launch_posterior_eigen_pair: take the tridiagonal route (state-independent time) instead of
                      the warm-started iteration — worth it while the chain moves fast (burn-in: the iteration needs 4 sweeps) */
                      bool root = false; /* write V := D·L⁻ᵀ (M = L·Lᵀ), S := 1 instead of the eigen-decomposition: the opt-in
                      Cholesky-root sampler (kernels_posterior.hip: k_posterior_root; ranks <= 64) */ };
bool launch_posterior_eigen_pair(hipStream_t st, int r, const double* sqrt_lambda, int n, const EigenRequest* rq);
// one decomposition as the rank <= 64 kernels see it (EigenRequest resolved into the kernel's own pointers)
struct EigenProblem {
  const double* M; const double* Vwarm /* may be Vout */; double* Vout; double* Vtout; double* Sout; int* status;
  double* rotlog; int* meta; double* vpos; EigenSpec spec; int launch_id; int* host_status; int* done_word; int done_value;
  const double* sqrt_lambda;  // of this problem's model (nullptr: the launch's)
};
EigenProblem eigen_problem_of(int r, const EigenRequest& rq);
// decompositions (root != 0: Cholesky factors, k_posterior_root) of n records that LIVE IN DEVICE MEMORY, skip[i] != 0 leaving record i
// out: the on-device chain loop's launch (its decide kernel writes records and skip flags)
void launch_posterior_eigen_resident(hipStream_t st, int r, int n, const EigenProblem* records, const int* skip, int root);
// any number of decompositions of one rank (the chains of icp_chain_step_batched) in ONE launch (up to 80; more: a second launch on
// the same stream); every request carries its model's sqrt_lambda.  pinned_records: eigen_many_record_bytes(n) bytes of pinned host
// memory that stay untouched until the launch has finished (the kernel reads its records there); arrive (may be null): a device
// counter every workgroup of the launch increments when it starts.  Returns the number of workgroups launched, -1 if the rank is
// not covered (nothing launched).
size_t eigen_many_record_bytes(int n);
int launch_posterior_eigen_many(hipStream_t st, int r, int n, const EigenRequest* rq, void* pinned_records, int* arrive);
void eigen_debug_dump(const double* work, int r);
void library_release_stream(hipStream_t st);  // drops the library handle kept for `st` (ranks > 64), before the stream is destroyed
void launch_posterior_eigen(hipStream_t st, int r, const double* M, const double* sqrt_lambda, const double* Vwarm, double* V,
                            double* Vt, double* S, double* work /* eigen_work_doubles(r) */, int* status,
                            const EigenSpec* spec = nullptr, int* host_status = nullptr /* pinned copy of *status; honoured
                            when eigen_speculation_supported(r) */,
                            int part = 0 /* 0: every launch; 1: the reduction to tridiagonal form only (ranks 65..256: the long
                            one-workgroup launch at the head of the chain), 2: the launches behind it — a caller with other work
                            to issue puts it between the two (a launch costs the host 3-6 µs, the chain has eleven) */);

// a8: c' = c + step·((G+σ²I)^-1 G (α + D^-1 V √S z) − c), with P = (G+σ²I)^-1
void launch_propose(hipStream_t st, int r, const double* alpha, const double* V, const double* S,
                    const double* inv_sqrt_lambda, const double* P, double sigma2, const double* c,
                    const double* z, double step, double* c_out, int root = 0,
                    const int* relay_in = nullptr /* three status words passed on to … */, int* relay_out = nullptr);

// ---- deterministic ICP (api/other/IcpBasedSurfaceFitting.scala:46-126)
// P[k] = x[ids[k]] (:72)
void launch_gather_points(hipStream_t st, int K, const double* x, const int* ids, double* P);
// correspondence records with isotropic noise: id = ids[k] (or nn[k]), e = pt − x̄_id − μ_id in WORLD space (:81), keep = 1, n̂ = 0
void launch_correspond_plain(hipStream_t st, int K, const int* ids, const double* pts, const double* ref, const double* mean,
                             const CorrBuffers& cb);
// c <- c + step·((G+σ²I)⁻¹ G α − c) with P = (G+σ²I)⁻¹ (:84-85)
void launch_mean_step(hipStream_t st, int r, const double* alpha, const double* P, double sigma2, double step, double* c);

// ---- posterior variability (apps/util/PosteriorVariability.scala:30-73): X = [S][N*3] sample meshes
void launch_accumulate(hipStream_t st, int n, const double* src, double scale_after /* 0 = none */, double* acc);
void launch_variability(hipStream_t st, int N, int S, const double* X, int mode, const double* normals, double* out);

// ---- evaluator reductions (kernels_posterior.hip)
// out[0] = Σ log N(sqrt(d2_k); mean, sigma)
void launch_sum_gauss_logpdf(hipStream_t st, int K, const double* d2, double mean, double sigma, double* out);
// out[0] = Σ kept distances, out[1] = max kept distance, out[2] = number kept;  flag[k] != 0 drops the point
// (flags may be null; idx (optional) indexes flags: flag = flags[idx[k]] if idx[k] < n_flags else 0)
// out_max (zero or a distance beforehand) = max(out_max, max_k sqrt(d2[k])): many workgroups, exact (the Hausdorff evaluator's reduction)
void launch_dist_max(hipStream_t st, int K, const double* d2, double* out_max);
void launch_dist_stats(hipStream_t st, int K, const double* d2, const unsigned char* flags, const int* idx,
                       int n_flags, double* out);

// ---- merged launches of one Metropolis–Hastings step (kernels_step.hip): five dependent launches do what the
// per-stage kernels above do in ~25, with identical arithmetic (same device bodies).

struct ProposeIn {   // a8 inputs (all device pointers; z may also point into kernel arguments)
  const double* alpha; const double* V; const double* S; const double* inv_sqrt_lambda; const double* P;
  const double* c; const double* z; double sigma2, step;
  int root = 0;  // 1: the opt-in Cholesky-root sampler — V holds L (M = L·Lᵀ, row-major lower triangle), S holds 1/diag(L), and
                 // D⁻¹·W·z = L⁻ᵀ·z is ONE back substitution instead of the product with the KL basis (k_posterior_root)
};

constexpr int kStepBeginPoints = 128;  // model points per workgroup of the first launch (256 threads; half of them carry a point:
                                       // the instance streams 24·r bytes per point through ONE CU's L2 port, so the points are
                                       // spread over twice as many CUs as threads alone would need)
constexpr int kStepInlineZ = 128;  // ranks up to this pass the r host-drawn numbers inside the kernel arguments
constexpr int kStepMaxOut = 6;

struct StepBeginArgs {  // launch 1: [propose] -> coefficients -> instance -> search initialisation
  int N, r, inst_blocks, tpr_log2;
  // Consecutive steps alternate between two streams, so the first four launches of a step run beside the last launch of
  // the step BEFORE them (which they do not depend on).  What they must not overtake is that step's searches — same scratch, same hints — so the launch waits, on
  // the device, for the word the finish launch of that step raises when it starts (*wait_flag - wait_seq >= 0); an event
  // there would hold that step's own launches back.  *wait_error (pinned) is set if the word does not come within 50 ms.
  const int* wait_flag; int wait_seq; int* wait_error;
  const int* wait2_flag; int wait2_seq;  // likewise: the word of the eigen-decomposition the proposal draws from (EigenRequest::done_word)
  long long* wait_ticks;  // profiling (may be null): workgroup 0 adds the 100 MHz ticks it spent waiting for the two words
  int hold_regs;  // the decomposition is still in flight: use the launch variant that holds the model data in registers across the wait
                  // (k_step_begin_reg: ≈ 3 µs slower by itself, ≈ 6 µs less behind the wait)
  const double* Qp; const double* ref; const double* mean;
  Pose pose;
  int propose;              // 1: coefficients = a8 from `prop` (prop.z ignored: see zin/z_ptr); 0: coefficients = zin / z_ptr
  ProposeIn prop;
  const double* z_ptr;      // host-visible pointer used when r > kStepInlineZ
  double zin[kStepInlineZ];
  int n_out;                // copies of the coefficient vector (state slot, posterior entries, pinned host result)
  double* out[kStepMaxOut];
  double* x;                // [N*3] instance
  int has_surf, has_vert;
  SurfaceTask surf;         // queries = model ids 0..K-1 of the NEW instance against the target surface
  VertexTask vert;          // searched set = the NEW instance (TargetSampling)
  int* zero2; int n_zero2;  // a second list of candidate counters to reset (the evaluator's target -> model queries)
};

// Filter grid of one task, XCD-aware: the `ksplit` workgroups that stream the SAME block of elements (one per query
// chunk) get indices that are equal modulo 8, i.e. the same XCD under round-robin dispatch, so that the block is
// fetched from HBM once and served to the others by that XCD's L2 (first-block index of the task must be a multiple of 8).
//   index l  ->  element block (l / (8·ksplit))·8 + l % 8,  query chunk (l % (8·ksplit)) / 8
inline int filter_grid_blocks(int elem_blocks, int ksplit) { return (elem_blocks + 7) / 8 * 8 * ksplit; }

struct StepSearchArgs {  // launches 2 (filter) and 3 (resolve + correspondences)
  int n_surf, n_vert;
  int fstart[5];           // filter: first block of each task (surface tasks first), fstart[n] = grid size
  int rstart[5];           // resolve: first block (= first query) of each task
  SurfaceTask s[2];
  VertexTask v[2];
  int s_corr[2], v_corr[2];  // index into corr[] of the posterior fed by the task, or -1
  CorrTask corr[2];
};

// every surface task of the step comes with resident spheres and bounds taken by launch 1 (the light form of the batched filter launch)
inline bool step_filter_prepared(const StepSearchArgs& a) {
  for (int i = 0; i < a.n_surf; ++i)
    if (a.s[i].spheres == nullptr || a.s[i].thrA == nullptr) return false;
  return true;
}

// 16×16 output tiles of the lower triangle of the (r+1)×(r+1) normal matrix (see regression_tile)
__host__ __device__ inline int regression_tiles(int r) { const int nt = (r + 1 + 15) >> 4; return nt * (nt + 1) / 2; }

struct StepRegressionArgs {  // launch 4: normal-equation partial sums of every posterior + the likelihood reduction
  int n, r, ntiles;          // posteriors; tiles per (r+1)x(r+1) matrix
  int ustart[3];             // first work unit (tile x split) of each posterior; ustart[n] = number of units
  int K[2], kchunk[2];
  int fold[2];               // leaves of kchunk correspondences a unit takes (regression_tile): 1, or all of a posterior's (one partial)
  int macro[2];              // folded posteriors only: a unit owns macro x macro neighbouring tiles (regression_macro_fold); 0 / 1: single tiles
  const double* Q;
  CorrBuffers cb[2];
  double wt[2], kappa[2];
  double* Mpart[2];
  int* status[2];            // the new entries' 3 status ints: {-, eigen sweeps, eigen} are cleared here
  // (folded posteriors of the wide step) the correspondences' operand rows, made by a launch of its own ahead of the regression
  // (k_wide_xrows): X[(k·4 + j)·xrs + xcol(col)] = row j of [Q_i | e_i] for j < 3, n̂ᵀ[Q_i | e_i] for j = 3, zeros behind column r — what
  // regression_load / regression_mac work out per tile and correspondence.  xcol interleaves the two 16-column tiles of a macro block
  // (column 32·m + 16·p + i at 32·m + 2·i + p): a lane's operands of both tiles are ONE 16-byte load (regression_macro_fold_x).
  // null: gathered from the basis (regression_macro_fold)
  double* X[2];
  int xrs, xpad_;

  int reduce_kind;           // 0 none, 1 Σ log N(d; mean, sigma), 2 {Σ d, max d, count}
  int Kred;                  // model -> target distances (0: that direction is not evaluated)
  const double* d2;
  int Kred2;                 // target -> model distances of a TargetToModel / Symmetric evaluator (0: none); results at red_out + 4
  const double* d2b;
  double mean, sigma;
  double* red_out;
};

struct StepFinishArgs {  // launch 5: per posterior Cholesky + alpha, then the backward tail; forward tails beside them
  int n, r, n_lds, tpr_log2;
  int tail_base;             // >= 0: LDS offset (doubles) where the factor workgroups stage M and G⁻¹ for their tail (set by the launcher)
  const double* Mpart[2]; int splits[2];
  double* M[2]; double* alpha[2]; int* status[2];      // status: the entry's 3 ints {chol, -, eigen}
  int* host_status[2];                                  // pinned copy of the Cholesky status
  TransitionTailIO bwd[2];   // prop -> cur, needs this launch's factorisation
  TransitionTailIO fwd[2];   // cur -> prop, from the cached posterior of the current state
  const double* Ginv; double sigma2;
  // completion signal: the last workgroup to finish stores `seq` into pinned host memory (the host polls it instead of
  // paying a stream synchronisation)
  int* done_counter; int* host_flag; int seq;
  // raised (to seq) as soon as this launch starts: everything the regression launch wrote is visible from then on — a
  // speculative eigen-decomposition enqueued on another stream waits for this word instead of for an event
  int* ready_flag;
};

// launch 4b (only where a posterior has many split-K partials — every model point a correspondence, configs[2]: 64 × 83 KB): the
// partials are summed, in split order, by as many threads as the matrix has entries, into the first one; the finish launch — ONE
// workgroup per posterior — then reads 83 KB instead of 5.3 MB (100 µs at one CU's fetch rate).  Same bits: the factorisation
// sums the splits in the same order from 0.0.
constexpr int kStepReduceSplits = 32;  // from this many splits on
struct StepReduceArgs { int n; int nn; double* Mpart[2]; int splits[2]; };
void launch_step_reduce(hipStream_t st, const StepReduceArgs& a);
bool step_finish_supported(int r);
void launch_step_begin(hipStream_t st, const StepBeginArgs& a);
void launch_step_filter(hipStream_t st, const StepSearchArgs& a);
void launch_step_resolve(hipStream_t st, const StepSearchArgs& a);
void launch_step_regression(hipStream_t st, const StepRegressionArgs& a);
void launch_step_finish(hipStream_t st, const StepFinishArgs& a);

// B chains per launch (icp_chain_step_batched).  While a capture is set for the calling thread, the five launchers above
// record their (finalised) arguments and grid sizes in it instead of launching; launch_step_batch then issues ONE
// sequence for all captured chains: blockIdx.y = chain, arguments from an array in device memory that the first kernel of
// the sequence copies out of `pinned` (step_batch_bytes(B) bytes each, 16-byte aligned).  All chains must have one rank.
struct StepCapture {
  StepBeginArgs begin; StepSearchArgs search; StepRegressionArgs regression; StepFinishArgs finish;
  int grid[5];
};
void step_capture(StepCapture* c);  // nullptr: launch as usual
size_t step_batch_bytes(int B);
// st_finish/ev (optional): the fifth launch goes to st_finish behind an event recorded on st.
// gate (optional): the sequence's first kernel (one workgroup's worth of copying) does not end before the device counter
// gate.counter has reached gate.expected — i.e. before every workgroup of the decompositions this batch waits for ON THE DEVICE
// has started (launch_posterior_eigen_many's `arrive`).  The launch behind it fills the chip with workgroups that spin on those
// decompositions' completion words; ordered like this they can never keep them from becoming resident.  After 2 s the gate opens
// anyway and says so in gate.error (pinned).
struct StepBatchGate { const int* counter = nullptr; int expected = 0; int* error = nullptr; };
void launch_step_batch(hipStream_t st, int B, const StepCapture* caps, void* pinned, void* device, hipStream_t st_finish = nullptr,
                       hipEvent_t ev = nullptr, StepBatchGate gate = StepBatchGate{});
// ---- the on-device Metropolis–Hastings loop (kernels_step.hip: k_mh_front / k_mh_decide; icp_abi.hip:
// icp_chains_run_on_device).  One record per chain in device memory.  The five merged launches read their per-chain arguments from
// device-resident arrays (as in icp_chain_step_batched); the arguments exist in TWO alternatives per chain — which of a proposal's two
// posterior entries holds the CURRENT state — and k_mh_front copies the right one into the live array at the head of every step and
// fills in what changes from step to step (which proposal generates, its standard normals or the random-walk sample).  k_mh_decide,
// behind launch 5, is Scalismo's MetropolisHastings.next for the step (prior, product value, mixture transition ratio by log-sum-exp over
// ALL leaves, accept/reject with the step's uniform draw, record) and, for an accepted state, writes the decomposition records of its
// posteriors.  Random numbers: the harness' counter-based generator (host/icp_host.hpp: splitmix64 over (seed, step, lane)) — the
// uniforms are integer arithmetic and are drawn on the device, bit for bit the host's; the standard normals need log / cos and are
// drawn on the host with the harness' own expression and streamed in ahead of the steps that use them.
struct WideProposeItem;
struct WideInstArgs;
// … of a chain that takes the WIDE step (kernels_wide.hip) inside the loop: its launch records live in device memory and are the same
// every step — an accepted state's posterior is COPIED into the current state's entries (k_mhw_adopt) instead of the two sets
// changing roles — but for what the step's head sets: the proposal's inputs and the proposed state's pose
struct MhWide {
  WideProposeItem* item;      // the chain's W1 record: kind / in / src
  ProposeIn prop_in[2];       // a8 from the CURRENT state's posterior, per ICP proposal (z: set per step)
  double* given;              // [r] coefficients of a proposal that does not draw them from a posterior (shape walk: c + σ·z; pose walk: c)
  WideInstArgs* inst;         // W2 record: the proposed state's pose
  StepSearchArgs* search[2];  // W4-W7 records of the main sequence: the correspondences' inverse rigid transform
  const int* chol[2];         // the factorisation's status word of the proposed state's posterior, per ICP proposal
};
struct MhChain {
  // -- fixed for the run
  unsigned long long seed;
  int r, n_icp;
  int n_outer, outer_kind[3];            // outer mixture in the reference's order (apps/bfm/BfmFittingPartial.scala:70: pose, ICP, shape walk);
                                         // kind 0 = the pose mixture, 1 = the ICP mixture, 2 = the shape random walk
  double outer_w[3];                     // normalised weights
  // the six pose walks (api/sampling/proposals/PoseProposals.scala:31-90; MixedProposalDistributions.scala:29-39: yaw, pitch, roll,
  // x, y, z, weight 0.5 each), leaf ids 3..8: the parameter each perturbs, its σ, log √(2π) + log σ
  int n_pose, pose_index[6];
  double pose_w[6], pose_sigma[6], pose_logc[6];
  int front_every_step;                  // (mixtures with pose walks: every step's head by k_mh_front — the proposed pose is made there)
  int pose_move;                         // this step's proposal is a pose walk (its leaf: `leaf`)
  double prop_pose[10];                  // … the proposed state's first ten parameters
  double icp_w[2];                       // normalised inner weights of the ICP mixture
  double rw_sigma, rw_logc;              // rw_logc = 0.5·(r·log 2π + r·log σ²)
  double prior_c;                        // 0.5·r·log 2π
  int eval_kind, eval_mode;              // icp_eval_kind / icp_eval_mode
  double gauss_mean, gauss_sigma, gauss_logn /* log sqrt(2π) + log σ */, exp_rate, exp_lograte;
  const StepBeginArgs* begin_alt[2]; StepBeginArgs* begin_live;
  const StepSearchArgs* search_alt[2]; StepSearchArgs* search_live;
  const StepRegressionArgs* regression_alt[2]; StepRegressionArgs* regression_live;
  const StepFinishArgs* finish_alt[2]; StepFinishArgs* finish_live;
  ProposeIn prop_alt[2][2];              // [cur_sel][ICP proposal]: the proposal from the current state's posterior
  EigenProblem eig_alt[2][2];            // [new cur_sel][ICP proposal]: decomposition of the accepted state's posterior, warm start = the other set's basis
  EigenProblem* eig_live;                // [n_icp] this chain's records in the launch's array
  int* eig_skip;                         // [n_icp]
  int pw_id_mask;                        // launch ids of the decompositions: 1 + q % mask
  const double* coeff_prop;              // proposed coefficients as launch 1 wrote them (the state slot's copy)
  const double* red;                     // launch 4's likelihood reductions [8]
  const double* tails;                   // fwd_i = tails[2i], bwd_i = tails[2i + 1]
  const int* tail_status;                // [2·n_icp] (fixed-point tail did not contract: the host's direct kernel would be needed)
  const int* chol_status;                // [n_icp]
  const double* normals;                 // [steps][r] standard normals of this chain, row = step − normals_first
  long long normals_first;
  int normals_rows;                      // rows of the block in `normals` (k_mh_decide prepares the NEXT step itself while its normals are there)
  double* records;                       // [steps][4 + 10 + r], row = step − rec_first
  long long rec_first;
  // -- chain state
  double* theta;                         // [10 + r] current state
  double cur_p;                          // its log product value (prior × likelihood)
  long long step, accepted;
  int cur_sel, gen, leaf, error;         // error != 0: the chain needs the host (non-contracting tail, non-finite value, …) and stands still
  int eig_seq[2];                        // decompositions of proposal i so far (cold every 128th, as the host path)
  MhWide* wide;                          // (fixed for the run) non-null: the chain takes the wide step — k_mhw_front / k_mh_decide_wide
};
void launch_mh_set_normals(hipStream_t st, int B, MhChain* chains, const double* base, int stride, int offset, int rows);
void launch_mh_front(hipStream_t st, int B, MhChain* chains);
void launch_mh_decide(hipStream_t st, int B, MhChain* chains);
// the wide step's head and tail (ranks up to 256; every step's head by the front kernel); then, per posterior q of the launch (chain-major,
// n_icp per chain), the proposed state's M, alpha and coefficients into the current state's entries where eig_skip[q] == 0
void launch_mhw_front(hipStream_t st, int B, MhChain* chains, int restate = 0 /* != 0: the "proposal" is the current state itself */);
void launch_mhw_decide(hipStream_t st, int B, int r, MhChain* chains);
struct MhAdopt {
  const double* M_from; double* M_to; const double* alpha_from; double* alpha_to; const double* c_from; double* c_to;
  // (ranks above 64: the proposed state's posterior has been decomposed AHEAD of the decision, beside the factorisation and the
  // evaluator's searches — its basis and the decomposition's status word move too; null: decomposed in place afterwards)
  const double* V_from; double* V_to; const double* Vt_from; double* Vt_to; const double* S_from; double* S_to; const int* st_from; int* st_to;
  // the correspondence records of the posterior (what icp_proposal_posterior hands out for the state)
  const unsigned char* corr_from[6]; unsigned char* corr_to[6]; int corr_bytes[6];
};
void launch_mhw_adopt(hipStream_t st, int r, int n, const MhAdopt* records, const int* skip /* null: every record */);
// the five merged launches for B chains from DEVICE-RESIDENT argument arrays (no copy kernel, no gate: one stream, in order)
void launch_step_batch_resident(hipStream_t st, int B, const int grid[5], int r, const StepBeginArgs* begin, const StepSearchArgs* search,
                                const StepRegressionArgs* regression, const StepFinishArgs* finish, bool filter_prepared = false,
                                bool reg_folded = false /* the records' regression_fold > 1: the folded regression kernel */);

// ---- the wide step (kernels_wide.hip): one Metropolis–Hastings step of B chains for the configurations the five merged launches
// above do not cover — targets WITH boundary (the nearest-vertex pass of NonRigidIcpProposal.scala:98-99 and of
// CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:44-63 as a second filter/resolve stage), the full-mesh Hausdorff
// evaluator (HausdorffDistanceEvaluator.scala:31-35), ranks up to 256 (factorisation, tails and decomposition as launches of their own:
// none of them fits the one-workgroup finish launch), pose moves (PoseProposals.scala:31-90: the instance is the kept deformations
// under the new pose).  Same device bodies as the per-stage kernels, no host round trip inside the step, B chains side by side:
//
//   W1  propose            one workgroup per chain               (a8; or the given coefficients handed on)
//   W2  instance           points × chain groups                 (the basis is read ONCE per group of up to 8 chains) + bounds of the
//                                                                  model -> target queries
//   W3  prepare            spheres of the new instance, bounds of the target -> model queries, counters
//   W4/W5  filter / resolve, stage 1   (surface searches both ways, TargetSampling's vertex search, correspondences without a flag test)
//   W6/W7  filter / resolve, stage 2   (nearest vertices of the surface points; ModelSampling correspondences with their boundary flag)
//   W8  regression + likelihood reductions
//   W9  sum of the split-K partials -> [eigen stream: assemble, reduce to tridiagonal form, solve, refine]
//   W10 factorisation    W11 transition tails    W12 results and completion flag into pinned host memory
constexpr int kWideMaxChains = 16;  // chains per launch sequence (several kernels take their per-chain arguments by value)

struct WideProposeItem {   // W1
  int kind;                // 1: a8 from `in` (in.z: the step's standard normals); 0: the coefficients are given (`src`)
  ProposeIn in;
  const double* src;
  int n_out; double* out[5];   // copies: the state slot, the new posterior entries, the pinned result
};
struct WideProposeArgs { int n; WideProposeItem it[kWideMaxChains]; };
void launch_wide_propose(hipStream_t st, int r, const WideProposeArgs& a);
// … of n records that live in device memory (the on-device loop; KL-basis sampler above rank 64)
void launch_wide_propose_resident(hipStream_t st, int r, int n, const WideProposeItem* items);

struct WideInstArgs {      // W2, per chain (device memory)
  int kind;                // 0: x = pose(ref + mean + Q·c), deformations kept; 1: x = pose(ref + defo_src) (a pose move)
  const double* coeffs;
  const double* defo_src;
  Pose pose;
  double* x; double* defo;
  int has_surf; SurfaceTask surf;   // model ids 0..K-1 of the new instance against the target surface: bounds from the hints
  // (a step whose evaluator searches run beside the proposal's: the ids from surf.K on are a task of their own, query k = id surf.K + k)
  int has_surf2; SurfaceTask surf2;
};
struct WidePrepArgs {      // W3, per chain (device memory)
  int T; const double* x; const int* tris; const int* order; float4* spheres;   // T = 0: no search of the new instance's surface
  int has_t2m; SurfaceTask t2m;
  int n_cnt; int* cnt[4]; int cnt_n[4];   // candidate counters of the vertex searches (their bounds are taken by the filter)
  double* zero_d; int n_zero_d;           // reduction outputs that accumulate (atomic maxima)
};
struct WideRegArgs {       // W8, per chain (device memory)
  StepRegressionArgs reg;  // (reduce_kind unused: the reductions below)
  int eval_kind, eval_m2t, eval_t2m;      // icp_eval_kind; which directions are evaluated
  int Km; const double* d2m; const unsigned char* flags_m; const int* idx_m;   // model -> target (flags null: nothing is dropped)
  int Kt; const double* d2t; const unsigned char* flags_t; const int* idx_t;   // target -> model
  int n_flags;
  double mean, sigma;
  double* red_out;         // [8] device: the layout finish_eval reads
};
struct WideDoneItem {      // W12
  const double* red_src; double* red_dst;      // 8 doubles
  const int* st_src[4]; int* st_dst[4];        // single status words (factorisations): device -> pinned
  int* host_flag; int seq;
};
struct WideDoneArgs { int n; WideDoneItem it[kWideMaxChains]; };

size_t wide_batch_bytes(int B);
struct WideLaunchPlan {
  int grid_xrows = 0;       // … and read their operand rows from StepRegressionArgs::X: workgroups of the launch that makes them (per chain and posterior)
  bool reg_folded = false;  // the chains' posteriors fold their split-K leaves (StepRegressionArgs::fold > 1): the folded regression kernel    // what the host has worked out for a batch: common model data, grids
  int B, N, r;
  const double* Qp; const double* ref; const double* mean;
  int grid_prep, grid_f1, grid_r1, grid_f2, grid_r2, grid_reg;
  int grid_f1b, grid_r1b, grid_f2b, grid_r2b, grid_regb;  // the evaluator's own sequence (s1b, s2b, regb), if any chain of the batch has one
  bool f1_prepared;
  // (round 6; the on-device loop) the first blocks of 64 model points of the instance launch that the MAIN sequence reads — the
  // proposal's K model ids and the corners of their triangles (vertex normals): launch_wide_head_resident part 1 makes these, the main
  // sequence follows at once, part 2 — the other 98 % of the instance and W3, which only the evaluator's searches read — runs beside it
  // on another stream.  0: no such split (a TargetSampling proposal searches the whole instance; a mesh whose ids are not local)
  int inst_head_blocks = 0;
};
// s1 / s2 / reg: the step's searches and reductions — or only what the PROPOSAL needs (its K model ids, their nearest vertices, the
// regression: what the factorisation, the tails and the decomposition wait for), the evaluator's searches and reductions being a
// sequence of their own (s1b, s2b, regb) that the host puts behind the main one — or, for the full-mesh Hausdorff distance (0.2 ms of
// chip-wide searches), beside it on another stream
struct WideChainArgs { WideInstArgs inst; WidePrepArgs prep; StepSearchArgs s1, s2; WideRegArgs reg; StepSearchArgs s1b, s2b; WideRegArgs regb; };
// copies the chains' records into `pinned` (wide_batch_bytes(B)); launches the copy to `device`, the instances and W3 on `st`
void launch_wide_head(hipStream_t st, const WideLaunchPlan& plan, const WideChainArgs* chains, void* pinned, void* device);
// the same records laid out in host memory `dst` (wide_batch_bytes(B)) / W2 and W3 from records already in `device`
void wide_pack_args(int B, const WideChainArgs* chains, void* dst);
void launch_wide_head_resident(hipStream_t st, const WideLaunchPlan& plan, void* device,
                               int part = 0 /* 1: the instance launch's head (plan.inst_head_blocks); 2: the rest of it and W3; 0: everything */);
// where chain b's records sit inside such a block
WideInstArgs* wide_inst_record(void* block, int B, int b);
StepSearchArgs* wide_search_record(void* block, int B, int stage /* 0: s1, 1: s2 */, int b);
// W4..W8 of the records in `device` (launch_wide_head): the main sequence, the evaluator's own
void launch_wide_main(hipStream_t st, const WideLaunchPlan& plan, void* device);
void launch_wide_eval(hipStream_t st, const WideLaunchPlan& plan, void* device);
int wide_reg_blocks(const WideRegArgs& a);
int wide_prep_grid(const WidePrepArgs& a);
void launch_wide_done(hipStream_t st, const WideDoneArgs& a);
// Σ of the split-K partials of n posteriors into their first partial (as launch_sum_partials)
void launch_sum_partials_many(hipStream_t st, int r, int n, double* const* Mpart, const int* splits);
// the tridiagonal route (ranks 65..256) for n decompositions side by side; assemble != 0: M = I + summed partial is written first
// (launch_assemble_posterior_matrix) from rq[i].spec... no: from `parts[i]`.  Every request needs its own `work`.  No Jacobi fall-back
// inside the sequence: a spectrum the multisection cannot separate ends with status 2 in the request's status words, and the caller
// decomposes that posterior again through launch_posterior_eigen.
bool eigen_tridiag_many_supported(int r);
constexpr int kTriManyMax = 32;  // posteriors per launch of the tridiagonal route (launch_posterior_eigen_tridiag_many takes any number, in launches of this many)
void launch_posterior_eigen_tridiag_many(hipStream_t st, int r, int n, const EigenRequest* rq, const double* const* parts /* may be null */,
                                         const int* skip = nullptr /* device, [n]: != 0 leaves request i alone (the on-device loop) */,
                                         int part = 0 /* 0: the whole sequence; 1: M and the reduction to tridiagonal form only; 2: what follows
                                                         it — the on-device loop issues the two apart: the decision falls while the reduction
                                                         runs, and part 2 then skips the chains that did not move */);

// the number of chains whose searches share the launch being put together (thread-local; 1 = a lone chain): how far a task's queries are
// split over workgroups (split_queries, split_surface_queries).  Never changes a result — only the partition of the work.
void search_chains_hint(int n_chains);
SurfaceTask make_surface_task(int T, const double* verts, const int* tris, const float4* spheres, int K, const double* P,
                              int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri);
VertexTask make_vertex_task(int V, const double* verts, int K, const double* P, int* hint, const QueryBuffers& qb, double* d2, int* idx);
int query_batch(int K, int n_elems, size_t cand_capacity);

}  // namespace icp
