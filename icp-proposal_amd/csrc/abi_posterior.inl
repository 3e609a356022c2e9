// abi_posterior.inl — part of icp_abi.hip (one translation unit; included there, in order).
// posterior memo entries, icp_proposal / icp_evaluator and their methods (NonRigidIcpProposal.scala:88-153, the evaluators)
namespace {
struct PosteriorEntry {
  std::vector<double> theta;
  bool valid = false, eig_valid = false, eig_checked = false;  // eig_checked: its status has reached the host copy
  bool reserved = false;  // handed out to a step whose launches are in flight: not to be recycled
  hipEvent_t eig_done = nullptr;  // recorded on the eigen stream behind the launch that holds the entry's decomposition
  hipEvent_t eig_done_shared = nullptr;  // … or, not owned, the event of the entry it shared that launch with
  // a shared event out of the device's ring (next_batch_event, icp_chain_step_batched): the slot's generation counter (static
  // storage) and its value when the event was recorded for this entry — a slot recorded again since stands for LATER work on
  // possibly another stream, which orders nothing of this entry's: such a waiter synchronises with the eigen streams on the host
  // instead (eigen_event() returns nullptr then; await_eigen / mpart_for_write fall back to sync_eigen)
  const uint64_t* eig_shared_gen = nullptr;
  uint64_t eig_shared_gen_value = 0;
  bool eigen_event_stale() const { return eig_event_valid && eig_done_shared && eig_shared_gen && *eig_shared_gen != eig_shared_gen_value; }
  bool eig_event_valid = false;   // an event stands for the latest decomposition of this entry (the chain step's own launches
                                  // of ranks <= 64 record none: their consumers wait for the completion word on the device, and
                                  // an event record is 2-3 µs of host time on the accepted path)
  int done_value = 0;             // … and what the entry's word in icp_proposal::eig_words holds once it is complete (0: none)
  uint64_t stamp = 0;
  DBuf<int> id, aux;
  DBuf<double> pt, nhat, e;
  DBuf<uint8_t> keep;
  DBuf<double> coeffs, M, alpha, V, Vt, S;
  int status_off = 0;  // this entry's 3 ints inside the proposal's status buffer
  ~PosteriorEntry() { if (eig_done) (void)hipEventDestroy(eig_done); }
  hipEvent_t eigen_event() const { return (!eig_event_valid || eigen_event_stale()) ? nullptr : (eig_done_shared ? eig_done_shared : eig_done); }
  CorrBuffers corr() const { return CorrBuffers{id.p, aux.p, pt.p, keep.p, nhat.p, e.p}; }
};

}  // namespace

struct icp_proposal {
  icp_ctx* ctx = nullptr;
  icp_proposal_params prm{};
  int K = 0;
  DBuf<double> target_pts;
  DBuf<int> hint_nn;      // TargetSampling: last nearest model vertex of each target point
  DBuf<int> nn_id;
  DBuf<double> work;      // r*r scratch of the eigen / direct-tail kernels
  DBuf<double> xrows;     // (wide step, folded regression) the correspondences' operand rows: K · 4 · 16·⌈(r + 1)/16⌉ values (StepRegressionArgs::X)
  DBuf<double> Mpart;     // split-K partial normal matrices of the regression kernel, two halves: the merged step alternates
                          // between them so that a speculative decomposition can still read the previous step's
  size_t mpart_half_doubles = 0;
  int mpart_half = 0;
  static constexpr int kMpartRing = 4;  // (a speculative decomposition reads the partials of the step that started it: with four
                                        // buffers used in turn the writer of a buffer practically never finds its reader still at work)
  PosteriorEntry* mpart_reader[kMpartRing] = {nullptr, nullptr, nullptr, nullptr};  // the entry whose decomposition reads the buffer
  double* mpart_for_write(int half, hipStream_t st);  // `st` (where the writer runs) waits for that reader first, if it is still at work
  DBuf<double> fscratch;  // (r+1)·r + 8 factorisation scratch (ranks too large for LDS)
  const double* warm_ptr = nullptr;  // eigenvectors of the most recent posterior (inside its memo entry): warm start of the next
  bool warm_valid = false;
  // The eigen-decompositions of this proposal run on the context's eigen stream (icp_ctx::eig_stream) in launch order — they
  // share `work` and the warm start, so they must not overlap each other — beside the chain's own streams: the decomposition
  // of a state that is not needed yet — the other ICP direction of a freshly accepted state — overlaps the chain's next
  // steps instead of delaying a later one.
  // Speculative decomposition (icp_chain_step, ICP_SPECULATION=1): the KL basis of the PROPOSED state's posterior is started as soon as its
  // normal matrix exists, before the caller has decided whether to accept.  The next call tells: its current state is
  // the proposed one (the basis is already on its way) or not (the decomposition is cancelled through `h_cancel`).
  int sampler = ICP_SAMPLER_EIGEN; // icp_proposal_set_sampler: what the "decomposition" of a posterior writes into V / S
  int* h_eig = nullptr;            // pinned: eigen status of every memo entry, written by the decomposition itself
  DBuf<int> eig_words;             // per memo entry: sequence number of its last finished decomposition (EigenRequest::done_word)
  int eig_seq = 0;
  int* h_cancel = nullptr;         // pinned, 16 slots: the decomposition with sequence number q gives up once slot q%16 holds q
  int spec_seq = 0;
  PosteriorEntry* spec_entry = nullptr;
  // fills the request of a speculative decomposition of `e` (the caller launches it, possibly together with another
  // proposal's, on the context's eigen stream and records e.eig_done behind it)
  void speculate_eigen(PosteriorEntry& e, const PosteriorEntry& cur, int splits, int half /* of Mpart: the step's partials */,
                       const int* ready, int ready_seq, EigenSpec* spec_out, EigenRequest* rq_out);
  void resolve_speculation(const double* theta_cur);
  DBuf<int> status;       // 3 ints per memo entry: {chol(M), chol(G+σ²M), eigen}
  // (pinned: the copy of `status` into it is a true asynchronous copy — into a pageable vector it was a synchronising one, 20-30 µs
  // per step of the per-stage paths)
  struct PinnedInts {
    int* p = nullptr; size_t n = 0;
    void assign(size_t count, int v) {
      if (p) pinned_free(p);
      pinned_alloc((void**)&p, sizeof(int) * count);
      n = count;
      for (size_t i = 0; i < count; ++i) p[i] = v;
    }
    int& operator[](size_t i) { return p[i]; }
    int* data() { return p; }
    ~PinnedInts() { if (p) pinned_free(p); }
  } h_status;
  std::unique_ptr<PosteriorEntry[]> memo;
  uint64_t clock = 0;

  // side: (optional) the stream the factorisation goes to, behind the regression on the context stream (see icp_chain_eval_step)
  PosteriorEntry& posterior(const double* theta, bool want_aux, hipStream_t side = nullptr);
  bool side_factor_pending = false;  // a factorisation on the side stream may still read Mpart / write fscratch
  bool side_asm_pending = false;     // … and a decomposition's first launch on the eigen stream the summed partials
  double* side_parts = nullptr;      // the summed partials of the latest posterior(…, side), until the next regression …
  const PosteriorEntry* side_parts_entry = nullptr;  // … and the entry they belong to
  void issue_factor(PosteriorEntry& e, PosteriorFactorIO io, double* parts, int splits, hipStream_t side, bool root_here);
  PosteriorEntry* find_entry(const double* theta);
  PosteriorEntry& fresh_entry();
  void alloc_entry(PosteriorEntry& e);
  void prepare_eigen(PosteriorEntry& e, EigenRequest* rq);
  void ensure_eigen(PosteriorEntry& e);  // enqueue on the context's eigen stream (no-op if done or in flight)
  // … or on `es` (eig_stream / eig_stream2) with that stream's work buffer; the caller has made `es` wait for the entry's M
  // part: as launch_posterior_eigen's — 1 issues the chain's head, 2 what follows it and the event behind everything
  void ensure_eigen_on(PosteriorEntry& e, hipStream_t es, int part = 0);
  EigenRequest pending_rq{};
  PosteriorEntry* pending_entry = nullptr;
  // icp_chain_bind: the evaluator of the chain this proposal is a component of, and its place in that chain's proposal list; a bound
  // proposal's propose() submits the chain's WHOLE step and parks the values the per-method calls behind it ask for (ChainBinding)
  icp_evaluator* bound_eval = nullptr;
  int bound_index = -1;
  DBuf<double> work2;  // eig_stream2's (ranks above 64)
  unsigned eig_flip = 0;
  void await_eigen(PosteriorEntry& e);   // make the context stream wait for it
  void check_status(PosteriorEntry& e);
};

namespace {
// Launches 1-3 of a merged step (proposal -> instance -> searches -> correspondences), enqueued.  icp_chain_step issues
// them itself, or finds them already issued by icp_chain_step_prelaunch for exactly its arguments.
struct StepFront {
  bool valid = false;
  int n_props = 0, generator = -1;
  icp_proposal* props[2] = {nullptr, nullptr};
  std::vector<double> theta_cur, key;  // key: z (generator >= 0) or the proposed state (generator < 0)
  PosteriorEntry* ec[2] = {nullptr, nullptr};
  PosteriorEntry* ep[2] = {nullptr, nullptr};
  StateSlot* s = nullptr;
  bool eigen_first_use = false;
  int parity = 0;  // which half of the pinned coefficient area its first launch writes, and which of the two streams the step uses
  hipStream_t stream = nullptr;
  int splits[2] = {1, 1};                  // launch 4: split count and partial-sum buffers of every posterior
  double* mpart[2] = {nullptr, nullptr};
  int mpart_half[2] = {0, 0};
  int Ksurf = 0;
};
constexpr int kCoeffArea = 512;  // doubles per half of that area (>= kMaxRank)
constexpr int kReduceArea = 16 + 2 * kCoeffArea;  // pinned result of launch 4's likelihood reduction: 8 doubles per parity
}  // namespace

struct icp_evaluator {
  icp_ctx* ctx = nullptr;
  icp_evaluator_params prm{};
  StepFront front;  // pre-launched first half of the next step, if any
  int front_parity = 0;
  // Acceptance estimate of the chain stepped through this evaluator (icp_chain_step): the decomposition of the PROPOSED
  // state is started speculatively unless next to nothing is being accepted.
  std::vector<double> last_prop;  // the state the previous merged step proposed
  double acc_ema = 0.5;
  DBuf<double> target_pts;
  // target-side queries against the CURRENT model surface
  int Kt = 0;              // number of target-side query points (decimated target, or all target vertices for Hausdorff)
  const double* d_tpts = nullptr;
  DBuf<int> hint_tri, hint_nnv, t2m_tri, t2m_nnv;
  uint64_t points_hash = 0;   // of its target-side query points (HintSeed::Eval)
  bool hints_filed = false;   // … its hints have been offered to the target's seed (or came from it)
  DBuf<double> t2m_cp, t2m_d2;
  struct Memo {
    std::vector<double> theta;
    bool valid = false;
    uint64_t stamp = 0;
    double value = 0.0, aux[4] = {0, 0, 0, 0};
    int status = 0;
  } memo[kEvalMemo];
  uint64_t clock = 0;
  // icp_chain_bind (include/icp_proposal.h): the chain's ICP proposals in the caller's order, the state of the latest logValue call
  // (MetropolisHastings.next evaluates the current state before it proposes: SURVEY App. B1), and the transition densities of the two
  // latest whole steps submitted on behalf of a per-method call, keyed by the exact (current, proposed) vectors
  struct ChainBinding {
    int n = 0;
    icp_proposal* props[8] = {};
    std::vector<double> last_theta;  // empty: no logValue call yet
    struct Parked {
      bool valid = false;
      std::vector<double> cur, prop;
      double fwd[8] = {}, bwd[8] = {};
    } parked[2];
    int next = 0;
    int64_t steps_from_propose = 0, steps_from_log_value = 0, parked_hits = 0;  // icp_chain_bind_stats
  } bind;
};

namespace {

template <class F>
int guard(F&& f) {
  try {
    f();
    return ICP_OK;
  } catch (const IcpError& e) {
    g_err = e.msg;
    return e.code;
  } catch (const std::bad_alloc&) {
    g_err = "host out of memory";
    return ICP_ERR_DEVICE;
  } catch (const std::exception& e) {
    g_err = e.what();
    return ICP_ERR_DEVICE;
  }
}

void require(bool ok, const char* msg) {
  if (!ok) fail(ICP_ERR_INVALID_ARG, msg);
}

void check_theta_finite(const icp_ctx* ctx, const double* theta) {
  require(theta != nullptr, "theta is null");
  for (int i = 0; i < 10 + ctx->r; ++i)
    if (!std::isfinite(theta[i])) fail(ICP_ERR_NOT_FINITE, "theta contains a non-finite value");
}

}  // namespace

// ===================================================================== posterior (NonRigidIcpProposal.scala:88-153)

PosteriorEntry* icp_proposal::find_entry(const double* theta) {
  const size_t P = 10 + (size_t)ctx->r;
  // (two states of a chain share their first ten numbers — the pose — more often than not: the last coefficient tells most entries
  // apart before the comparison of the whole vector; compared as bits, like memcmp does)
  uint64_t last;
  std::memcpy(&last, theta + P - 1, sizeof last);
  for (int i = 0; i < kPosteriorMemo; ++i) {
    if (!memo[i].valid) continue;
    uint64_t mine;
    std::memcpy(&mine, memo[i].theta.data() + P - 1, sizeof mine);
    if (mine == last && std::memcmp(memo[i].theta.data(), theta, sizeof(double) * P) == 0) return &memo[i];
  }
  return nullptr;
}

// least recently used memo entry, emptied; the caller fills it and sets `valid`
// device buffers of a memo entry (all entries at proposal creation, see icp_ctx::alloc_slot)
void icp_proposal::alloc_entry(PosteriorEntry& e) {
  if (e.M.p) return;
  const int r = ctx->r, Ka = std::max(K, 1);
  e.id.alloc(Ka); e.aux.alloc(Ka); e.pt.alloc(3 * (size_t)Ka); e.nhat.alloc(3 * (size_t)Ka); e.e.alloc(3 * (size_t)Ka);
  e.keep.alloc(Ka);
  e.coeffs.alloc(r); e.M.alloc((size_t)r * r);
  e.alpha.alloc(r); e.V.alloc((size_t)r * r); e.Vt.alloc((size_t)r * r); e.S.alloc(r);
  e.status_off = 3 * (int)(&e - &memo[0]);
  HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
}

PosteriorEntry& icp_proposal::fresh_entry() {
  PosteriorEntry* lru = nullptr;
  for (int i = 0; i < kPosteriorMemo; ++i) {
    PosteriorEntry& e = memo[i];
    if (e.reserved) continue;
    if (!lru) { lru = &e; continue; }
    if (!e.valid) { if (lru->valid) lru = &e; }
    else if (lru->valid && e.stamp < lru->stamp) lru = &e;
  }
  if (!lru) fail(ICP_ERR_DEVICE, "internal: every posterior entry is reserved");
  PosteriorEntry& e = *lru;
  alloc_entry(e);
  e.valid = false;
  e.eig_valid = false;
  e.eig_checked = false;
  return e;
}

PosteriorEntry& icp_proposal::posterior(const double* theta, bool want_aux, hipStream_t side) {
  icp_ctx& c = *ctx;
  const int r = c.r;
  const size_t P = 10 + (size_t)r;
  if (side_factor_pending) {
    // a factorisation that went to the side stream ahead of its use (icp_chain_eval_step, pose moves) may still be writing the M and
    // alpha of an entry this call hands out — and it reads the partials and the factor scratch a new posterior would overwrite
    HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));
    side_factor_pending = false;
  }
  if (PosteriorEntry* hit = find_entry(theta)) {
    PosteriorEntry& e = *hit;
    e.stamp = ++clock;
    if (want_aux && prm.direction == ICP_MODEL_SAMPLING) {
      // diagnostic request for corr_aux on a cached entry: recompute the nearest-vertex ids into it
      StateSlot& s = c.state(theta);
      c.ensure_nnv_prefix(s, K);
      HIP_OK(hipMemcpyAsync(e.aux.p, s.surf_nnv.p, sizeof(int) * K, hipMemcpyDeviceToDevice, c.stream));
    }
    return e;
  }
  // With a side stream the WHOLE posterior goes there — searches, correspondences, regression and (as before) the factorisation —
  // behind the state's instance: the caller's evaluator searches the same state on the context stream at the same time (its own
  // query range, its own scratch set) instead of 65 µs later.  fs / sw: the stream and the scratch set of this posterior's front.
  const hipStream_t fs = side ? side : c.stream;
  const int sw = side ? 1 : 0;
  PosteriorEntry& e = fresh_entry();
  if (e.eig_event_valid && e.eigen_event()) {  // a decomposition that may still read this entry's M (started ahead, its state not kept)
    HIP_OK(hipStreamWaitEvent(fs, e.eigen_event(), 0));
    e.eig_event_valid = false;
  }
  e.theta.assign(theta, theta + P);
  e.valid = true;
  e.stamp = ++clock;
  StateSlot& s = c.state(theta);  // :141 currentMesh
  if (side) {  // (the points, the slot's coefficients: launched or copied on the context stream, possibly just now)
    if (c.ev_inst_slot != (const void*)&s) HIP_OK(hipEventRecord(c.ev_inst, c.stream));  // (else: on record already, ahead of launches that need not be waited for)
    HIP_OK(hipStreamWaitEvent(side, c.ev_inst, 0));
  }
  const EntryInit init{s.coeffs.p, e.coeffs.p, r, status.p + e.status_off};  // (status: {chol, eigen sweeps (diagnostic), eigen})
  if (prm.direction == ICP_TARGET_SAMPLING) {
    // :117-118 nearest vertex of the current mesh for every decimated-target point
    QueryBuffers qb = c.query_scratch(K, c.N, sw);
    launch_vertex_query(fs, c.N, s.x.p, K, target_pts.p, hint_nn.p, qb, nullptr, nn_id.p);
    launch_correspond_target(fs, K, s.x.p, target_pts.p, nn_id.p, c.boundary.p, prm.boundary_aware, s.pose, c.ref.p,
                             c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, e.corr(), init);
  } else {
    // :94-99 closest surface point of the target for model ids 0 until K; nearest target vertex only when the
    // boundary test can change anything (the target has boundary vertices) or the caller asked for it
    c.ensure_surface_prefix(s, K, fs, sw);
    const bool need_nnv = want_aux || (prm.boundary_aware && c.target.n_boundary > 0);
    if (need_nnv) c.ensure_nnv_prefix(s, K, fs, sw);
    launch_correspond_model(fs, K, s.x.p, s.surf_cp.p, need_nnv ? s.surf_nnv.p : nullptr, c.target.boundary.p,
                            prm.boundary_aware, s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, e.corr(), init);
  }
  // :152 interpolatedModel.posterior(uncertainDisplacements)
  const double wt = 1.0 / (prm.tangential_noise * prm.tangential_noise);
  const double kappa = 1.0 / (prm.noise_along_normal * prm.noise_along_normal) - wt;
  int splits = 1;
  if (side_factor_pending) {  // the partials and the factor scratch are still being read / written over there
    if (!side) HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));  // (on the side stream itself: stream order)
    side_factor_pending = false;
  }
  if (side_asm_pending) {
    HIP_OK(hipStreamWaitEvent(fs, c.ev_asm, 0));
    side_asm_pending = false;
  }
  side_parts = nullptr;
  side_parts_entry = nullptr;
  double* parts = mpart_for_write(0, fs);
  launch_regression(fs, K, r, c.Q.p, e.corr(), wt, kappa, parts, &splits);
  if (side) {  // what the state's slot now holds of this front (surface points, distances, nearest vertices of ids 0..K) is complete
    HIP_OK(hipEventRecord(c.ev_front, side));
    c.front_on_side = true;
    c.front_side_K = prm.direction == ICP_TARGET_SAMPLING ? 0 : K;
  }
  PosteriorFactorIO io{parts, splits, e.M.p, e.alpha.p, status.p + e.status_off, fscratch.p};
  // the Cholesky-root sampler at ranks above 64 (below, k_posterior_root runs where the decomposition would): the factorisation
  // itself hands the factor out — V := L, S := 1/diag(L) — and nothing is decomposed at all
  const bool root_here = sampler == ICP_SAMPLER_CHOLESKY_ROOT && !eigen_speculation_supported(r);
  if (root_here) { io.Lout = e.V.p; io.Sout = e.S.p; }
  issue_factor(e, io, parts, splits, side, root_here);  // (behind the regression in stream order, on either stream)
  return e;
}

// the one-workgroup part of a posterior — sum of the split-K partials, factorisation — on `side` (the caller has made it wait
// for the regression) or on the context stream
void icp_proposal::issue_factor(PosteriorEntry& e, PosteriorFactorIO io, double* parts, int splits, hipStream_t side, bool root_here) {
  icp_ctx& c = *ctx;
  const int r = c.r;
  if (side) {
    launch_sum_partials(side, r, parts, splits);
    io.splits = 1;
    HIP_OK(hipEventRecord(c.ev_sum, side));
    side_parts = parts;
    side_parts_entry = &e;
    launch_posterior_factor(side, r, 1, &io);
    HIP_OK(hipEventRecord(c.ev_side, side));
    side_factor_pending = true;
  } else {
    launch_posterior_factor(c.stream, r, 1, &io);
  }
  if (root_here) {  // "decomposed" as soon as the factorisation is through: an event behind it stands for the basis
    if (!e.eig_done) HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
    HIP_OK(hipEventRecord(e.eig_done, side ? side : c.stream));
    e.eig_done_shared = nullptr; e.eig_shared_gen = nullptr;
    e.eig_event_valid = true;
    e.done_value = 0;
    e.eig_valid = true;
    e.eig_checked = false;
    h_eig[e.status_off / 3] = 0;
  }
}

void sync_eigen(icp_ctx& c);

double* icp_proposal::mpart_for_write(int half, hipStream_t st) {
  if (PosteriorEntry* rd = mpart_reader[half]) {
    // a cancelled reader (eig_valid withdrawn) may read anything; a finished one has left its status in pinned memory
    // (-1 while in flight): the wait — an API call per step otherwise — is only enqueued for a kept one still at work
    const bool at_work = rd->eig_valid && *(volatile int*)(h_eig + rd->status_off / 3) == -1;
    if (at_work) {
      if (rd->eigen_event()) HIP_OK(hipStreamWaitEvent(st, rd->eigen_event(), 0));
      else sync_eigen(*ctx);  // (no event on record: wait on the host — four steps behind, never seen in practice)
    }
    mpart_reader[half] = nullptr;
  }
  return Mpart.p + (size_t)half * mpart_half_doubles;
}

// the request of the (ordinary) decomposition of `e`; the caller launches it on the context's eigen stream behind an
// ev_ready wait and records e.eig_done
void icp_proposal::prepare_eigen(PosteriorEntry& e, EigenRequest* rq) {
  if (!e.eig_done) HIP_OK(hipEventCreateWithFlags(&e.eig_done, hipEventDisableTiming));
  // the kernel reads all of Vwarm before it writes V, so the two may be the same buffer (a reused memo entry)
  h_eig[e.status_off / 3] = -1;  // in flight; the decomposition stores its status here when it ends
  e.done_value = ++eig_seq;
  // Every decomposition inherits the (tiny) deviation from orthogonality of the basis it starts from and adds that of its own
  // first-order correction (<= 1e-11): every 128th starts cold, from the identity, which puts an end to the accumulation.
  if (((eig_seq + 1) & 127) == 0) warm_valid = false;
  *rq = EigenRequest{e.M.p, warm_valid ? warm_ptr : nullptr, e.V.p, e.Vt.p, e.S.p, work.p, status.p + e.status_off + 2, nullptr,
                     h_eig + e.status_off / 3, eig_words.p + e.status_off / 3, e.done_value, ctx->sqrt_lambda.p};
  rq->root = sampler == ICP_SAMPLER_CHOLESKY_ROOT;
  if (rq->root) rq->Vwarm = nullptr;
  warm_ptr = e.V.p;
  warm_valid = true;
  e.eig_valid = true;
}

// the eigen stream of the batch whose first chain lives on `owner`, out of the launch context's pool
hipStream_t batch_eigen_stream(icp_ctx& lead, const void* owner, int second = 0) {
  // A slot's streams are made when the slot is first handed out — the second one only at ranks above 64, where a wide step's
  // decompositions alternate between two —, not all eight with the first batch: most launch contexts carry one batch at a time, a
  // stream takes 3.3 ms to make, and every stream more makes it likelier that two of them share one of the runtime's hardware queues.
  auto slot = [&](int k) {
    if (!lead.batch_eig[k]) {
      std::lock_guard<std::mutex> lk(g_eig_streams_mu);
      lead.batch_eig[k] = take_stream(lead.device, false, 0);
      g_eig_streams.insert(lead.batch_eig[k]);
      if (lead.r > 64) {  // (the two made one after the other: neighbours among the hardware queues)
        lead.batch_eig2[k] = take_stream(lead.device, false, 0);
        g_eig_streams.insert(lead.batch_eig2[k]);
      }
    }
    return (second && lead.batch_eig2[k]) ? lead.batch_eig2[k] : lead.batch_eig[k];
  };
  for (int k = 0; k < icp_ctx::kBatchRing; ++k)
    if (lead.batch_eig_owner[k] == owner) return slot(k);
  for (int k = 0; k < icp_ctx::kBatchRing; ++k)
    if (!lead.batch_eig_owner[k]) { lead.batch_eig_owner[k] = owner; return slot(k); }
  const int k = (lead.batch_eig_evict = (lead.batch_eig_evict + 1) % icp_ctx::kBatchRing);  // (more batches than the ring holds — not through icp_chain_step_batched_issue, which refuses them: shared)
  lead.batch_eig_owner[k] = owner;
  return slot(k);
}

// waits for every decomposition of this context that may still be running
void sync_eigen(icp_ctx& c) {
  if (c.eig_last && c.eig_last != c.eig_stream.peek()) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(c.eig_last)) HIP_OK(hipStreamSynchronize(c.eig_last));
  }
  if (c.eig_last2 && c.eig_last2 != c.eig_stream2.peek()) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(c.eig_last2)) HIP_OK(hipStreamSynchronize(c.eig_last2));
  }
  c.eig_stream.sync();
  c.eig_stream2.sync();
}
// the stream the next decompositions of this context go to (see g_eig_streams)
hipStream_t eigen_stream_for(icp_ctx& c, hipStream_t want) {
  if (c.eig_last && c.eig_last != want) sync_eigen(c);
  c.eig_last = want;
  return want;
}
// the wide step's pair of eigen streams (the second one only carries decompositions that use the proposals' second work buffer)
void eigen_streams_for(icp_ctx& c, hipStream_t e0, hipStream_t e1) {
  if ((c.eig_last && c.eig_last != e0) || (c.eig_last2 && c.eig_last2 != e1)) sync_eigen(c);
  c.eig_last = e0;
  c.eig_last2 = e1;
}

void icp_proposal::ensure_eigen(PosteriorEntry& e) {
  if (e.eig_valid) return;
  icp_ctx& c = *ctx;
  EigenRequest rq;
  prepare_eigen(e, &rq);
  HIP_OK(hipEventRecord(c.ev_ready, c.stream));  // M of this entry may still be in flight on the context stream
  const hipStream_t es = eigen_stream_for(c, c.eig_stream.get());
  HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
  if (!launch_posterior_eigen_pair(es, c.r, c.sqrt_lambda.p, 1, &rq)) {  // ranks > 64: no completion word
    e.done_value = 0;
    launch_posterior_eigen(es, c.r, rq.M, c.sqrt_lambda.p, rq.Vwarm, rq.V, rq.Vt, rq.S, rq.work, rq.status, nullptr, rq.host_status);
  }
  HIP_OK(hipEventRecord(e.eig_done, es));
  e.eig_done_shared = nullptr;
  e.eig_shared_gen = nullptr;
  e.eig_event_valid = true;
}

void icp_proposal::ensure_eigen_on(PosteriorEntry& e, hipStream_t es, int part) {
  icp_ctx& c = *ctx;
  if (part != 2) {
    if (e.eig_valid) return;
    prepare_eigen(e, &pending_rq);
    if (es && es == c.eig_stream2.peek()) pending_rq.work = work2.p;  // (allocated and zeroed with the proposal: a memset issued here could land in the kernels)
    e.done_value = 0;
    pending_entry = &e;
  } else if (pending_entry != &e) {
    return;  // (part 1 found the basis on record: nothing was started)
  }
  const EigenRequest& rq = pending_rq;
  launch_posterior_eigen(es, c.r, rq.M, c.sqrt_lambda.p, rq.Vwarm, rq.V, rq.Vt, rq.S, rq.work, rq.status, nullptr, rq.host_status, part);
  if (part == 1) return;
  pending_entry = nullptr;
  HIP_OK(hipEventRecord(e.eig_done, es));
  e.eig_done_shared = nullptr;
  e.eig_shared_gen = nullptr;
  e.eig_event_valid = true;
}

void icp_proposal::await_eigen(PosteriorEntry& e) {
  if (e.eigen_event()) HIP_OK(hipStreamWaitEvent(ctx->stream, e.eigen_event(), 0));
  else if (e.eig_valid) sync_eigen(*ctx);  // started by a chain step without an event: wait on the host
}

// ready / ready_seq: the word the regression launch that fills the current half of Mpart raises when it is done — the
// decomposition waits for it on the device (an event between that launch and the next one on the context stream would
// hold the latter back by several µs)
void icp_proposal::speculate_eigen(PosteriorEntry& e, const PosteriorEntry& cur, int splits, int half, const int* ready, int ready_seq,
                                   EigenSpec* spec_out, EigenRequest* rq_out) {
  e.eig_event_valid = false;
  ++spec_seq;
  *spec_out = EigenSpec{splits, h_cancel + (spec_seq & 15), spec_seq, ready, ready_seq, ctx->profiling ? ctx->d_wait_ticks.p + 1 : nullptr};
  // warm start: the basis of the current state's posterior (complete, or ahead of this launch on the same stream)
  const double* warm = (eig_seq & 127) == 127 ? nullptr : (cur.eig_valid ? cur.V.p : (warm_valid ? warm_ptr : nullptr));  // (see prepare_eigen)
  h_eig[e.status_off / 3] = -1;  // in flight
  e.done_value = ++eig_seq;
  *rq_out = EigenRequest{Mpart.p + (size_t)half * mpart_half_doubles, warm, e.V.p, e.Vt.p, e.S.p, work.p, status.p + e.status_off + 2,
                         spec_out, h_eig + e.status_off / 3, eig_words.p + e.status_off / 3, e.done_value};
  rq_out->root = sampler == ICP_SAMPLER_CHOLESKY_ROOT;
  mpart_reader[half] = &e;
  e.eig_valid = true;
  e.eig_checked = false;
  spec_entry = &e;
}

// the caller's next current state decides the fate of the decomposition started for the last proposed state
void icp_proposal::resolve_speculation(const double* theta_cur) {
  if (!spec_entry) return;
  PosteriorEntry& e = *spec_entry;
  spec_entry = nullptr;
  // A decomposition that gave up waiting for its input (k_posterior_eigen_rr) has said so in its pinned status.  That
  // happens when the runtime puts its stream on a hardware queue ahead of the launch it waits for — many streams in one
  // process, or a tool that serialises kernels — and each occurrence stalls the step for the time-out: once is enough.
  if (h_eig[e.status_off / 3] == kEigenGaveUp && !ctx->speculation_off) {
    ctx->speculation_off = true;
    ++ctx->stats.speculation_giveups; ++g_runtime_stats.speculation_giveups;
  }
  const size_t P = 10 + (size_t)ctx->r;
  if (e.valid && e.eig_valid && std::memcmp(e.theta.data(), theta_cur, sizeof(double) * P) == 0) {
    warm_ptr = e.V.p;  // accepted: this is the basis the next decompositions start from
    warm_valid = true;
    return;
  }
  __atomic_store_n(h_cancel + (spec_seq & 15), spec_seq, __ATOMIC_RELEASE);  // rejected (or the entry was recycled meanwhile)
  e.eig_valid = false;
  e.eig_checked = false;
}

// must be called after a synchronising copy of `status` into h_status
void icp_proposal::check_status(PosteriorEntry& e) {
  const int* st = h_status.data() + e.status_off;
  if (st[0]) {
    e.valid = false;
    fail(ICP_ERR_NOT_SPD, "posterior normal equations are not positive definite (non-finite correspondences?)");
  }
  if (st[2]) {
    e.eig_valid = false;
    warm_valid = false;
    fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
  }
}

namespace {

void sync_proposal_status(icp_proposal* p) {
  icp_ctx& c = *p->ctx;
  HIP_OK(hipMemcpyAsync(p->h_status.data(), p->status.p, sizeof(int) * 3 * kPosteriorMemo, hipMemcpyDeviceToHost, c.stream));
}

void sync_proposal_status_if(icp_proposal* p, bool needed) {
  if (needed) sync_proposal_status(p);
}

// ===================================================================== evaluators

// enqueue everything logValue(theta) needs; partial results land in d_res[base .. base+8).  In two parts, so that a caller can put a
// posterior of the same state on the side stream between them (icp_chain_eval_step): the searches — with the model ids below
// `reserve` (`reserve_nnv` for their nearest target vertices) left to that posterior's own searches — and the whole target-to-model
// half first; the model-to-target reductions, which read what both streams' searches have written, behind the side stream's event.
void enqueue_eval_searches(icp_evaluator* ev, StateSlot& s, int base, int reserve = 0, int reserve_nnv = 0) {
  icp_ctx& c = *ev->ctx;
  const icp_evaluator_params& p = ev->prm;
  double* out = c.d_res.p + base;
  HIP_OK(hipMemsetAsync(out, 0, sizeof(double) * 8, c.stream));
  const bool m2t = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_TARGET_TO_MODEL;
  const bool t2m = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_MODEL_TO_TARGET;
  const int Km = p.kind == ICP_EVAL_HAUSDORFF ? c.N : p.n_model_ids;
  if (m2t) {
    c.ensure_surface_prefix(s, Km, nullptr, 0, reserve);
    if (p.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE && c.target.n_boundary > 0) {
      // (a front already on the side stream that leaves the nearest vertices of its ids to this search: wait for its surface points)
      if (c.front_on_side && s.n_nnv < c.front_side_K && reserve_nnv <= s.n_nnv) {
        HIP_OK(hipStreamWaitEvent(c.stream, c.ev_front, 0));
        c.front_on_side = false;
      }
      c.ensure_nnv_prefix(s, Km, nullptr, 0, reserve_nnv);
    }
  }
  if (t2m) {
    const int Kt = ev->Kt;
    c.ensure_model_spheres(s);
    QueryBuffers qb = c.query_scratch(Kt, c.T);
    launch_surface_query(c.stream, c.T, s.x.p, c.tris.p, s.spheres.p, Kt, ev->d_tpts, ev->hint_tri.p, qb, ev->t2m_cp.p,
                         ev->t2m_d2.p, ev->t2m_tri.p);
    if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
      launch_sum_gauss_logpdf(c.stream, Kt, ev->t2m_d2.p, p.gauss_mean, p.gauss_sigma, out + 4);  // :49-54
    } else if (p.kind == ICP_EVAL_HAUSDORFF) {
      launch_dist_max(c.stream, Kt, ev->t2m_d2.p, out + 5);
    } else {
      // Collective…Evaluator.scala:56-60: nearest MODEL-sample vertex of the surface point, tested against the
      // TARGET's boundary flags (sic); ids beyond the target's vertex count count as interior.
      const bool flags = c.target.n_boundary > 0;
      if (flags) {
        QueryBuffers qb2 = c.query_scratch(Kt, c.N);
        launch_vertex_query(c.stream, c.N, s.x.p, Kt, ev->t2m_cp.p, ev->hint_nnv.p, qb2, nullptr, ev->t2m_nnv.p);
      }
      launch_dist_stats(c.stream, Kt, ev->t2m_d2.p, flags ? c.target.boundary.p : nullptr, flags ? ev->t2m_nnv.p : nullptr,
                        c.target.V, out + 4);
    }
  }
}
void enqueue_eval_reductions(icp_evaluator* ev, StateSlot& s, int base) {
  icp_ctx& c = *ev->ctx;
  const icp_evaluator_params& p = ev->prm;
  double* out = c.d_res.p + base;
  // a posterior of this state whose searches ran on the side stream has filled the slot's ids 0..K: the reductions wait for it
  if (c.front_on_side) { HIP_OK(hipStreamWaitEvent(c.stream, c.ev_front, 0)); c.front_on_side = false; }
  const bool m2t = p.kind == ICP_EVAL_HAUSDORFF || p.mode != ICP_TARGET_TO_MODEL;
  const int Km = p.kind == ICP_EVAL_HAUSDORFF ? c.N : p.n_model_ids;
  if (!m2t) return;
  if (s.n_surf < Km) fail(ICP_ERR_DEVICE, "internal: the evaluator's model ids were not all searched");
  if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
    launch_sum_gauss_logpdf(c.stream, Km, s.surf_d2.p, p.gauss_mean, p.gauss_sigma, out + 0);  // IndependentPointDistanceEvaluator.scala:40-46
  } else if (p.kind == ICP_EVAL_HAUSDORFF) {
    launch_dist_max(c.stream, Km, s.surf_d2.p, out + 1);  // (finish_eval reads the maxima only: res[1], res[5])
  } else {
    const bool flags = c.target.n_boundary > 0;  // Collective…Evaluator.scala:44-48
    if (flags && s.n_nnv < Km) fail(ICP_ERR_DEVICE, "internal: the evaluator's nearest vertices were not all searched");
    launch_dist_stats(c.stream, Km, s.surf_d2.p, flags ? c.target.boundary.p : nullptr, flags ? s.surf_nnv.p : nullptr,
                      c.target.V, out + 0);
  }
}
void enqueue_eval(icp_evaluator* ev, StateSlot& s, int base) {
  enqueue_eval_searches(ev, s, base);
  enqueue_eval_reductions(ev, s, base);
}

double gauss_logpdf(double x, double mu, double sigma) {  // Breeze Gaussian.logPdf
  double d = (x - mu) / sigma;
  return -d * d / 2.0 - (std::log(std::sqrt(2.0 * M_PI)) + std::log(sigma));
}
double expo_logpdf(double x, double rate) { return -rate * x + std::log(rate); }  // Breeze Exponential.logPdf

// combine the partial reductions exactly as the reference's computeLogValue does
int finish_eval(const icp_evaluator* ev, const double* res, double* value, double* aux) {
  const icp_evaluator_params& p = ev->prm;
  int status = ICP_OK;
  aux[0] = aux[1] = aux[2] = aux[3] = 0.0;
  if (p.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE) {
    double m2t = res[0], t2m = res[4];
    aux[0] = m2t; aux[1] = t2m;
    *value = p.mode == ICP_MODEL_TO_TARGET ? m2t : p.mode == ICP_TARGET_TO_MODEL ? t2m : 0.5 * m2t + 0.5 * t2m;  // :60-64
  } else if (p.kind == ICP_EVAL_HAUSDORFF) {
    double hd = std::max(res[1], res[5]);
    aux[0] = hd; aux[1] = res[1]; aux[2] = res[5];
    *value = expo_logpdf(hd, p.exp_rate);  // HausdorffDistanceEvaluator.scala:33-34
  } else {
    double a, h;
    const double a0 = res[0] / res[2], h0 = res[1], a1 = res[4] / res[6], h1 = res[5];
    if (p.mode == ICP_MODEL_TO_TARGET) { a = a0; h = h0; if (res[2] == 0.0) status = ICP_ERR_EMPTY; }
    else if (p.mode == ICP_TARGET_TO_MODEL) { a = a1; h = h1; if (res[6] == 0.0) status = ICP_ERR_EMPTY; }
    else {
      a = 0.5 * a0 + 0.5 * a1; h = std::max(h0, h1);  // :71-75
      if (res[2] == 0.0 || res[6] == 0.0) status = ICP_ERR_EMPTY;
    }
    aux[0] = a; aux[1] = h; aux[2] = res[2]; aux[3] = res[6];
    *value = gauss_logpdf(a, p.gauss_mean, p.gauss_sigma) + expo_logpdf(h, p.exp_rate);  // :77
  }
  if (status == ICP_OK && std::isnan(*value)) status = ICP_ERR_NOT_FINITE;
  return status;
}

icp_evaluator::Memo* eval_lookup(icp_evaluator* ev, const double* theta) {
  const size_t P = 10 + (size_t)ev->ctx->r;
  for (auto& m : ev->memo)
    if (m.valid && std::memcmp(m.theta.data(), theta, sizeof(double) * P) == 0) {
      m.stamp = ++ev->clock;
      return &m;
    }
  return nullptr;
}

icp_evaluator::Memo* eval_store(icp_evaluator* ev, const double* theta) {
  const size_t P = 10 + (size_t)ev->ctx->r;
  icp_evaluator::Memo* lru = &ev->memo[0];
  for (auto& m : ev->memo) {
    if (!m.valid) { lru = &m; break; }
    if (m.stamp < lru->stamp) lru = &m;
  }
  lru->theta.assign(theta, theta + P);
  lru->valid = true;
  lru->stamp = ++ev->clock;
  return lru;
}

bool pose_equal(const double* a, const double* b) {  // NonRigidIcpProposal.scala:72: everything but the shape must match
  for (int i = 0; i < 10; ++i)
    if (a[i] != b[i]) return false;
  return true;
}

}  // namespace
