// abi_wide.inl — part of icp_abi.hip (one translation unit; included there, in order).
// the wide step's host side (kernels_wide.hip)
// ===================================================================== the wide step (kernels_wide.hip; icp_kernels.hpp "the wide step")
// One Metropolis–Hastings step of the configurations the five merged launches do not cover — reference:
// apps/bfm/BfmFittingPartial.scala:62-83 (open target, boundary-aware ModelSampling, collective / full-mesh Hausdorff evaluator, rank
// 200, pose walks) — without a host round trip inside the step and for B chains per launch sequence.  Host side per chain: the
// choices enqueue_front makes (posterior entries of the current and of the proposed state, the proposed state's slot), the arguments
// of every launch; then ONE sequence of launches for all chains on the launch context's stream, the one-workgroup kernels
// (factorisation, tails) on its second stream, the proposed states' decompositions on the batch's eigen stream.
namespace {

struct WideItem {
  bool on = false;          // the chain takes the wide step of this ticket
  bool shape_only = false;  // only the shape differs between the current and the proposed state (transition densities exist)
  bool do_post = false;     // the proposed state's posteriors are computed (always for a shape move; ahead, for a pose move)
  bool do_spec = false;     // … and decomposed ahead
  bool eigen_first_use = false;
  StateSlot* s = nullptr;
  PosteriorEntry* ec[2] = {nullptr, nullptr};
  PosteriorEntry* ep[2] = {nullptr, nullptr};
  int Ksurf = 0, Knnv = 0;
  bool spheres = false;     // the new instance's bounding spheres are made (a target -> model search)
  int seq = 0;
  int n_tails = 0;
  TransitionTailIO tails[4];  // fwd_0, bwd_0, fwd_1, bwd_1 (as recorded: for the rare direct-tail fall-back)
};

// the rank-dependent kernels of a wide step exist for this rank and sampler
bool wide_rank_covered(int r, int sampler) {
  if (r < 3 || r > 256) return false;
  if (sampler == ICP_SAMPLER_CHOLESKY_ROOT) return r <= kCholMaxRankAbi;
  return eigen_speculation_supported(r) || eigen_tridiag_many_supported(r);
}

// the configuration (proposal set + evaluator) is one a wide step covers
bool wide_pipeline_covers(icp_evaluator* e, int n_props, icp_proposal* const* props) {
  icp_ctx& c = *e->ctx;
  if (n_props < 1 || n_props > 2) return false;
  const icp_evaluator_params& ep = e->prm;
  const bool hd = ep.kind == ICP_EVAL_HAUSDORFF;
  const bool m2t = hd || ep.mode != ICP_TARGET_TO_MODEL, t2m = hd || ep.mode != ICP_MODEL_TO_TARGET;
  const int Km = hd ? c.N : ep.n_model_ids;
  if (m2t && Km < 1) return false;
  if (t2m && (e->Kt < 1 || c.T < 1)) return false;
  if (c.target.T < 1 || c.target.V < 1) return false;
  int n_model = 0, n_target = 0, ksurf = m2t ? Km : 0;
  for (int i = 0; i < n_props; ++i) {
    const icp_proposal* p = props[i];
    if (p->K < 1) return false;
    if (!wide_rank_covered(c.r, p->sampler)) return false;
    if (p->sampler != props[0]->sampler) return false;
    if (p->prm.direction == ICP_MODEL_SAMPLING) { ++n_model; ksurf = std::max(ksurf, p->K); }
    else {
      ++n_target;
      if ((size_t)(p->K + 8) * (size_t)kCandStride > kMaxCandidates) return false;
    }
  }
  if (n_model > 1 || n_target > 1) return false;
  if ((size_t)(ksurf + 8) * (size_t)kCandStride > kMaxCandidates) return false;
  if (t2m && (size_t)(e->Kt + 8) * (size_t)kCandStride > kMaxCandidates) return false;
  return true;
}

// … and so is this call (a proposed state the caches already know has nothing to compute: the per-stage entry points answer it)
bool wide_chain_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in) {
  if (!wide_pipeline_covers(e, n_props, props)) return false;
  if (generator < 0) {
    icp_ctx& c = *e->ctx;
    if (c.find_state(theta_prop_in) || eval_lookup(e, theta_prop_in)) return false;
    for (int i = 0; i < n_props; ++i)
      if (props[i]->find_entry(theta_prop_in)) return false;
  }
  return true;
}

}  // namespace

struct BatchItem {
  icp_evaluator* e = nullptr;
  icp_proposal* const* props = nullptr;
  int generator = -1;
  const double* key = nullptr;
  bool batched = false, issued = false, redo = false;
  bool wide = false;  // takes the wide step (kernels_wide.hip) instead of the five merged launches
  WideItem W;
  StepFront F;
  StepFinishArgs f{};
  std::unique_lock<std::recursive_mutex> lk;
};

struct icp_step_ticket {
  int n_chains = 0, n_props = 0, nb = 0;
  icp_ctx* lead = nullptr;
  bool counted = false;  // included in lead->tickets_in_flight
  hipStream_t finish_stream = nullptr;  // where the batch's last launch went, if not lead->stream
  hipStream_t wide_streams[2] = {nullptr, nullptr};  // the streams of the ticket's wide step, if it has one
  std::vector<BatchItem> items;
  std::vector<StepCapture> caps;
  std::vector<icp_proposal*> props;
  std::vector<const double*> theta_cur, z;
  std::vector<double*> theta_prop;
  double* log_value_prop = nullptr;
  double* fwd = nullptr;
  double* bwd = nullptr;
  int32_t* status = nullptr;
};

namespace {
void wide_release(BatchItem& it, bool recorded = false);
// whatever happened, nothing stays reserved or locked; a failed batch leaves its launches to drain
void batch_release(icp_step_ticket& t) {
  if (t.counted && t.lead) { --t.lead->tickets_in_flight; t.counted = false; }
  for (auto& it : t.items) {
    if ((!it.batched && !it.wide) || !it.e) continue;
    icp_ctx& c = *it.e->ctx;
    if (!it.lk.owns_lock()) it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
    if (it.issued) {
      (void)hipSetDevice(c.device);
      if (t.lead) (void)hipStreamSynchronize(t.lead->stream);
      if (t.finish_stream) (void)hipStreamSynchronize(t.finish_stream);
      for (hipStream_t ws : t.wide_streams)
        if (ws) (void)hipStreamSynchronize(ws);
      if (it.wide) wide_release(it);
      else release_front(it.F);
      it.issued = false;
    } else if (it.wide) {
      wide_release(it);
    }
    c.batch_busy = false;
    it.lk.unlock();
  }
}
}  // namespace

// ---------------------------------------------------------------------------------------------------------------------
// the wide step's host side: wide_issue (everything onto the device) / wide_collect (results, bookkeeping), called by
// icp_chain_step_batched_issue / _collect for the items marked `wide`
namespace {

void wide_release(BatchItem& it, bool recorded) {  // gives back what the item holds (recorded: its step has been booked)
  WideItem& w = it.W;
  if (w.s) w.s->reserved = false;
  for (int i = 0; i < 2; ++i)
    if (w.ep[i]) w.ep[i]->reserved = false;
  if (!recorded && w.on && w.eigen_first_use && it.generator >= 0 && w.ec[it.generator]) w.ec[it.generator]->eig_checked = false;
  w.on = false; w.s = nullptr;
  w.ec[0] = w.ec[1] = w.ep[0] = w.ep[1] = nullptr;
}

// `S` waits for the decomposition `en` may still be the subject of (an event), or the host does (none on record)
void wide_await_entry(icp_ctx& c, icp_proposal* p, PosteriorEntry& en, hipStream_t S, std::vector<hipEvent_t>& waited) {
  // (a finished decomposition has left its status in pinned memory, −1 while in flight: nothing to wait for then)
  if (!en.eig_valid || *(volatile int*)(p->h_eig + en.status_off / 3) != -1) return;
  if (hipEvent_t ev = en.eigen_event()) {
    if (std::find(waited.begin(), waited.end(), ev) == waited.end()) {
      HIP_OK(hipStreamWaitEvent(S, ev, 0));
      waited.push_back(ev);
    }
  } else if (en.eig_valid && (en.eig_event_valid || en.done_value != 0)) {
    sync_eigen(c);  // (started by another kind of step without an event, or its event slot has been handed out again)
  }
}

// What wide_issue has worked out for its chains, handed back INSTEAD of being launched: the on-device loop (abi_device_loop.inl)
// uploads the records once and replays the launches step after step, the per-step entries (proposal inputs, poses) set by its own
// front kernel.  Every chain's step is captured as an ICP move (shape only, posteriors of the proposed state computed, no
// decomposition ahead).
struct WideCapture {
  WideLaunchPlan plan{};
  bool any_split = false;
  std::vector<WideChainArgs> chain_args;
  std::vector<WideProposeItem> prop_items;
  std::vector<double*> sum_parts; std::vector<int> sum_splits;
  std::vector<PosteriorFactorIO> factors;  // n_props per chain
  std::vector<TransitionTailIO> tails;     // 2·n_props per chain: fwd_0, bwd_0, fwd_1, bwd_1
  // in: a current state whose posterior is not on record gets an EMPTY entry instead of being computed the per-stage way (2.5 ms per
  // chain at rank 200, one chain after the other) — the caller fills it itself, for all chains at once (out: whether any chain has one)
  bool allow_unfilled = false, any_unfilled = false;
};

void wide_issue(icp_step_ticket& t, icp_ctx& lead, icp_ctx& elead, WideCapture* capture = nullptr) {
  const int n_props = t.n_props;
  std::vector<int> idx;
  for (int b = 0; b < t.n_chains; ++b)
    if (t.items[b].wide) idx.push_back(b);
  if (idx.empty()) return;
  const int r = elead.r, nW = (int)idx.size();
  std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);  // (its streams, its record ring)
  const hipStream_t S = lead.stream, S2 = lead.front_stream.get();
  // Two eigen streams: the decompositions of a chain's consecutive steps alternate between them (and between the proposal's two work
  // buffers), so that one started ahead for a state that was then not kept — or whose successor was a pose move that did not wait
  // for it — does not hold the next one back: at rank 200 a decomposition takes 0.6-0.7 ms, a step that does not wait for one half
  // of that.  A lone chain uses its context's own pair (made with the context, on hardware queues of their own), a batch the launch
  // context's pool.  (Ranks <= 64: one stream — the Jacobi iteration is warm-started from the decomposition before it.)
  const bool two_eig = eigen_tridiag_many_supported(r) && elead.eig_stream2.armed();
  const bool lone = t.n_chains == 1 && &lead == &elead;
  hipStream_t Es[2];
  Es[0] = lone ? elead.eig_stream.get() : batch_eigen_stream(lead, &elead, 0);
  Es[1] = !two_eig ? Es[0] : (lone ? elead.eig_stream2.get() : batch_eigen_stream(lead, &elead, 1));
  const int turn = capture ? lead.wide_turn : (lead.wide_turn = (lead.wide_turn + 1) % icp_ctx::kBatchRing);
  if (!capture) {
    Bound _b(&lead, true, true);
    if (!lead.ev_wide_sum[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_sum[turn], hipEventDisableTiming));
    if (!lead.ev_wide_fac[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_fac[turn], hipEventDisableTiming));
    if (!lead.ev_wide_head[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_head[turn], hipEventDisableTiming));
    if (!lead.ev_wide_eval[turn]) HIP_OK(hipEventCreateWithFlags(&lead.ev_wide_eval[turn], hipEventDisableTiming));
    const size_t bytes = wide_batch_bytes(nW);
    if (bytes > lead.wide_bytes[turn]) {
      // (the slot's previous reader was the batch kBatchRing tickets ago: collected — tickets_in_flight —, its launches finished)
      if (lead.wide_pinned[turn]) { pinned_free(lead.wide_pinned[turn]); lead.wide_pinned[turn] = nullptr; }
      const size_t cap = std::max(bytes, wide_batch_bytes(kWideMaxChains));
      pinned_alloc((void**)&lead.wide_pinned[turn], cap);
      lead.wide_device[turn].alloc(cap);
      lead.wide_bytes[turn] = cap;
    }
  }
  struct SearchHintScope { SearchHintScope(int n) { search_chains_hint(n); } ~SearchHintScope() { search_chains_hint(1); } } search_hint_scope(nW);
  std::vector<hipEvent_t> waited;
  std::vector<WideProposeItem> prop_items;
  std::vector<WideChainArgs> chain_args(nW);
  std::vector<double*> sum_parts; std::vector<int> sum_splits;
  std::vector<PosteriorFactorIO> factors;
  std::vector<PosteriorEntry*> root_entries;  // (Cholesky-root sampler above rank 64: the factorisation hands the "basis" out)
  std::vector<icp_proposal*> root_props;
  std::vector<TransitionTailIO> tails;
  std::vector<EigenRequest> spec_rq[2]; std::vector<const double*> spec_parts[2]; std::vector<PosteriorEntry*> spec_entries[2];
  std::vector<EigenRequest> pre_rq[2]; std::vector<PosteriorEntry*> pre_entries[2];
  std::vector<WideDoneItem> dones;
  WideLaunchPlan plan{};
  plan.B = nW; plan.N = elead.N; plan.r = r; plan.Qp = elead.Qp.p; plan.ref = elead.ref.p; plan.mean = elead.mean.p;
  plan.f1_prepared = true;
  bool any_split = false;
  int inst_head_points = 0;  // (INT_MAX: some chain's main sequence reads the whole instance)
  const bool concurrent = t.items[idx[0]].e->prm.kind == ICP_EVAL_HAUSDORFF;  // (how a split step's two sequences are scheduled: below)
  const int spec_mode = speculation_mode();

  for (int k = 0; k < nW; ++k) {
    BatchItem& it = t.items[idx[k]];
    WideItem& w = it.W;
    icp_evaluator* e = it.e;
    icp_ctx& c = *e->ctx;
    const double* theta_cur = t.theta_cur[idx[k]];
    double* theta_prop = t.theta_prop[idx[k]];
    const int generator = it.generator;
    Bound _b(&c, true, capture != nullptr);  // (a captured step belongs to a run that has claimed its contexts already)
    w = WideItem{};
    w.on = true;
    if (!e->last_prop.empty() && !capture) {  // did the caller keep the state the previous step proposed?
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur);  // (a merged step's speculation, if the chain changed paths)
    w.shape_only = generator >= 0 || pose_equal(theta_cur, theta_prop);
    const bool spec = !capture && (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1)) && !c.speculation_off;
    // a pose move changes the state too: if it is kept, the next ICP proposal draws from the posterior at the NEW state, which
    // nothing else on this path would compute (the transition densities across a pose change are −∞): started here, ahead
    w.do_post = w.shape_only || spec;
    w.do_spec = w.do_post && spec;
    const bool root = it.props[0]->sampler == ICP_SAMPLER_CHOLESKY_ROOT;
    const bool root_here = root && !eigen_speculation_supported(r);  // (the factorisation itself hands the factor out)

    // ---- cached side: the current state's posteriors (a shape move's tails and proposal read them)
    PosteriorEntry** ec = w.ec;
    PosteriorEntry** ep = w.ep;
    if (w.shape_only) {
      bool missing = false;
      for (int i = 0; i < n_props; ++i) missing = missing || !it.props[i]->find_entry(theta_cur);
      for (int i = 0; i < n_props; ++i) {
        if (capture && capture->allow_unfilled && !it.props[i]->find_entry(theta_cur)) {
          PosteriorEntry& f = it.props[i]->fresh_entry();
          f.valid = false; f.eig_valid = false; f.eig_checked = false; f.eig_event_valid = false; f.done_value = 0;
          f.reserved = true;
          ec[i] = &f;
          capture->any_unfilled = true;
          continue;
        }
        ec[i] = &it.props[i]->posterior(theta_cur, false);  // NonRigidIcpProposal.scala:54,76 (the per-stage way if not on record)
      }
      if (missing || c.stream_used_elsewhere) { HIP_OK(hipStreamSynchronize(c.stream)); c.stream_used_elsewhere = false; c.stage_used = 0; }
      for (int i = 0; i < n_props; ++i) { ec[i]->reserved = true; }  // (not to be recycled for the proposed state's entries below)
    }
    if (generator >= 0) {
      icp_proposal* pg = it.props[generator];
      PosteriorEntry& g = *ec[generator];
      if (!g.eig_valid && capture && capture->allow_unfilled) {
        // (decomposed by the caller: with the entry's posterior, if that is not on record, or ahead of the loop)
      } else if (!g.eig_valid) {
        if (root_here) fail(ICP_ERR_DEVICE, "internal: a posterior of the Cholesky-root sampler without its factor");
        EigenRequest rq;
        pg->prepare_eigen(g, &rq);
        rq.sqrt_lambda = c.sqrt_lambda.p;
        g.eig_event_valid = false; g.eig_done_shared = nullptr; g.eig_shared_gen = nullptr;
        eigen_streams_for(c, Es[0], Es[1]);
        const int fl = two_eig ? (int)(pg->eig_flip++ & 1) : 0;
        if (fl) rq.work = pg->work2.p;
        pre_rq[fl].push_back(rq);
        pre_entries[fl].push_back(&g);
      } else {
        wide_await_entry(c, pg, g, S, waited);
      }
      w.eigen_first_use = !g.eig_checked;
      g.eig_checked = true;
    }

    // ---- new side: one state slot, one memo entry per proposal
    StateSlot* same = nullptr;  // a pose move: a state with these coefficients whose deformations are kept
    if (generator < 0 && !w.shape_only)
      for (auto& o : c.slots)
        if (o.valid && o.defo_valid && std::memcmp(o.theta.data() + 10, theta_prop + 10, sizeof(double) * r) == 0) { same = &o; break; }
    if (same) { same->stamp = ++c.clock; same->reserved = true; }
    StateSlot& s = c.fresh_state();
    if (same) same->reserved = false;
    s.reserved = true;
    w.s = &s;
    s.pose = c.pose_of(generator >= 0 ? theta_cur : theta_prop);
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) {
        ep[i] = &it.props[i]->fresh_entry();
        ep[i]->reserved = true;
        wide_await_entry(c, it.props[i], *ep[i], S, waited);  // (a decomposition started ahead for a state that was not kept may still read / write it)
        ep[i]->eig_event_valid = false;
        ep[i]->done_value = 0;
      }
    if (w.shape_only)
      for (int i = 0; i < n_props; ++i) ec[i]->reserved = false;

    // ---- W1: coefficients of the proposed state
    WideProposeItem pi{};
    if (generator >= 0) {
      PosteriorEntry& g = *ec[generator];
      if (!c.h_wide_z) pinned_alloc((void**)&c.h_wide_z, sizeof(double) * kMaxRank);
      std::memcpy(c.h_wide_z, t.z[idx[k]], sizeof(double) * r);  // posterior.sample()'s standard normals (:55)
      pi.kind = 1;
      pi.in = ProposeIn{g.alpha.p, g.V.p, g.S.p, c.inv_sqrt_lambda.p, c.P.p, g.coeffs.p, c.h_wide_z, kSigma2,
                        it.props[generator]->prm.step_length, root ? 1 : 0};
    } else {
      if (!c.h_wide_z) pinned_alloc((void**)&c.h_wide_z, sizeof(double) * kMaxRank);
      std::memcpy(c.h_wide_z, theta_prop + 10, sizeof(double) * r);
      pi.kind = 0;
      pi.src = c.h_wide_z;
    }
    pi.n_out = 0;
    pi.out[pi.n_out++] = s.coeffs.p;
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) pi.out[pi.n_out++] = ep[i]->coeffs.p;
    pi.out[pi.n_out++] = c.h_res + 16;
    prop_items.push_back(pi);

    // ---- searches
    const icp_evaluator_params& evp = e->prm;
    const bool hd = evp.kind == ICP_EVAL_HAUSDORFF, coll = evp.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE;
    const bool ev_m2t = hd || evp.mode != ICP_TARGET_TO_MODEL, ev_t2m = hd || evp.mode != ICP_MODEL_TO_TARGET;
    const int Km = hd ? c.N : evp.n_model_ids, Kt = e->Kt;
    icp_proposal* pm = nullptr; icp_proposal* pt = nullptr;
    int im = -1, itx = -1;
    if (w.do_post)
      for (int i = 0; i < n_props; ++i) {
        if (it.props[i]->prm.direction == ICP_MODEL_SAMPLING) { pm = it.props[i]; im = i; }
        else { pt = it.props[i]; itx = i; }
      }
    const bool open_target = c.target.n_boundary > 0;
    const int Ksurf = std::max(ev_m2t ? Km : 0, pm ? pm->K : 0);
    const bool prop_nnv = pm && pm->prm.boundary_aware && open_target;      // NonRigidIcpProposal.scala:98-99
    const bool eval_nnv = coll && open_target && ev_m2t;                     // Collective…Evaluator.scala:44-48
    const bool t2m_nnv = coll && open_target && ev_t2m;                      // :56-60
    const int Knnv = std::max(prop_nnv ? pm->K : 0, eval_nnv ? Km : 0);
    require(Ksurf <= c.N, "model id count exceeds the number of model points");
    w.Ksurf = Ksurf; w.Knnv = Knnv; w.spheres = ev_t2m;
    // Split: the PROPOSAL's chain — its K model ids -> their nearest vertices -> correspondences -> regression -> partial sums: what the
    // factorisation, the tails and the decomposition wait for — is the MAIN sequence; the evaluator's searches and reductions are a
    // sequence of their own behind (or beside) it:
    //   concurrent (the batch's evaluator is the full-mesh Hausdorff distance: every model vertex against the target surface, every
    //     target vertex against the model's, 0.2 ms of chip-wide searches): the evaluator's sequence on S BESIDE the main one on the
    //     second stream; the maxima are order-independent, each sequence reduces its own range (atomic maxima);
    //   serial (any other evaluator): main first, the evaluator's sequence behind it on the same stream — the side streams
    //     (factorisation + tails, decomposition) start as soon as the partial sums exist instead of behind every search of the step
    //     (10 chains of the face configuration: ≈ 0.2 ms earlier); the reductions run at the end, over all ids.
    const bool split = w.do_post && (concurrent ? hd : true);
    const int Kp = split ? (pm ? std::min(pm->K, Ksurf) : 0) : 0;
    any_split = any_split || split;
    // (the instance launch's head: the blocks of model points the MAIN sequence reads — ids below Kp and the corners of their triangles)
    if (capture && split && pm && !pt && Kp > 0 && (size_t)Kp <= c.shared_model->ring_prefix_max.size())
      inst_head_points = std::max(inst_head_points, c.shared_model->ring_prefix_max[(size_t)Kp - 1] + 1);
    else inst_head_points = INT_MAX;
    int nnv_main = Knnv, nnv_lo = 0, nnv_hi = 0;  // nearest vertices: ids [0, nnv_main) by the main sequence, [nnv_lo, nnv_hi) by the evaluator's
    if (split) {
      nnv_main = prop_nnv ? pm->K : 0;
      if (eval_nnv) { nnv_lo = prop_nnv ? std::min(pm->K, Km) : 0; nnv_hi = Km; }
    }
    QueryBuffers qs{}, qv{}, qt{}, qn{}, qtn{}, qp{}, qen{};
    if (Ksurf - Kp > 0) qs = c.query_scratch(Ksurf - Kp, c.target.T, 0);
    if (split && Kp > 0) qp = c.query_scratch(Kp, c.target.T, 5);
    if (pt) qv = c.query_scratch(pt->K, c.N, 1);
    if (ev_t2m) qt = c.query_scratch(Kt, c.T, 2);
    if (nnv_main > 0) qn = c.query_scratch(nnv_main, c.target.V, 3);
    if (nnv_hi > nnv_lo) qen = c.query_scratch(nnv_hi - nnv_lo, c.target.V, 6);
    if (t2m_nnv) qtn = c.query_scratch(Kt, c.N, 4);

    WideChainArgs& A = chain_args[k];
    std::memset(&A, 0, sizeof(A));
    SurfaceTask st_surf{}, st_t2m{}, st_surfp{};
    VertexTask st_vert{}, st_nnv{}, st_ennv{}, st_tnn{};
    if (Ksurf - Kp > 0)  // (ids Kp..Ksurf; Kp = 0 unless the evaluator has a sequence of its own)
      st_surf = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Ksurf - Kp, s.x.p + 3 * (size_t)Kp,
                                  c.hint_surf.p + Kp, qs, s.surf_cp.p + 3 * (size_t)Kp, s.surf_d2.p + Kp, s.surf_tri.p + Kp);
    if (split && Kp > 0)
      st_surfp = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Kp, s.x.p, c.hint_surf.p, qp,
                                   s.surf_cp.p, s.surf_d2.p, s.surf_tri.p);
    if (ev_t2m)
      st_t2m = make_surface_task(c.T, s.x.p, c.tris.p, s.spheres.p, Kt, e->d_tpts, e->hint_tri.p, qt, e->t2m_cp.p, e->t2m_d2.p, e->t2m_tri.p);
    if (pt) { st_vert = make_vertex_task(c.N, s.x.p, pt->K, pt->target_pts.p, pt->hint_nn.p, qv, nullptr, pt->nn_id.p); st_vert.thr2 = nullptr; }
    if (nnv_main > 0) { st_nnv = make_vertex_task(c.target.V, c.target.verts.p, nnv_main, s.surf_cp.p, c.hint_nnv.p, qn, nullptr, s.surf_nnv.p); st_nnv.thr2 = nullptr; }
    if (nnv_hi > nnv_lo) {
      st_ennv = make_vertex_task(c.target.V, c.target.verts.p, nnv_hi - nnv_lo, s.surf_cp.p + 3 * (size_t)nnv_lo, c.hint_nnv.p + nnv_lo, qen, nullptr,
                                 s.surf_nnv.p + nnv_lo);
      st_ennv.thr2 = nullptr;
    }
    if (t2m_nnv) { st_tnn = make_vertex_task(c.N, s.x.p, Kt, e->t2m_cp.p, e->hint_nnv.p, qtn, nullptr, e->t2m_nnv.p); st_tnn.thr2 = nullptr; }

    // W2
    A.inst.kind = same ? 1 : 0;
    A.inst.coeffs = s.coeffs.p;
    A.inst.defo_src = same ? same->defo.p : nullptr;
    A.inst.pose = s.pose;
    A.inst.x = s.x.p; A.inst.defo = s.defo.p;
    if (split) {
      A.inst.has_surf = Kp > 0 ? 1 : 0; A.inst.surf = st_surfp;
      A.inst.has_surf2 = Ksurf - Kp > 0 ? 1 : 0; A.inst.surf2 = st_surf;
    } else {
      A.inst.has_surf = Ksurf > 0 ? 1 : 0; A.inst.surf = st_surf;
    }
    // W3
    A.prep.T = ev_t2m ? c.T : 0; A.prep.x = s.x.p; A.prep.tris = c.tris.p; A.prep.order = c.tri_order.p; A.prep.spheres = s.spheres.p;
    A.prep.has_t2m = ev_t2m ? 1 : 0; A.prep.t2m = st_t2m;
    A.prep.n_cnt = 0;
    auto reset_cnt = [&](const VertexTask& v) { A.prep.cnt[A.prep.n_cnt] = v.cnt; A.prep.cnt_n[A.prep.n_cnt++] = v.Kpad; };
    if (pt) reset_cnt(st_vert);
    if (nnv_main > 0) reset_cnt(st_nnv);
    if (nnv_hi > nnv_lo) reset_cnt(st_ennv);
    if (t2m_nnv) reset_cnt(st_tnn);
    A.prep.zero_d = c.d_res.p; A.prep.n_zero_d = 8;
    // ---- the search sequences: tasks are appended to a StepSearchArgs (surface tasks first)
    struct Seq { StepSearchArgs* q; int nt = 0, n_corr = 0; };
    auto seq_init = [](StepSearchArgs& q) { q.s_corr[0] = q.s_corr[1] = q.v_corr[0] = q.v_corr[1] = -1; q.fstart[0] = 0; q.rstart[0] = 0; return Seq{&q}; };
    auto add_surface = [&](Seq& sq, const SurfaceTask& t, const CorrTask* corr) {
      StepSearchArgs& q = *sq.q;
      q.s[q.n_surf] = t;
      q.fstart[sq.nt + 1] = q.fstart[sq.nt] + filter_grid_blocks(t.tblocks, t.ksplit);
      q.rstart[sq.nt + 1] = q.rstart[sq.nt] + t.K;
      if (corr) { q.corr[sq.n_corr] = *corr; q.s_corr[q.n_surf] = sq.n_corr++; }
      ++q.n_surf; ++sq.nt;
    };
    auto add_vertex = [&](Seq& sq, const VertexTask& t, const CorrTask* corr) {
      StepSearchArgs& q = *sq.q;
      q.v[q.n_vert] = t;
      q.fstart[sq.nt + 1] = q.fstart[sq.nt] + filter_grid_blocks(t.vblocks, t.ksplit);
      q.rstart[sq.nt + 1] = q.rstart[sq.nt] + t.K;
      if (corr) { q.corr[sq.n_corr] = *corr; q.v_corr[q.n_vert] = sq.n_corr++; }
      ++q.n_vert; ++sq.nt;
    };
    auto seq_close = [](Seq& sq) { for (int u = sq.nt + 1; u < 5; ++u) { sq.q->fstart[u] = sq.q->fstart[sq.nt]; sq.q->rstart[u] = sq.q->rstart[sq.nt]; } };
    CorrTask corr_m{}, corr_t{};
    if (pm) corr_m = CorrTask{pm->K, ep[im]->corr(), s.x.p, nullptr, c.target.boundary.p, nullptr, pm->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, prop_nnv ? s.surf_cp.p : nullptr};
    if (pt) corr_t = CorrTask{pt->K, ep[itx]->corr(), s.x.p, pt->target_pts.p, c.boundary.p, nullptr, pt->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p, nullptr};
    // main, stage 1 (W4/W5) and stage 2 (W6/W7: nearest vertices of the surface points, ModelSampling correspondences with their flag)
    Seq m1 = seq_init(A.s1), m2 = seq_init(A.s2), e1 = seq_init(A.s1b), e2 = seq_init(A.s2b);
    const SurfaceTask& st_first = split ? st_surfp : st_surf;
    if (st_first.K > 0) add_surface(m1, st_first, (pm && !prop_nnv) ? &corr_m : nullptr);
    if (ev_t2m && !split) add_surface(m1, st_t2m, nullptr);
    if (pt) add_vertex(m1, st_vert, &corr_t);
    if (nnv_main > 0) add_vertex(m2, st_nnv, prop_nnv ? &corr_m : nullptr);
    if (t2m_nnv && !split) add_vertex(m2, st_tnn, nullptr);
    // the evaluator's own sequence (split): the model ids behind the proposal's, the target -> model direction, their nearest vertices
    if (split) {
      if (st_surf.K > 0) add_surface(e1, st_surf, nullptr);
      if (ev_t2m) add_surface(e1, st_t2m, nullptr);
      if (nnv_hi > nnv_lo) add_vertex(e2, st_ennv, nullptr);
      if (t2m_nnv) add_vertex(e2, st_tnn, nullptr);
    }
    seq_close(m1); seq_close(m2); seq_close(e1); seq_close(e2);
    // W8: regressions + the likelihood's reductions
    StepRegressionArgs& g = A.reg.reg;
    g.n = w.do_post ? n_props : 0; g.r = r; g.ntiles = regression_tiles(r); g.Q = c.Q.p;
    g.ustart[0] = 0; g.ustart[1] = 0; g.ustart[2] = 0;
    int splits[2] = {1, 1};
    double* parts[2] = {nullptr, nullptr};
    for (int i = 0; i < g.n; ++i) {
      icp_proposal* p = it.props[i];
      const int leaves = regression_splits(p->K);
      g.K[i] = p->K;
      g.kchunk[i] = std::max(1, (p->K + leaves - 1) / leaves);
      g.fold[i] = regression_fold(p->K, r, nW * n_props);
      g.macro[i] = regression_macro(r, g.fold[i]);
      plan.reg_folded = plan.reg_folded || g.fold[i] > 1;
      g.X[i] = nullptr;
      if (g.fold[i] > 1 && g.macro[i] > 1) {  // the operand rows of the posterior's correspondences, made ahead of the regression launch
        const int xrs = 32 * ((r + 1 + 31) / 32);  // (whole 2 x 16-column macro blocks: the rows are stored interleaved, see StepRegressionArgs::X)
        const size_t need = (size_t)std::max(p->K, 1) * 4 * xrs;
        if (p->xrows.n < need) p->xrows.alloc(need);
        g.X[i] = p->xrows.p;
        g.xrs = xrs;
        plan.grid_xrows = std::max(plan.grid_xrows, (int)(((size_t)p->K * xrs + 255) / 256));
      }
      splits[i] = leaves / g.fold[i];
      const int units_i = regression_units(r, leaves, g.fold[i], g.macro[i]);
      g.cb[i] = ep[i]->corr();
      g.wt[i] = 1.0 / (p->prm.tangential_noise * p->prm.tangential_noise);
      g.kappa[i] = 1.0 / (p->prm.noise_along_normal * p->prm.noise_along_normal) - g.wt[i];
      if (p->side_factor_pending || p->side_asm_pending) {  // (a per-stage step of this proposal left work on its side streams)
        c.front_stream.sync(); sync_eigen(c);
        p->side_factor_pending = false; p->side_asm_pending = false;
      }
      p->side_parts = nullptr; p->side_parts_entry = nullptr;
      p->mpart_half = (p->mpart_half + 1) % icp_proposal::kMpartRing;
      parts[i] = g.Mpart[i] = p->mpart_for_write(p->mpart_half, S);
      g.status[i] = p->status.p + ep[i]->status_off;
      g.ustart[i + 1] = g.ustart[i] + units_i;
    }
    if (g.n == 1) g.ustart[2] = g.ustart[1];
    // the likelihood's reductions over ALL ids (the layout finish_eval reads) …
    WideRegArgs full{};
    full.eval_kind = evp.kind; full.eval_m2t = ev_m2t ? 1 : 0; full.eval_t2m = ev_t2m ? 1 : 0;
    full.Km = Km; full.d2m = s.surf_d2.p;
    full.flags_m = eval_nnv ? c.target.boundary.p : nullptr; full.idx_m = eval_nnv ? s.surf_nnv.p : nullptr;
    full.Kt = Kt; full.d2t = e->t2m_d2.p;
    full.flags_t = t2m_nnv ? c.target.boundary.p : nullptr; full.idx_t = t2m_nnv ? e->t2m_nnv.p : nullptr;  // (sic: SURVEY App. D5)
    full.n_flags = c.target.V;
    full.mean = evp.gauss_mean; full.sigma = evp.gauss_sigma;
    full.red_out = c.d_res.p;
    const StepRegressionArgs reg_only = g;
    if (!split) {                 // … behind every search of the one sequence
      A.reg = full; A.reg.reg = reg_only;
    } else if (concurrent) {      // … each sequence its own range (the Hausdorff maxima are order-independent: atomic maxima into one word)
      A.reg = full; A.reg.reg = reg_only;
      A.reg.eval_m2t = Kp > 0 ? 1 : 0; A.reg.Km = Kp; A.reg.eval_t2m = 0;
      A.regb = full;
      A.regb.eval_m2t = Km - Kp > 0 ? 1 : 0; A.regb.Km = Km - Kp; A.regb.d2m = s.surf_d2.p + Kp;
    } else {                      // … at the end of the evaluator's sequence, which runs behind the main one on the same stream
      A.reg = full; A.reg.reg = reg_only;
      A.reg.eval_m2t = 0; A.reg.eval_t2m = 0;
      A.regb = full;
    }
    plan.grid_prep = std::max(plan.grid_prep, wide_prep_grid(A.prep));
    plan.grid_f1 = std::max(plan.grid_f1, A.s1.fstart[m1.nt]);
    plan.grid_r1 = std::max(plan.grid_r1, A.s1.rstart[m1.nt]);
    plan.grid_f2 = std::max(plan.grid_f2, A.s2.fstart[m2.nt]);
    plan.grid_r2 = std::max(plan.grid_r2, A.s2.rstart[m2.nt]);
    plan.grid_reg = std::max(plan.grid_reg, wide_reg_blocks(A.reg));
    plan.grid_f1b = std::max(plan.grid_f1b, A.s1b.fstart[e1.nt]);
    plan.grid_r1b = std::max(plan.grid_r1b, A.s1b.rstart[e1.nt]);
    plan.grid_f2b = std::max(plan.grid_f2b, A.s2b.fstart[e2.nt]);
    plan.grid_r2b = std::max(plan.grid_r2b, A.s2b.rstart[e2.nt]);
    if (split) plan.grid_regb = std::max(plan.grid_regb, wide_reg_blocks(A.regb));

    // ---- W9..W12
    for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
    for (int i = 0; i < 16; ++i) c.h_status[i] = 0;
    w.seq = ++c.step_seq;
    WideDoneItem di{};
    di.red_src = c.d_res.p; di.red_dst = c.h_res;
    di.host_flag = c.h_flag; di.seq = w.seq;
    for (int i = 0; i < g.n; ++i) {
      icp_proposal* p = it.props[i];
      if (splits[i] > 1) { sum_parts.push_back(parts[i]); sum_splits.push_back(splits[i]); }
      PosteriorFactorIO io{parts[i], 1, ep[i]->M.p, ep[i]->alpha.p, p->status.p + ep[i]->status_off, p->fscratch.p};
      if (root_here) { io.Lout = ep[i]->V.p; io.Sout = ep[i]->S.p; root_entries.push_back(ep[i]); root_props.push_back(p); }
      factors.push_back(io);
      di.st_src[i] = p->status.p + ep[i]->status_off; di.st_dst[i] = c.h_status + 8 + i;
      if (w.shape_only) {
        w.tails[2 * i] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, ec[i]->coeffs.p, ep[i]->coeffs.p, p->prm.step_length,
                                          c.h_res + 8 + 2 * i, c.h_status + 2 * i};
        w.tails[2 * i + 1] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, ep[i]->coeffs.p, ec[i]->coeffs.p, p->prm.step_length,
                                              c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
        tails.push_back(w.tails[2 * i]); tails.push_back(w.tails[2 * i + 1]);
        w.n_tails = 2 * (i + 1);
      }
      if (w.do_spec && !root_here) {
        EigenRequest rq;
        p->prepare_eigen(*ep[i], &rq);
        rq.sqrt_lambda = c.sqrt_lambda.p;
        ep[i]->eig_checked = false;
        ep[i]->eig_event_valid = false; ep[i]->eig_done_shared = nullptr; ep[i]->eig_shared_gen = nullptr;
        p->mpart_reader[p->mpart_half] = ep[i];
        eigen_streams_for(c, Es[0], Es[1]);
        const int fl = two_eig ? (int)(p->eig_flip++ & 1) : 0;
        if (fl) rq.work = p->work2.p;
        spec_rq[fl].push_back(rq);
        spec_parts[fl].push_back(parts[i]);
        spec_entries[fl].push_back(ep[i]);
      }
    }
    dones.push_back(di);
    it.issued = true;
  }

  // ---- one sequence of launches for all of them
  Bound _b(&lead, true, true);
  for (int fl = 0; fl < 2; ++fl) {
    if (pre_rq[fl].empty()) continue;  // KL bases of current states that have none yet (a chain's first ICP proposal; speculation off)
    const hipStream_t E = Es[fl];
    if (eigen_tridiag_many_supported(r)) launch_posterior_eigen_tridiag_many(E, r, (int)pre_rq[fl].size(), pre_rq[fl].data(), nullptr);
    else
      for (auto& rq : pre_rq[fl])
        if (!launch_posterior_eigen_pair(E, r, rq.sqrt_lambda, 1, &rq)) fail(ICP_ERR_DEVICE, "internal: wide step at a rank without a decomposition kernel");
    BatchEventSlot* ev_pre = &next_batch_event(elead.device);
    HIP_OK(hipEventRecord(ev_pre->ev, E));
    HIP_OK(hipStreamWaitEvent(S, ev_pre->ev, 0));
    for (PosteriorEntry* en : pre_entries[fl]) {
      en->eig_done_shared = ev_pre->ev; en->eig_shared_gen = &ev_pre->gen; en->eig_shared_gen_value = ev_pre->gen;
      en->eig_event_valid = true; en->done_value = 0;
    }
  }
  if (capture) {  // (nothing launched; what the chains hold stays reserved until the caller's wide_release)
    // the instance launch's head: the main sequence goes ahead after these blocks of 64 points, the rest is beside it (abi_device_loop.inl)
    if (any_split && inst_head_points > 0 && inst_head_points != INT_MAX) {
      const int hb = (inst_head_points + 63) / 64, all = (plan.N + 63) / 64;
      plan.inst_head_blocks = 4 * hb <= all ? hb : 0;
    }
    capture->plan = plan; capture->any_split = any_split;
    capture->chain_args = std::move(chain_args); capture->prop_items = std::move(prop_items);
    capture->sum_parts = std::move(sum_parts); capture->sum_splits = std::move(sum_splits);
    capture->factors = std::move(factors); capture->tails = std::move(tails);
    return;
  }
  for (size_t p0 = 0; p0 < prop_items.size(); p0 += kWideMaxChains) {
    WideProposeArgs pa{};
    pa.n = (int)std::min<size_t>(kWideMaxChains, prop_items.size() - p0);
    for (int i = 0; i < pa.n; ++i) pa.it[i] = prop_items[p0 + i];
    launch_wide_propose(S, r, pa);
  }
  launch_wide_head(S, plan, chain_args.data(), lead.wide_pinned[turn], lead.wide_device[turn].p);
  // the step's searches, regressions and reductions: on `S` — or, where the evaluator's searches are a sequence of their own, those
  // on `S` and the proposals' chain (searches of their K ids, regression, then factorisation and tails) on the second stream beside them
  const hipStream_t Sm = (any_split && concurrent) ? S2 : S;
  if (any_split && concurrent) {
    HIP_OK(hipEventRecord(lead.ev_wide_head[turn], S));
    HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_head[turn], 0));
  }
  launch_wide_main(Sm, plan, lead.wide_device[turn].p);
  for (size_t p0 = 0; p0 < sum_parts.size(); p0 += kWideMaxChains)
    launch_sum_partials_many(Sm, r, (int)std::min<size_t>(kWideMaxChains, sum_parts.size() - p0), sum_parts.data() + p0, sum_splits.data() + p0);
  HIP_OK(hipEventRecord(lead.ev_wide_sum[turn], Sm));
  // the one-workgroup kernels on the second stream: the evaluator's sequence and the next batch's chip-wide launches on `S` run beside them
  if (Sm != S2) HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_sum[turn], 0));
  const bool any_spec = !spec_rq[0].empty() || !spec_rq[1].empty();
  const bool jacobi_spec = any_spec && !eigen_tridiag_many_supported(r);  // (ranks <= 64: the iteration reads the finished M)
  // The decompositions and factorisations of this step — the critical path — are handed to the device BEFORE the evaluator's own
  // sequence (chip-wide launches on `S`, which they then run beside), the factorisations in one launch (two took 200 + 880 µs in a
  // 30-chain step: the second one started among the evaluator's searches).  configs[4], 30 chains a step: 11.2k -> 12.6k it/s with
  // the tridiagonalisation's load prologue (icp_tridiag.hpp: tridiag_kernel_body), tools/r4_trace_c4.sh.
  if (any_spec && !jacobi_spec)
    for (int fl = 0; fl < 2; ++fl) {
      if (spec_rq[fl].empty()) continue;
      // the proposed states' KL bases BESIDE their factorisations: M = I + the summed partials is written at the head of the
      // decomposition as well (the values the factorisation's own assembly writes)
      HIP_OK(hipStreamWaitEvent(Es[fl], lead.ev_wide_sum[turn], 0));
      launch_posterior_eigen_tridiag_many(Es[fl], r, (int)spec_rq[fl].size(), spec_rq[fl].data(), spec_parts[fl].data());
    }
  {
    const size_t fmax = (size_t)posterior_factor_max();
    for (size_t p0 = 0; p0 < factors.size(); p0 += fmax)
      launch_posterior_factor(S2, r, (int)std::min(fmax, factors.size() - p0), factors.data() + p0);
  }
  if (any_split) {
    launch_wide_eval(S, plan, lead.wide_device[turn].p);
    HIP_OK(hipEventRecord(lead.ev_wide_eval[turn], S));
  }
  if (!root_entries.empty()) {  // "decomposed" as soon as the factorisation is through: an event behind it stands for the basis
    BatchEventSlot& done = next_batch_event(elead.device);
    HIP_OK(hipEventRecord(done.ev, S2));
    for (size_t q = 0; q < root_entries.size(); ++q) {
      PosteriorEntry* en = root_entries[q];
      en->eig_done_shared = done.ev; en->eig_shared_gen = &done.gen; en->eig_shared_gen_value = done.gen;
      en->eig_event_valid = true; en->done_value = 0; en->eig_valid = true; en->eig_checked = false;
      root_props[q]->h_eig[en->status_off / 3] = 0;
    }
  }
  if (jacobi_spec) {
    HIP_OK(hipEventRecord(lead.ev_wide_fac[turn], S2));
    HIP_OK(hipStreamWaitEvent(Es[0], lead.ev_wide_fac[turn], 0));
    for (size_t q = 0; q < spec_rq[0].size(); ++q)
      if (!launch_posterior_eigen_pair(Es[0], r, spec_rq[0][q].sqrt_lambda, 1, &spec_rq[0][q])) fail(ICP_ERR_DEVICE, "internal: wide step at a rank without a decomposition kernel");
  }
  for (int fl = 0; fl < 2; ++fl) {
    if (spec_rq[fl].empty()) continue;
    BatchEventSlot& done = next_batch_event(elead.device);
    HIP_OK(hipEventRecord(done.ev, Es[fl]));
    for (PosteriorEntry* en : spec_entries[fl]) {
      en->eig_done_shared = done.ev; en->eig_shared_gen = &done.gen; en->eig_shared_gen_value = done.gen;
      en->eig_event_valid = true; en->done_value = 0;
    }
  }
  for (size_t t0 = 0; t0 < tails.size(); t0 += 2 * kWideMaxChains)
    launch_transition_tails(S2, r, (int)std::min<size_t>(2 * kWideMaxChains, tails.size() - t0), tails.data() + t0, elead.Ginv.p, kSigma2);
  if (any_split) HIP_OK(hipStreamWaitEvent(S2, lead.ev_wide_eval[turn], 0));  // (the reductions of the evaluator's own sequence)
  for (size_t p0 = 0; p0 < dones.size(); p0 += kWideMaxChains) {
    WideDoneArgs da{};
    da.n = (int)std::min<size_t>(kWideMaxChains, dones.size() - p0);
    for (int i = 0; i < da.n; ++i) da.it[i] = dones[p0 + i];
    launch_wide_done(S2, da);
  }
  t.wide_streams[0] = S; t.wide_streams[1] = S2;
}

// results of one wide item (its flag has been waited for) -> false: the step has to be done again
bool wide_record(icp_step_ticket& t, int b, double* log_value_prop, double* fwd, double* bwd, int* status) {
  BatchItem& it = t.items[b];
  WideItem& w = it.W;
  icp_evaluator* e = it.e;
  icp_ctx& c = *e->ctx;
  const int r = c.r, n_props = t.n_props, generator = it.generator;
  const size_t P = 10 + (size_t)r;
  const double* theta_cur = t.theta_cur[b];
  double* theta_prop = t.theta_prop[b];
  PosteriorEntry** ec = w.ec;
  PosteriorEntry** ep = w.ep;
  if (w.eigen_first_use) {  // this step drew from a basis whose status nobody has looked at yet (the decomposition left it in pinned memory)
    icp_proposal* p = it.props[generator];
    PosteriorEntry& g = *ec[generator];
    int st = p->h_eig[g.status_off / 3];
    {  // (test-hooks build only: the n-th such look pretends the decomposition reported 2 — tests/test_gpu_wide.py)
      static const int pretend_at = dev_env("ICP_TEST_WIDE_EIGEN_STATUS") ? std::atoi(dev_env("ICP_TEST_WIDE_EIGEN_STATUS")) : 0;
      static std::atomic<int> looks{0};
      if (pretend_at > 0 && st == 0 && ++looks == pretend_at) st = 2;
    }
    if (st != 0) {
      // the multisection could not separate the spectrum (or the iteration did not converge): the per-stage decomposition, which
      // has the Jacobi fall-back in its launch sequence, takes over, and the step is done again from the basis it leaves
      g.eig_valid = false; g.eig_checked = false; g.eig_event_valid = false;
      p->warm_valid = false;
      p->ensure_eigen(g);
      sync_eigen(c);
      st = p->h_eig[g.status_off / 3];
      if (st != 0) {
        p->h_status[g.status_off + 2] = st;
        p->check_status(g);  // throws
      }
      return false;
    }
    p->h_status[g.status_off + 2] = 0;
  }
  const double* h_coeffs = c.h_res + 16;
  if (generator >= 0) {
    std::memcpy(theta_prop, theta_cur, sizeof(double) * 10);  // NonRigidIcpProposal.scala:64-66: only the shape changes
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(h_coeffs[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_prop[10 + j] = h_coeffs[j];
    }
  }
  StateSlot& s = *w.s;
  s.theta.assign(theta_prop, theta_prop + P);
  s.valid = true;
  s.stamp = ++c.clock;
  s.defo_valid = true;
  s.spheres_valid = w.spheres;
  s.n_surf = w.Ksurf;
  s.n_nnv = w.Knnv;
  s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0;
  if (w.do_post)
    for (int i = 0; i < n_props; ++i) {
      icp_proposal* p = it.props[i];
      ep[i]->theta.assign(theta_prop, theta_prop + P);
      ep[i]->valid = true;
      ep[i]->stamp = ++p->clock;
      p->h_status[ep[i]->status_off] = c.h_status[8 + i];
      p->h_status[ep[i]->status_off + 1] = 0;
      p->h_status[ep[i]->status_off + 2] = 0;
      if (w.shape_only) p->check_status(*ec[i]);
      p->check_status(*ep[i]);
    }
  for (int tl = 0; tl < w.n_tails; ++tl)
    if (c.h_status[tl] != 0) {  // rare: the fixed-point tail did not contract -> direct kernel
      std::vector<double> saved(c.h_res, c.h_res + 16);
      icp_proposal* p = it.props[tl / 2];
      TransitionTailIO io = w.tails[tl];
      io.out = c.d_res.p;
      io.status = c.d_status.p + 32;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 64);
      if (c.h_status[32] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
      saved[8 + tl] = c.h_res[0];
      std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
    }
  icp_evaluator::Memo* m = eval_store(e, theta_prop);
  m->status = finish_eval(e, c.h_res, &m->value, m->aux);
  *log_value_prop = m->value;
  *status = m->status;
  for (int i = 0; i < n_props; ++i) {
    if (!w.shape_only) { fwd[i] = -INFINITY; bwd[i] = -INFINITY; continue; }
    fwd[i] = c.h_res[8 + 2 * i];
    bwd[i] = c.h_res[9 + 2 * i];
    if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
  }
  e->last_prop.assign(theta_prop, theta_prop + P);
  ++c.paths.n[1]; ++g_step_paths.n[1];
  return true;
}

}  // namespace
