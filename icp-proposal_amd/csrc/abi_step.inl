// abi_step.inl — part of icp_abi.hip (one translation unit; included there, in order).
// the merged step (five launches): fronts, speculative decompositions, icp_chain_step / _prelaunch
namespace {

// posteriors the regression launch being prepared will carry over all its chains (set by the batched issuers around their per-chain
// preparation; 1 for a lone chain): decides regression_fold
thread_local int tl_regression_posteriors = 1;

// give back what a front holds without recording anything (its launches, if any, are harmless: they wrote to a state
// slot and memo entries that nobody refers to, to the search scratch and to the hints, which may be stale by design)
void release_front(StepFront& F);
// A half step launched ahead for an outcome that did not happen.  Its launches may still be running — they write the search
// scratch, the hints, a state slot and memo entries that the replacement is about to be given — so the step that replaces it
// must be ordered behind it: it takes the dropped front's own stream (enqueue_front toggles the parity back), where stream
// order does that, instead of the other one, where nothing would.
void drop_front(icp_evaluator* e) {
  if (!e->front.valid) return;
  const int parity = e->front.parity;
  release_front(e->front);
  e->front_parity = parity ^ 1;
}
void release_front(StepFront& F) {
  if (F.s) F.s->reserved = false;
  for (int i = 0; i < F.n_props; ++i)
    if (F.ep[i]) F.ep[i]->reserved = false;
  if (F.valid && F.eigen_first_use && F.generator >= 0 && F.ec[F.generator]) F.ec[F.generator]->eig_checked = false;
  F = StepFront{};
}

// KL bases of a state's posteriors `ec` that are not on record yet: all of them are started now, in ONE launch on the eigen
// stream (they run side by side); a step waits for the one it draws from only (through its completion word) — the other
// is ready when a later step draws from it.  m_in_flight: `stream` may still be writing what they read.
struct EigenCollect {  // the decompositions of a batch of chains, launched together (icp_chain_step_batched)
  hipStream_t stream;                  // the eigen stream of the batch's first context
  std::vector<EigenRequest> rq;
  std::vector<PosteriorEntry*> all;    // every entry with a request: ONE event, recorded behind the launch, stands for them all
};

// Events for the decompositions of whole batches: one ring PER DEVICE (an event belongs to the device that was current when it
// was created) that lives as long as the process — the entries of many chains, many proposals, many contexts, destroyed in any
// order, point at its slots.  A slot is handed out again after 64 batches; the entries of the earlier batch notice by the slot's
// generation counter (PosteriorEntry::eigen_event_stale) and wait on the host instead of on an event that now stands for other work.
// One record per batch instead of one per chain that moved: ≈ 2.3 µs of host time each.
struct BatchEventSlot { hipEvent_t ev = nullptr; uint64_t gen = 0; };
BatchEventSlot& next_batch_event(int device) {
  constexpr int kMaxDevices = 64, kRing = 64;
  static std::mutex mu;
  static BatchEventSlot ring[kMaxDevices][kRing];
  static unsigned turn[kMaxDevices] = {};
  require(device >= 0 && device < kMaxDevices, "device ordinal out of range");
  std::lock_guard<std::mutex> lk(mu);
  BatchEventSlot& e = ring[device][turn[device]++ % kRing];
  if (!e.ev) HIP_OK(hipEventCreateWithFlags(&e.ev, hipEventDisableTiming));  // (the caller has bound `device`)
  ++e.gen;
  return e;
}
// A launch that waits ON THE DEVICE for a word another stream's launch raises is safe as long as the waiting workgroups cannot keep
// the launch they wait for from becoming resident.  One chain's step cannot (its first launch is 14 workgroups, a decomposition
// six), two chains' neither; the batched step orders its chip-wide first launch behind the residency of its decompositions
// explicitly (StepBatchGate).  MANY contexts stepped one by one from many threads could, together, fill the chip with spinning first
// launches (round 2 saw the batched form of this: a 50 ms time-out): from three live contexts on, the single-chain
// step therefore takes its cross-stream order from events and stream order — no device-side wait at all.
bool device_side_waits_allowed() { return g_live_contexts.load(std::memory_order_relaxed) <= 2; }

void start_decompositions(icp_ctx& c, int n_props, icp_proposal* const* props, PosteriorEntry* const* ec, bool m_in_flight,
                          EigenCollect* collect = nullptr) {
  const int r = c.r;
  EigenRequest rqs[2];
  PosteriorEntry* need[2];
  int nn = 0;
  for (int i = 0; i < n_props; ++i)
    if (!ec[i]->eig_valid) { props[i]->prepare_eigen(*ec[i], &rqs[nn]); need[nn++] = ec[i]; }
  if (nn == 0) return;
  need[0]->eig_done_shared = nullptr;
  for (int i = 1; i < nn; ++i) need[i]->eig_done_shared = need[0]->eig_done;
  for (int i = 0; i < nn; ++i) need[i]->eig_shared_gen = nullptr;
  if (collect) {
    if (m_in_flight) HIP_OK(hipStreamSynchronize(c.stream));
    for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = true;  // (recorded by the caller behind the batch's launch)
    (void)eigen_stream_for(c, collect->stream);
    for (int i = 0; i < nn; ++i) collect->rq.push_back(rqs[i]);
    for (int i = 0; i < nn; ++i) collect->all.push_back(need[i]);
    return;
  }
  const hipStream_t es = eigen_stream_for(c, c.eig_stream.get());
  // M of these entries is complete when they were recorded by a finished chain step (the host has seen its results);
  // only work another entry point has put on `stream` may still be writing them
  if (m_in_flight) {
    HIP_OK(hipEventRecord(c.ev_ready, c.stream));
    HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
  }
  if (launch_posterior_eigen_pair(es, r, c.sqrt_lambda.p, nn, rqs)) {
    if (!device_side_waits_allowed()) {
      // many contexts in the process: an event instead of the completion word (see device_side_waits_allowed)
      HIP_OK(hipEventRecord(need[0]->eig_done, es));
      for (int i = 0; i < nn; ++i) { need[i]->eig_event_valid = true; need[i]->done_value = 0; }
      return;
    }
    // completion words: the step's first launch waits for the one it draws from on the device; no event (host time on the
    // accepted path) — whoever else needs the basis waits for the eigen stream on the host (await_eigen)
    for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = false;
    return;
  }
  for (int i = 0; i < nn; ++i) {  // (ranks > 64: one after the other)
    need[i]->done_value = 0;
    launch_posterior_eigen(es, r, rqs[i].M, c.sqrt_lambda.p, rqs[i].Vwarm, rqs[i].V, rqs[i].Vt, rqs[i].S, rqs[i].work, rqs[i].status,
                           nullptr, rqs[i].host_status);
  }
  HIP_OK(hipEventRecord(need[0]->eig_done, es));  // (one event for two launches in a row)
  for (int i = 0; i < nn; ++i) need[i]->eig_event_valid = true;
}

void front_launches(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* key, StepFront& F, bool batched,
                    bool two_streams);

// launches 1-3 of the step (theta_cur --generator/key--> proposal); `key` = z or the proposed state (see StepFront)
// (batched: the launches are being captured for icp_chain_step_batched — one stream, nothing to wait for on the device
// but the decomposition)
void enqueue_front(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                   const double* key, StepFront& F, bool batched = false) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  F = StepFront{};
  F.n_props = n_props; F.generator = generator;
  for (int i = 0; i < n_props; ++i) F.props[i] = props[i];
  F.theta_cur.assign(theta_cur, theta_cur + 10 + r);
  F.key.assign(key, key + (generator >= 0 ? r : 10 + r));
  F.parity = (e->front_parity ^= 1);
  // ---- cached side: posterior of the current state for every proposal (+ its KL basis for the generating one)
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  bool missing = false;
  for (int i = 0; i < n_props; ++i) missing = missing || !props[i]->find_entry(theta_cur);
  if (missing && c.front_stream_used) {  // the posteriors are computed on `stream` with the scratch a step in flight may still use
    c.front_stream.sync();
    c.front_stream_used = false;
  }
  for (int i = 0; i < n_props; ++i) ec[i] = &props[i]->posterior(theta_cur, false);  // NonRigidIcpProposal.scala:54,76
  if (missing) c.stream_used_elsewhere = true;  // … and this step reads them
  const bool m_in_flight = c.stream_used_elsewhere;  // `stream` may still be writing what the decompositions below read
  const bool two_streams = !c.pipeline_off && !batched && device_side_waits_allowed();
  F.stream = (F.parity && two_streams) ? c.front_stream.get() : c.stream;
  if (F.stream == c.front_stream.peek()) {
    if (c.stream_used_elsewhere) {  // another entry point has work on `stream` that this step may depend on: join once
      HIP_OK(hipEventRecord(c.ev_join, c.stream));
      HIP_OK(hipStreamWaitEvent(c.front_stream.get(), c.ev_join, 0));
    }
    c.front_stream_used = true;
  }
  c.stream_used_elsewhere = false;
  start_decompositions(c, n_props, props, ec, m_in_flight);
  bool eigen_first_use = false;
  if (generator >= 0) {
    // (await_eigen on the front's stream: through the decomposition's own completion word when it has one — see launch 1)
    if (ec[generator]->done_value == 0 && ec[generator]->eigen_event()) HIP_OK(hipStreamWaitEvent(F.stream, ec[generator]->eigen_event(), 0));
    eigen_first_use = !ec[generator]->eig_checked;  // (possibly of an earlier prefetch or speculation): fetch its status
    ec[generator]->eig_checked = true;
  }
  F.eigen_first_use = eigen_first_use;

  // ---- new side: one state slot, one memo entry per proposal
  StateSlot& s = c.fresh_state();
  s.reserved = true;
  F.s = &s;
  s.pose = c.pose_of(generator >= 0 ? theta_cur : key);
  for (int i = 0; i < n_props; ++i) { ep[i] = &props[i]->fresh_entry(); ep[i]->reserved = true; }
  front_launches(e, n_props, props, generator, key, F, batched, two_streams);
  F.valid = true;
}

// launches 1-4 of a merged step with every choice made: F.ec / F.ep (the posterior entries of the current and of the proposed state),
// F.s (the proposed state's slot, its pose set), F.parity, F.stream.  enqueue_front above makes those choices from the memo; the
// on-device chain loop (icp_chains_run_on_device) fixes them once per chain and captures the arguments.
void front_launches(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* key, StepFront& F, bool batched,
                    bool two_streams) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  StateSlot& s = *F.s;
  const icp_evaluator_params& evp = e->prm;
  icp_proposal* pm = nullptr;  // ModelSampling proposal
  icp_proposal* pt = nullptr;  // TargetSampling proposal
  int im = -1, it = -1;
  for (int i = 0; i < n_props; ++i) {
    if (props[i]->prm.direction == ICP_MODEL_SAMPLING) { pm = props[i]; im = i; }
    else { pt = props[i]; it = i; }
  }
  const bool ev_m2t = evp.mode != ICP_TARGET_TO_MODEL, ev_t2m = evp.mode != ICP_MODEL_TO_TARGET;
  const int Ksurf = std::max(ev_m2t ? evp.n_model_ids : 0, pm ? pm->K : 0);
  F.Ksurf = Ksurf;
  require(Ksurf <= c.N, "model id count exceeds the number of model points");
  QueryBuffers qs = c.query_scratch(Ksurf, c.target.T);
  QueryBuffers qv{};
  if (pt) qv = c.query_scratch(pt->K, c.N, 1);

  SurfaceTask st_surf = make_surface_task(c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, Ksurf, s.x.p,
                                          c.hint_surf.p, qs, s.surf_cp.p, s.surf_d2.p, s.surf_tri.p);
  VertexTask st_vert{};
  if (pt) st_vert = make_vertex_task(c.N, s.x.p, pt->K, pt->target_pts.p, pt->hint_nn.p, qv, nullptr, pt->nn_id.p);
  st_vert.thr2 = nullptr;  // bounds are computed by the filter launch itself (see vertex_filter)
  // the evaluator's reverse direction (IndependentPointDistanceEvaluator.scala:49-54, Collective…Evaluator.scala:55-64): its
  // decimated-target points against the surface of the NEW instance — spheres and bounds are taken by the filter launch itself
  SurfaceTask st_t2m{};
  if (ev_t2m) {
    QueryBuffers qt = c.query_scratch(e->Kt, c.T, 2);
    st_t2m = make_surface_task(c.T, s.x.p, c.tris.p, nullptr, e->Kt, e->d_tpts, e->hint_tri.p, qt, e->t2m_cp.p, e->t2m_d2.p, e->t2m_tri.p);
    st_t2m.order = c.tri_order.p;
    st_t2m.thrA = nullptr;
  }

  // 1: coefficients -> instance -> bounds
  StepBeginArgs b{};
  b.N = c.N; b.r = r; b.inst_blocks = (c.N + 255) / 256;
  b.Qp = c.Qp.p; b.ref = c.ref.p; b.mean = c.mean.p; b.pose = s.pose;
  b.propose = generator >= 0 ? 1 : 0;
  const double* src = generator >= 0 ? key : key + 10;
  // (the merged launches cover ranks whose factor fits LDS — step_finish_supported, about 116 — so the r host-drawn numbers
  // always travel inside the kernel arguments)
  require(r <= kStepInlineZ, "internal: merged step at a rank above the inline-argument limit");
  std::memcpy(b.zin, src, sizeof(double) * r);
  if (generator >= 0) {
    PosteriorEntry& g = *ec[generator];
    b.prop = ProposeIn{g.alpha.p, g.V.p, g.S.p, c.inv_sqrt_lambda.p, c.P.p, g.coeffs.p, nullptr, kSigma2,
                       props[generator]->prm.step_length, props[generator]->sampler == ICP_SAMPLER_CHOLESKY_ROOT};
    int t = 0;
    while (t < 6 && (r << (t + 1)) <= 256 && (r >> (t + 1)) >= 8) ++t;  // = matvec_tpr_log2(r, 256) of k_propose
    b.tpr_log2 = t;
  }
  b.n_out = 0;
  b.out[b.n_out++] = s.coeffs.p;
  for (int i = 0; i < n_props; ++i) b.out[b.n_out++] = ep[i]->coeffs.p;
  b.out[b.n_out++] = c.h_res + 16 + F.parity * kCoeffArea;
  b.x = s.x.p;
  b.has_surf = 1; b.surf = st_surf;
  b.has_vert = pt ? 1 : 0; b.vert = st_vert;
  b.zero2 = ev_t2m ? st_t2m.cnt : nullptr; b.n_zero2 = ev_t2m ? st_t2m.Kpad : 0;
  // (nothing to wait for before the first finish launch, nor when every step is on one stream)
  b.wait_flag = (c.last_back_seq > 0 && two_streams) ? c.d_done.p + 2 : nullptr;
  // test hook: the first launch waits for a word that never comes, times out, and the step is repeated unpipelined
  static const int starve_pipeline = dev_env("ICP_TEST_STARVE_PIPELINE") ? (1 << 24) : 0;
  b.wait_seq = c.last_back_seq + starve_pipeline;
  b.wait_error = c.h_wait_error;
  b.wait_ticks = c.profiling ? c.d_wait_ticks.p : nullptr;
  if (generator >= 0 && ec[generator]->done_value != 0) {
    b.wait2_flag = props[generator]->eig_words.p + ec[generator]->status_off / 3;
    b.wait2_seq = ec[generator]->done_value;
    // still in flight (the register-holding variant of launch 1 multiplies with the KL basis: not for the Cholesky-root sampler)
    b.hold_regs = *(volatile int*)(props[generator]->h_eig + ec[generator]->status_off / 3) == -1 && !b.prop.root;
  }
  launch_step_begin(F.stream, b);

  // 2 + 3: searches and correspondences
  StepSearchArgs q{};
  q.n_surf = ev_t2m ? 2 : 1; q.n_vert = pt ? 1 : 0;
  q.s[0] = st_surf;
  q.fstart[0] = 0; q.fstart[1] = filter_grid_blocks(st_surf.tblocks, st_surf.ksplit);
  q.rstart[0] = 0; q.rstart[1] = Ksurf;
  int nt = 1;  // tasks so far (surface tasks first)
  if (ev_t2m) {
    q.s[1] = st_t2m;
    q.fstart[2] = q.fstart[1] + filter_grid_blocks(st_t2m.tblocks, st_t2m.ksplit);
    q.rstart[2] = q.rstart[1] + e->Kt;
    nt = 2;
  }
  q.s_corr[0] = q.s_corr[1] = q.v_corr[0] = q.v_corr[1] = -1;
  int n_corr = 0;
  if (pm) {
    q.corr[n_corr] = CorrTask{pm->K, ep[im]->corr(), s.x.p, nullptr, c.target.boundary.p, nullptr, pm->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p};
    q.s_corr[0] = n_corr++;
  }
  if (pt) {
    q.v[0] = st_vert;
    q.fstart[nt + 1] = q.fstart[nt] + filter_grid_blocks(st_vert.vblocks, st_vert.ksplit);
    q.rstart[nt + 1] = q.rstart[nt] + pt->K;
    q.corr[n_corr] = CorrTask{pt->K, ep[it]->corr(), s.x.p, pt->target_pts.p, c.boundary.p, nullptr, pt->prm.boundary_aware,
                              s.pose, c.ref.p, c.mean.p, c.tris.p, c.adj_off.p, c.adj.p};
    q.v_corr[0] = n_corr++;
  }
  launch_step_filter(F.stream, q);
  launch_step_resolve(F.stream, q);

  // 4: regressions + likelihood reduction
  StepRegressionArgs g{};
  g.n = n_props; g.r = r;
  g.ntiles = regression_tiles(r);
  g.Q = c.Q.p;
  g.ustart[0] = 0;
  int* splits = F.splits;
  for (int i = 0; i < n_props; ++i) {
    icp_proposal* p = props[i];
    const int leaves = regression_splits(p->K);
    g.K[i] = p->K;
    g.kchunk[i] = std::max(1, (p->K + leaves - 1) / leaves);
    // (chains stepped side by side fold a posterior's leaves into one partial: tl_regression_posteriors = posteriors in the launch)
    g.fold[i] = regression_fold(p->K, r, tl_regression_posteriors);
    g.macro[i] = regression_macro(r, g.fold[i]);
    splits[i] = leaves / g.fold[i];  // (fold is 1 or `leaves`)
    g.cb[i] = ep[i]->corr();
    g.wt[i] = 1.0 / (p->prm.tangential_noise * p->prm.tangential_noise);
    g.kappa[i] = 1.0 / (p->prm.noise_along_normal * p->prm.noise_along_normal) - g.wt[i];
    p->mpart_half = (p->mpart_half + 1) % icp_proposal::kMpartRing;
    g.Mpart[i] = p->mpart_for_write(p->mpart_half, F.stream);
    g.status[i] = p->status.p + ep[i]->status_off;
    g.ustart[i + 1] = g.ustart[i] + regression_units(r, leaves, g.fold[i], g.macro[i]);
  }
  if (n_props == 1) g.ustart[2] = g.ustart[1];
  g.reduce_kind = evp.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE ? 1 : 2;
  g.Kred = ev_m2t ? evp.n_model_ids : 0; g.d2 = s.surf_d2.p; g.mean = evp.gauss_mean; g.sigma = evp.gauss_sigma;
  g.Kred2 = ev_t2m ? e->Kt : 0; g.d2b = e->t2m_d2.p;
  g.red_out = c.h_res + kReduceArea + F.parity * 8;  // (its own half: the host may still be reading the previous step's)
  for (int i = 0; i < n_props; ++i) { F.mpart[i] = g.Mpart[i]; F.mpart_half[i] = props[i]->mpart_half; }
  launch_step_regression(F.stream, g);
  if (!batched) {  // 4b: many partials are summed by many CUs before the one-workgroup factorisation (see launch_step_reduce)
    bool many = false;
    for (int i = 0; i < n_props; ++i) many = many || splits[i] >= kStepReduceSplits;
    if (many) {
      StepReduceArgs ra{};
      ra.n = n_props; ra.nn = (r + 1) * (r + 1);
      for (int i = 0; i < n_props; ++i) { ra.Mpart[i] = g.Mpart[i]; ra.splits[i] = splits[i]; splits[i] = 1; }
      launch_step_reduce(F.stream, ra);
    }
  }
}

// Host side of a merged step whose results have arrived in the context's pinned memory: status of the decomposition it
// drew from, the proposed state, the memo entries, the rare direct transition tail, likelihood and densities.
// -> false: the step has to be done again (nothing of it has been recorded)
bool chain_step_record(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur, StepFront& F,
                       const StepFinishArgs& f, double* theta_prop, double* log_value_prop, double* fwd, double* bwd, int* status) {
  icp_ctx& c = *e->ctx;
  const int r = c.r;
  PosteriorEntry** ec = F.ec;
  PosteriorEntry** ep = F.ep;
  StateSlot& s = *F.s;
  const int Ksurf = F.Ksurf;
  const bool eigen_first_use = F.eigen_first_use;
  const bool eigen_status_pinned = eigen_speculation_supported(r);
  const double* h_coeffs = c.h_res + 16 + F.parity * kCoeffArea;
  if (eigen_first_use && eigen_status_pinned) {  // this step's first launch waited for that decomposition: its status is in
    icp_proposal* p = props[generator];
    const int st = p->h_eig[ec[generator]->status_off / 3];
    if (st == kEigenGaveUp) {  // a speculative decomposition that never saw its input (see k_posterior_eigen_rr): the
      // step just computed drew from a stale basis — drop it (nothing of it has been recorded) and do it again
      ++c.stats.speculation_giveups; ++g_runtime_stats.speculation_giveups;
      ec[generator]->eig_valid = false;
      ec[generator]->eig_checked = false;
      p->warm_valid = false;  // (it pointed at the basis that was never written)
      return false;
    }
    p->h_status[ec[generator]->status_off + 2] = st;
  }
  const size_t P = 10 + (size_t)r;
  if (generator >= 0) {
    std::memcpy(theta_prop, theta_cur, sizeof(double) * 10);  // :64-66 only the shape changes
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(h_coeffs[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_prop[10 + j] = h_coeffs[j];
    }
  }
  s.theta.assign(theta_prop, theta_prop + P);
  s.valid = true;
  s.stamp = ++c.clock;
  s.n_surf = Ksurf;
  for (int i = 0; i < n_props; ++i) {
    icp_proposal* p = props[i];
    ep[i]->theta.assign(theta_prop, theta_prop + P);
    ep[i]->valid = true;
    ep[i]->stamp = ++p->clock;
    p->h_status[ep[i]->status_off] = c.h_status[8 + i];
    p->h_status[ep[i]->status_off + 1] = 0;
    p->h_status[ep[i]->status_off + 2] = 0;
    p->check_status(*ec[i]);
    p->check_status(*ep[i]);
  }
  for (int t = 0; t < 2 * n_props; ++t)
    if (c.h_status[t] != 0) {  // rare: the fixed-point tail did not contract -> direct kernel
      std::vector<double> saved(c.h_res, c.h_res + 16);
      icp_proposal* p = props[t / 2];
      TransitionTailIO io = (t % 2 == 0) ? f.fwd[t / 2] : f.bwd[t / 2];
      io.out = c.d_res.p;
      io.status = c.d_status.p + 32;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 64);
      if (c.h_status[32] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
      saved[8 + t] = c.h_res[0];
      std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
    }
  for (int i = 0; i < 8; ++i) c.h_res[i] = c.h_res[kReduceArea + F.parity * 8 + i];  // launch 4's reductions, where finish_eval looks
  icp_evaluator::Memo* m = eval_store(e, theta_prop);
  m->status = finish_eval(e, c.h_res, &m->value, m->aux);
  *log_value_prop = m->value;
  *status = m->status;
  for (int i = 0; i < n_props; ++i) {
    fwd[i] = c.h_res[8 + 2 * i];
    bwd[i] = c.h_res[9 + 2 * i];
    if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
  }
  return true;
}

// ICP_SPECULATION: 0 never, 1 always, unset = adaptive (2): while the chain's running acceptance rate is high

bool front_matches(const StepFront& F, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                   const double* key, int r) {
  if (!F.valid || F.n_props != n_props || F.generator != generator) return false;
  for (int i = 0; i < n_props; ++i)
    if (F.props[i] != props[i]) return false;
  return std::memcmp(F.theta_cur.data(), theta_cur, sizeof(double) * (10 + r)) == 0 &&
         std::memcmp(F.key.data(), key, sizeof(double) * F.key.size()) == 0;
}

bool wide_chain_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in);  // (the wide step: further down)

// shared argument checks of icp_chain_step and icp_chain_step_prelaunch; -> the merged launches cover this call
bool chain_step_covered(icp_evaluator* e, int n_props, icp_proposal* const* props, int generator, const double* theta_cur,
                        const double* theta_prop_in) {
  icp_ctx& c = *e->ctx;
  bool per_stage = !step_pipeline_covers(e, n_props, props);
  if (!per_stage && generator < 0) {
    // a state the caches already know (or a pose move, whose ICP transition densities are -inf) has nothing to merge
    per_stage = !pose_equal(theta_cur, theta_prop_in) || c.find_state(theta_prop_in) || eval_lookup(e, theta_prop_in);
    for (int i = 0; i < n_props && !per_stage; ++i) per_stage = props[i]->find_entry(theta_prop_in) != nullptr;
  }
  return !per_stage;
}

}  // namespace

extern "C" {

int icp_chain_step_prelaunch(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, int32_t generator, const double* theta_cur,
                             const double* z_or_theta_prop) {
  return guard([&] {
    require(e != nullptr, "null argument");
    if (n_props == 0) {  // "nothing further": drop a pending half step
      std::lock_guard<std::recursive_mutex> lk0(e->ctx->mu);
      drop_front(e);
      return;
    }
    require(theta_cur && z_or_theta_prop, "null argument");
    require(n_props >= 1 && n_props <= 2 && props, "bad proposal list");
    require(generator < n_props, "generator index out of range");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    if (generator < 0) check_theta_finite(&c, z_or_theta_prop);
    else
      for (int j = 0; j < c.r; ++j)
        if (!std::isfinite(z_or_theta_prop[j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    drop_front(e);
    if (c.pipeline_off) return;
    // with several chains in the process the device is not idle during one chain's turn-around, and the launches of a
    // dropped half step cost the others host time (tools/multichain.py)
    if (g_live_contexts.load(std::memory_order_relaxed) > 1) return;
    if (!chain_step_covered(e, n_props, props, generator, theta_cur, z_or_theta_prop)) return;  // nothing to pre-launch
    // every posterior of the assumed current state must be on record already (the step in flight computed them)
    for (int i = 0; i < n_props; ++i)
      if (!props[i]->find_entry(theta_cur)) return;
    Bound _b(&c, true);
    try {
      enqueue_front(e, n_props, props, generator, theta_cur, z_or_theta_prop, e->front);
    } catch (...) {
      release_front(e->front);
      throw;
    }
  });
}

int icp_chain_step(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, int32_t generator, const double* theta_cur,
                   const double* z, double* theta_prop, double* log_value_prop, double* fwd, double* bwd) {
  int status = ICP_OK;
  bool per_stage = false, redo = false, wide = false;
  static thread_local int wide_depth = 0;  // (a wide step that has to be repeated comes back through this entry point: bounded)
  int rc = guard([&] {
    require(e && theta_cur && theta_prop && log_value_prop, "null argument");
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    require(generator < n_props, "generator index out of range");
    require(generator < 0 || z, "z is null");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    if (generator < 0) check_theta_finite(&c, theta_prop);
    else
      for (int j = 0; j < c.r; ++j)
        if (!std::isfinite(z[j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    const int r = c.r;
    const double* key = generator >= 0 ? z : theta_prop;
    const bool reuse = n_props <= 2 && front_matches(e->front, n_props, props, generator, theta_cur, key, r);
    if (e->front.valid && !reuse) drop_front(e);  // pre-launched for another outcome: dropped
    per_stage = !reuse && !chain_step_covered(e, n_props, props, generator, theta_cur, theta_prop);
    if (per_stage) {
      // what the five merged launches do not cover as a configuration takes the wide step (a batch of one chain)
      wide = wide_depth < 2 && n_props >= 1 && n_props <= 2 && !step_pipeline_covers(e, n_props, props) &&
             wide_chain_covered(e, n_props, props, generator, theta_cur, theta_prop);
      return;
    }
    Bound _b(&c, true);
    g_host_timing.start();

    for (int i = 0; i < n_props; ++i) props[i]->resolve_speculation(theta_cur);
    if (!e->last_prop.empty()) {  // did the caller keep the state the previous step proposed?
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    StepFront F;
    struct FrontGuard {  // whatever happens below, the slots of this step are not left reserved
      StepFront* f;
      ~FrontGuard() { if (f) release_front(*f); }
    } front_guard{&F};
    if (reuse) { F = e->front; e->front = StepFront{}; }
    else enqueue_front(e, n_props, props, generator, theta_cur, key, F);
    g_host_timing.mark(0);
    PosteriorEntry** ec = F.ec;
    PosteriorEntry** ep = F.ep;
    StateSlot& s = *F.s;
    const bool eigen_first_use = F.eigen_first_use;
    // the decompositions of ranks <= 64 leave their status in pinned memory themselves; the others need a copy
    const bool eigen_status_pinned = eigen_speculation_supported(r);
    const bool eigen_enqueued = eigen_first_use && !eigen_status_pinned;
    for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
    for (int i = 0; i < 16; ++i) c.h_status[i] = 0;

    // Adaptive (speculation_mode): an accepted step finds its basis ≈ 50 µs earlier; a rejected one has paid one launch
    // (≈ 3 µs of host time, a few CUs for at most one sweep) for nothing — worth it unless next to nothing is accepted
    // (measured: 8.3k against 7.4k it/s over a chain's first 20 steps, 13.6k against 13.7k at one acceptance in three).
    const int spec_mode = speculation_mode();
    const bool speculate = (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1)) && !c.speculation_off &&
                           g_live_contexts.load(std::memory_order_relaxed) <= 2 && n_props > 0 && eigen_speculation_supported(r);
    // test hook: the speculative decompositions wait for a word that never comes, time out and are repeated
    static const int starve = dev_env("ICP_TEST_STARVE_SPECULATION") ? (1 << 24) : 0;
    const int step_seq = ++c.step_seq;

    // 5: factorisations + tails (results go straight to pinned host memory)
    StepFinishArgs f{};
    f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
    for (int i = 0; i < n_props; ++i) {
      icp_proposal* p = props[i];
      f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
      f.M[i] = ep[i]->M.p; f.alpha[i] = ep[i]->alpha.p;
      f.status[i] = p->status.p + ep[i]->status_off;
      f.host_status[i] = c.h_status + 8 + i;
      f.fwd[i] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, ec[i]->coeffs.p, ep[i]->coeffs.p, p->prm.step_length,
                                  c.h_res + 8 + 2 * i, c.h_status + 2 * i};
      f.bwd[i] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, ep[i]->coeffs.p, ec[i]->coeffs.p, p->prm.step_length,
                                  c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
    }
    f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = step_seq;
    f.ready_flag = c.d_done.p + 2;  // (speculative decompositions and the next step's first launches wait for it)
    c.last_back_seq = step_seq;
    launch_step_finish(F.stream, f);
    g_host_timing.mark(1);
    // the caller's outcome-independent host work runs beside the device — first of all the pre-launch of the next step's
    // first half, which the device can start as soon as the finish launch above has
    // KL bases of the proposed state's posteriors, in case it is accepted: they run on the eigen stream beside the
    // factorisations and the host's round trip; the next call keeps or cancels them (resolve_speculation).  One launch (both
    // directions side by side), issued BEFORE the caller's hook: the accepted path waits for nothing else.
    if (speculate) {
      EigenSpec specs[2];
      EigenRequest rqs[2];
      for (int i = 0; i < n_props; ++i) props[i]->speculate_eigen(*ep[i], *ec[i], F.splits[i], F.mpart_half[i], c.d_done.p + 2, step_seq + starve, &specs[i], &rqs[i]);
      // (developer switch: the tridiagonal route for these decompositions while the running acceptance rate is above a threshold —
      // its time does not depend on how far the chain has moved, the warm-started iteration's does (124 µs on average over a chain's
      // first steps against 84 in the steady state); measured: 7.2k against 8.7k it/s over the first 20 steps — from input to
      // completion word the two launches take ≈ 110 µs, the iteration with its replay beside it ≈ 100 even at four sweeps.)
      static const double direct_above = dev_env("ICP_DIRECT_ABOVE") ? std::atof(dev_env("ICP_DIRECT_ABOVE")) : 2.0;
      for (int i = 0; i < n_props; ++i) rqs[i].direct = e->acc_ema >= direct_above;
      const hipStream_t es = eigen_stream_for(c, c.eig_stream.get());
      launch_posterior_eigen_pair(es, r, c.sqrt_lambda.p, n_props, rqs);  // (no event: completion words, see start_decompositions)
    }
    // the caller's outcome-independent host work runs beside the device — first of all the pre-launch of the next step's
    // first half (under the rejection assumption), which the device can start as soon as the finish launch above has
    if (c.idle_fn) c.idle_fn(c.idle_arg);
    g_host_timing.mark(2);
    if (eigen_enqueued) {
      if (F.stream != c.stream) HIP_OK(hipStreamSynchronize(F.stream));
      sync_proposal_status_if(props[generator], true);
      c.finish(0, 0);
    } else {
      // results and flag are written into pinned memory by the kernels: poll the flag (≈ 4 µs less than a stream
      // synchronisation); give up after 2 s and let the synchronisation report what went wrong
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != f.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) break;
      }
      if (*flag != f.seq) {
        if (F.stream != c.stream) HIP_OK(hipStreamSynchronize(F.stream));
        c.finish(0, 0);
      }
      c.stage_used = 0;
    }

    // ---- bookkeeping with the results in hand
    g_host_timing.mark_wait(eigen_first_use);
    if (c.h_wait_error[0]) {
      // A first launch did not see the word it waits for within 50 ms and went ahead unordered.  That is what a tool does
      // that lets one kernel run at a time in an order of its own (rocprofv3 --pmc): drain everything, switch the pipelining
      // off for this context, and do the step again — nothing of it has been recorded.
      HIP_OK(hipStreamSynchronize(c.stream));
      c.front_stream.sync();  // (a half step launched ahead may time out here, too)
      sync_eigen(c);
      c.h_wait_error[0] = 0;
      ++c.stats.wait_timeouts; ++g_runtime_stats.wait_timeouts;
      if (c.pipeline_off) fail(ICP_ERR_DEVICE, "internal: a step's first launch timed out on its word");
      c.pipeline_off = true;
      ++c.stats.pipeline_fallbacks; ++g_runtime_stats.pipeline_fallbacks;
      ++c.stats.step_redos; ++g_runtime_stats.step_redos;
      if (e->front.valid) release_front(e->front);
      redo = true;
      return;
    }
    if (!chain_step_record(e, n_props, props, generator, theta_cur, F, f, theta_prop, log_value_prop, fwd, bwd, &status)) {
      ++c.stats.step_redos; ++g_runtime_stats.step_redos;
      redo = true;
      return;
    }
    s.reserved = false;
    for (int i = 0; i < n_props; ++i) ep[i]->reserved = false;
    front_guard.f = nullptr;
    ++c.paths.n[0]; ++g_step_paths.n[0];
    e->last_prop.assign(theta_prop, theta_prop + 10 + r);
    g_host_timing.mark(4);
    g_host_timing.end();
  });
  if (rc != ICP_OK) return rc;
  if (redo) return icp_chain_step(e, n_props, props, generator, theta_cur, z, theta_prop, log_value_prop, fwd, bwd);
  if (per_stage && wide) {
    int32_t gen = generator, st = ICP_OK;
    const double* tc = theta_cur;
    const double* zz = z;
    double* tp = theta_prop;
    ++wide_depth;
    rc = icp_chain_step_batched(1, &e, n_props, props, &gen, &tc, generator >= 0 ? &zz : nullptr, &tp, log_value_prop, fwd, bwd, &st);
    --wide_depth;
    return rc != ICP_OK ? rc : st;
  }
  if (per_stage) {  // same results through the per-stage kernels
    ++e->ctx->paths.n[2]; ++g_step_paths.n[2];
    if (generator >= 0) {
      rc = icp_proposal_propose(props[generator], theta_cur, z, theta_prop, nullptr);
      if (rc != ICP_OK) return rc;
    }
    return icp_chain_eval_step(e, n_props, props, theta_cur, theta_prop, log_value_prop, fwd, bwd);
  }
  return status;
}

// B chains per launch.  Every chain takes the merged step of icp_chain_step with its own context's buffers; the five
// launches are recorded per chain (StepCapture) and issued ONCE for all of them on the first chain's stream, the
// decompositions of chains that moved run on their own contexts' eigen streams beside it (launch 1 waits for each on the
// device, as in the single-chain step).  Chains this does not cover (another device or rank than the first chain's, a
// context that already has a chain in the batch, a configuration the merged launches do not cover) take icp_chain_step
// one after the other, behind the batch.
// The work is split in two so that a caller can keep two batches in flight (the decompositions of one run beside the
// launches of the other): _issue ends when everything is on the device, _collect waits and records.
} // extern "C" (helpers)

// StepRandom::normal of the C++ harness (host/icp_host.hpp; = orc_rng_normal of the oracle): Box–Muller over the counter-based uniforms
static inline uint64_t harness_splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
static inline double harness_uniform(uint64_t seed, uint64_t step, uint64_t lane) {
  const uint64_t h = harness_splitmix64(harness_splitmix64(harness_splitmix64(seed) ^ (step * 0xD1342543DE82EF95ull)) ^ (lane * 0x2545F4914F6CDD1Dull));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
static inline double harness_normal(uint64_t seed, uint64_t step, uint64_t lane) {
  const double u1 = harness_uniform(seed, step, 2 * lane + 1000), u2 = harness_uniform(seed, step, 2 * lane + 1001);
  return std::sqrt(-2.0 * std::log(u1)) * std::cos(2.0 * M_PI * u2);
}
