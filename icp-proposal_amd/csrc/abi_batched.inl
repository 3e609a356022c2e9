// abi_batched.inl — part of icp_abi.hip (one translation unit; included there, in order).
// C ABI: icp_chain_step_batched_issue / _collect / _abandon
extern "C" {

int icp_chain_step_batched_issue(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props,
                                 const int32_t* generator, const double* const* theta_cur_in, const double* const* z_in,
                                 double* const* theta_prop_in, double* log_value_prop, double* fwd, double* bwd, int32_t* status,
                                 icp_ctx* launch_ctx, icp_step_ticket** out) {
  if (out) *out = nullptr;
  icp_step_ticket* tk = nullptr;
  int rc = guard([&] {
    require(out != nullptr, "null argument");
    require(n_chains >= 1 && evaluators && generator && theta_cur_in && theta_prop_in && log_value_prop && status, "null argument");
    tk = new icp_step_ticket();
    icp_step_ticket& t = *tk;
    t.n_chains = n_chains; t.n_props = n_props;
    t.theta_cur.assign(theta_cur_in, theta_cur_in + n_chains);
    t.theta_prop.assign(theta_prop_in, theta_prop_in + n_chains);
    t.z.assign(n_chains, nullptr);
    if (z_in) t.z.assign(z_in, z_in + n_chains);
    t.log_value_prop = log_value_prop; t.fwd = fwd; t.bwd = bwd; t.status = status;
    std::vector<BatchItem>& items = t.items;
    std::vector<StepCapture>& caps = t.caps;
    const double* const* theta_cur = t.theta_cur.data();
    const double* const* z = z_in ? t.z.data() : nullptr;
    double* const* theta_prop = t.theta_prop.data();
    typedef BatchItem Item;
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    t.props.assign(props, props + (size_t)n_chains * n_props);
    items.resize(n_chains);
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      it.e = evaluators[b];
      it.props = t.props.data() + (size_t)b * n_props;
      it.generator = generator[b];
      require(it.e && theta_cur[b] && theta_prop[b], "null argument");
      require(it.generator < n_props, "generator index out of range");
      require(it.generator < 0 || (z && z[b]), "z is null");
      icp_ctx& c = *it.e->ctx;
      for (int i = 0; i < n_props; ++i) require(it.props[i] && it.props[i]->ctx == &c, "proposal belongs to another context");
      check_theta_finite(&c, theta_cur[b]);
      if (it.generator < 0) check_theta_finite(&c, theta_prop[b]);
      else
        for (int j = 0; j < c.r; ++j)
          if (!std::isfinite(z[b][j])) fail(ICP_ERR_NOT_FINITE, "z contains a non-finite value");
      it.key = it.generator >= 0 ? z[b] : theta_prop[b];
      status[b] = ICP_OK;
    }
    // `lead` carries the launches (its stream, its argument buffers); the decompositions go to the first chain's eigen stream
    icp_ctx& elead = *items[0].e->ctx;
    icp_ctx& lead = launch_ctx ? *launch_ctx : elead;
    require(lead.device == elead.device, "launch context on another device");
    t.lead = &lead;
    // (the launch context's rings hold kBatchRing batches: argument slots, eigen records, gate words, events — a ticket more would
    // rewrite what a batch still on the device reads)
    if (lead.tickets_in_flight.fetch_add(1) >= icp_ctx::kBatchRing) {
      --lead.tickets_in_flight;
      fail(ICP_ERR_BUSY, "the launch context already carries ICP_MAX_BATCHES_IN_FLIGHT uncollected batches: collect or abandon one first");
    }
    t.counted = true;
    // ---- which chains share the launches
    int n_batched = 0, wide_first = -1;
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      icp_ctx& c = *it.e->ctx;
      bool ok = n_props >= 1 && n_props <= 2 && c.device == elead.device && c.r == elead.r;
      for (int a = 0; a < b && ok; ++a) ok = !(items[a].batched && items[a].e->ctx == &c);
      if (!ok) continue;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      if (c.batch_busy) fail(ICP_ERR_BUSY, "a chain's context already belongs to a batch in flight");
      if (it.e->front.valid || c.front_stream_used) {  // half steps launched ahead by the pipelined entry points: drained
        if (it.e->front.valid) release_front(it.e->front);
        c.bind();
        HIP_OK(hipStreamSynchronize(c.stream));
        c.front_stream.sync();
        c.front_stream_used = false;
      }
      if (!chain_step_covered(it.e, n_props, it.props, it.generator, theta_cur[b], theta_prop[b])) {
        // what the five merged launches do not cover takes the wide step (open targets, the Hausdorff evaluator, ranks up to 256, pose
        // moves), side by side with the other such chains of the batch that share the first one's model
        const bool same_model = wide_first < 0 || items[wide_first].e->ctx->Qp.p == c.Qp.p;
        if (!step_pipeline_covers(it.e, n_props, it.props) && same_model &&
            wide_chain_covered(it.e, n_props, it.props, it.generator, theta_cur[b], theta_prop[b])) {
          it.wide = true;
          if (wide_first < 0) wide_first = b;
          continue;  // (its lock stays held until the end of this call, as the batched chains')
        }
        it.lk.unlock();
        continue;
      }
      it.batched = true;
      ++n_batched;
    }
    // ---- per chain: host side of the step, launches captured
    g_batch_timing.start();
    caps.resize(n_batched > 0 ? n_batched : 1);
    // the decompositions of the chains that moved go out first, together, so that they run while the host prepares the
    // launches (a chain whose posteriors are not on record yet starts its own in enqueue_front)
    EigenCollect eigens{[&] { std::lock_guard<std::recursive_mutex> lead_lk(lead.mu); return batch_eigen_stream(lead, &elead); }(), {}, {}};
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      PosteriorEntry* ec[2] = {nullptr, nullptr};
      bool all = true;
      for (int i = 0; i < n_props; ++i) { ec[i] = it.props[i]->find_entry(theta_cur[b]); all = all && ec[i]; }
      if (!all) continue;
      Bound _b(&c, true);
      for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur[b]);
      // (ranks > 64 decompose through the library, each chain on its own eigen stream, and are awaited on the host below)
      start_decompositions(c, n_props, it.props, ec, c.stream_used_elsewhere, eigen_speculation_supported(c.r) ? &eigens : nullptr);
    }
    StepBatchGate gate{};
    if (!eigens.rq.empty()) {  // … in ONE launch, on the batch's eigen stream
      std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);  // (the record ring and the gate counter are the launch context's)
      Bound _b(&elead, true);
      const int turn = (lead.batch_eig_turn = (lead.batch_eig_turn + 1) % icp_ctx::kBatchRing);
      const size_t bytes = eigen_many_record_bytes((int)eigens.rq.size());
      if (bytes > lead.batch_eig_rec_bytes[turn]) {
        // (the slot's previous reader was the batch kBatchRing tickets ago: collected — tickets_in_flight —, its launches finished)
        if (lead.batch_eig_rec[turn]) { pinned_free(lead.batch_eig_rec[turn]); lead.batch_eig_rec[turn] = nullptr; }
        const size_t cap_bytes = std::max(bytes, eigen_many_record_bytes(128));
        pinned_alloc((void**)&lead.batch_eig_rec[turn], cap_bytes);
        lead.batch_eig_rec_bytes[turn] = cap_bytes;
      }
      if (!lead.batch_gate.p) {
        lead.batch_gate.alloc(16);
        lead.batch_gate.fill_bytes(0);  // (… and waits for the fill: the decompositions launched below count into it)
        pinned_alloc((void**)&lead.h_gate_error, sizeof(int) * 16);
        lead.h_gate_error[0] = 0;
        for (int k = 0; k < icp_ctx::kBatchRing; ++k) lead.batch_gate_expected[k] = 0;
      }
      // (the slot's own counter word: up to kBatchRing batches are in flight per launch context, and the workgroups of a LATER batch's
      // decompositions counting into one shared word could open an earlier batch's gate before its own decompositions are resident)
      const int wgs = launch_posterior_eigen_many(eigens.stream, elead.r, (int)eigens.rq.size(), eigens.rq.data(), lead.batch_eig_rec[turn],
                                                  lead.batch_gate.p + turn);
      require(wgs > 0, "internal: batched decompositions at a rank the kernel does not cover");
      HIP_OK(hipGetLastError());  // (a launch that failed would leave the gate below waiting for arrivals that never come)
      lead.batch_gate_expected[turn] = (int)((unsigned)lead.batch_gate_expected[turn] + (unsigned)wgs);  // (wraps with the counter)
      gate = StepBatchGate{lead.batch_gate.p + turn, lead.batch_gate_expected[turn], lead.h_gate_error};
      // test hook (tools/r3_timeout_repro.py: round 2's schedule, for the record): the launch sequence is not held back
      static const bool no_gate = dev_env("ICP_TEST_NO_GATE") != nullptr;
      if (no_gate) gate = StepBatchGate{};
      g_batch_timing.mark(4);
      BatchEventSlot& done = next_batch_event(elead.device);
      HIP_OK(hipEventRecord(done.ev, eigens.stream));
      for (PosteriorEntry* en : eigens.all) {  // (eig_event_valid is set where the requests were collected)
        en->eig_done_shared = done.ev;
        en->eig_shared_gen = &done.gen;
        en->eig_shared_gen_value = done.gen;
      }
      g_batch_timing.mark(5);
    }
    int nb = 0;
    struct FoldScope { FoldScope(int n) { tl_regression_posteriors = n; } ~FoldScope() { tl_regression_posteriors = 1; } } fold_scope(std::max(1, n_batched * n_props));
    struct SearchHintScope2 { SearchHintScope2(int n) { search_chains_hint(n); } ~SearchHintScope2() { search_chains_hint(1); } } search_hint_scope2(n_batched);
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      const int r = c.r;
      Bound _b(&c, true);
      for (int i = 0; i < n_props; ++i) it.props[i]->resolve_speculation(theta_cur[b]);
      bool other_work = c.stream_used_elsewhere;  // another entry point may still be busy on `stream` …
      for (int i = 0; i < n_props; ++i) other_work = other_work || !it.props[i]->find_entry(theta_cur[b]);  // … or is about to be
      StepCapture& cap = caps[nb];
      std::memset(cap.grid, 0, sizeof(cap.grid));
      struct CaptureScope { CaptureScope(StepCapture* c) { step_capture(c); } ~CaptureScope() { step_capture(nullptr); } } scope(&cap);
      enqueue_front(it.e, n_props, it.props, it.generator, theta_cur[b], it.key, it.F, true);
      it.issued = true;
      StepFront& F = it.F;
      // the batch runs on the first chain's stream: what it needs from this chain's own streams is awaited here
      if (other_work) HIP_OK(hipStreamSynchronize(c.stream));
      if (it.generator >= 0 && F.ec[it.generator]->done_value == 0) sync_eigen(c);
      for (int i = 0; i < 16; ++i) c.h_res[i] = 0.0;
      for (int i = 0; i < 16; ++i) c.h_status[i] = 0;
      const int step_seq = ++c.step_seq;
      StepFinishArgs& f = it.f;
      f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = it.props[i];
        f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
        f.M[i] = F.ep[i]->M.p; f.alpha[i] = F.ep[i]->alpha.p;
        f.status[i] = p->status.p + F.ep[i]->status_off;
        f.host_status[i] = c.h_status + 8 + i;
        f.fwd[i] = TransitionTailIO{F.ec[i]->alpha.p, F.ec[i]->M.p, F.ec[i]->coeffs.p, F.ep[i]->coeffs.p, p->prm.step_length,
                                    c.h_res + 8 + 2 * i, c.h_status + 2 * i};
        f.bwd[i] = TransitionTailIO{F.ep[i]->alpha.p, F.ep[i]->M.p, F.ep[i]->coeffs.p, F.ec[i]->coeffs.p, p->prm.step_length,
                                    c.h_res + 9 + 2 * i, c.h_status + 2 * i + 1};
      }
      f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = step_seq;
      f.ready_flag = c.d_done.p + 2;
      c.last_back_seq = step_seq;
      launch_step_finish(c.stream, f);  // (captured)
      it.f = cap.finish;                // as finalised by the launcher
      ++nb;
    }
    // ---- one sequence of launches for all of them
    g_batch_timing.mark(0);
    if (nb > 0) {
      // (the launch context may itself be busy — a member of an earlier batch on the same stream — but its stream and its
      // argument ring are used under its lock)
      std::lock_guard<std::recursive_mutex> lead_lk(lead.mu);
      Bound _b(&lead, true, true);
      const size_t bytes = step_batch_bytes(nb);
      const int turn = (lead.batch_turn = (lead.batch_turn + 1) % icp_ctx::kBatchRing);
      if (bytes > lead.batch_bytes[turn]) {
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (lead.batch_pinned[turn]) { pinned_free(lead.batch_pinned[turn]); lead.batch_pinned[turn] = nullptr; }
        const size_t cap_bytes = std::max(bytes, step_batch_bytes(16));
        pinned_alloc((void**)&lead.batch_pinned[turn], cap_bytes);
        lead.batch_device[turn].alloc(cap_bytes);
        lead.batch_bytes[turn] = cap_bytes;
      }
      static const bool finish_aside = dev_env("ICP_BATCH_FINISH_INLINE") == nullptr;  // (A/B switch)
      launch_step_batch(lead.stream, nb, caps.data(), lead.batch_pinned[turn], lead.batch_device[turn].p,
                        finish_aside ? lead.front_stream.get() : nullptr, lead.ev_join, gate);
      if (finish_aside) t.finish_stream = lead.front_stream.get();
    }
    t.nb = nb;
    t.lead = &lead;
    if (wide_first >= 0) wide_issue(t, lead, *items[wide_first].e->ctx);
    g_batch_timing.mark(1);
    // no mutex is held across the API boundary: the member contexts are marked busy instead (other entry points fail with
    // ICP_ERR_BUSY until the ticket is collected or abandoned — by any thread)
    for (auto& it : items)
      if (it.lk.owns_lock()) { it.e->ctx->batch_busy = true; it.lk.unlock(); }
  });
  if (rc != ICP_OK) {
    if (tk) { batch_release(*tk); delete tk; }
    return rc;
  }
  *out = tk;
  return ICP_OK;
}

int icp_chain_step_batched_collect(icp_step_ticket* tk) {
  if (!tk) return ICP_ERR_INVALID_ARG;
  icp_step_ticket& t = *tk;
  const int n_chains = t.n_chains, n_props = t.n_props, nb = t.nb;
  std::vector<BatchItem>& items = t.items;
  const double* const* theta_cur = t.theta_cur.data();
  const double* const* z = t.z.data();
  double* const* theta_prop = t.theta_prop.data();
  double* log_value_prop = t.log_value_prop;
  double* fwd = t.fwd;
  double* bwd = t.bwd;
  int32_t* status = t.status;
  typedef BatchItem Item;
  int rc = guard([&] {
    icp_ctx& lead = *t.lead;
    g_batch_timing.start();
    if (lead.h_gate_error && lead.h_gate_error[0]) {  // (counted; the launches went ahead and their own time-outs take it from there)
      lead.h_gate_error[0] = 0;
      ++lead.stats.gate_timeouts; ++g_runtime_stats.gate_timeouts;
    }
    // ---- results, chain by chain
    bool first_wait = true;
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.batched) continue;
      icp_ctx& c = *it.e->ctx;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      Bound _b(&c, true, true);
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != it.f.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(2)) break;
      }
      if (*flag != it.f.seq) {
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (t.finish_stream) HIP_OK(hipStreamSynchronize(t.finish_stream));
      }
      if (first_wait) { g_batch_timing.mark(2); first_wait = false; }
      if (it.F.eigen_first_use && !eigen_speculation_supported(c.r)) {  // (its status is not written to pinned memory)
        sync_proposal_status_if(it.props[it.generator], true);
        c.finish(0, 0);
      }
      c.stage_used = 0;
      if (c.h_wait_error[0]) {  // the decomposition this chain draws from did not finish in time (a tool that serialises kernels)
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (t.finish_stream) HIP_OK(hipStreamSynchronize(t.finish_stream));
        sync_eigen(c);
        c.h_wait_error[0] = 0;
        ++c.stats.wait_timeouts; ++g_runtime_stats.wait_timeouts;
        it.redo = true;
      } else {
        int st = ICP_OK;
        it.redo = !chain_step_record(it.e, n_props, it.props, it.generator, theta_cur[b], it.F, it.f, theta_prop[b], log_value_prop + b,
                                     fwd + (size_t)b * n_props, bwd + (size_t)b * n_props, &st);
        status[b] = st;
      }
      if (it.redo) { ++c.stats.step_redos; ++g_runtime_stats.step_redos; }
      if (it.redo) release_front(it.F);
      else {
        ++c.paths.n[0]; ++g_step_paths.n[0];
        it.F.s->reserved = false;
        for (int i = 0; i < n_props; ++i) it.F.ep[i]->reserved = false;
      }
      it.issued = false;
      c.batch_busy = false;
      it.lk.unlock();
    }
    // ---- the chains that took the wide step
    for (int b = 0; b < n_chains; ++b) {
      Item& it = items[b];
      if (!it.wide) continue;
      icp_ctx& c = *it.e->ctx;
      it.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      Bound _b(&c, true, true);
      volatile int* flag = c.h_flag;
      const auto t_start = std::chrono::steady_clock::now();
      long spins = 0;
      while (*flag != it.W.seq) {
        if ((++spins & 0xFFFF) == 0 && std::chrono::steady_clock::now() - t_start > std::chrono::seconds(5)) break;
      }
      if (*flag != it.W.seq) {
        for (hipStream_t ws : t.wide_streams)
          if (ws) HIP_OK(hipStreamSynchronize(ws));
        if (*flag != it.W.seq) fail(ICP_ERR_DEVICE, "internal: a wide step's completion flag did not arrive");
      }
      c.stage_used = 0;
      int st = ICP_OK;
      it.redo = !wide_record(t, b, log_value_prop + b, fwd + (size_t)b * n_props, bwd + (size_t)b * n_props, &st);
      status[b] = st;
      if (it.redo) { ++c.stats.step_redos; ++g_runtime_stats.step_redos; }
      wide_release(it, !it.redo);
      it.issued = false;
      c.batch_busy = false;
      it.lk.unlock();
    }
    g_batch_timing.mark(3);
    if (g_batch_timing.on) { ++g_batch_timing.calls; g_batch_timing.chains += nb; }
  });
  batch_release(t);
  if (rc != ICP_OK) { delete tk; return rc; }
  // ---- the others, and whatever has to be done again, one after the other
  int first_bad = ICP_OK;
  for (int b = 0; b < n_chains; ++b) {
    Item& it = items[b];
    if ((it.batched || it.wide) && !it.redo) continue;
    if (g_batch_timing.on) ++g_batch_timing.stepped_alone;
    const int st = icp_chain_step(it.e, n_props, it.props, it.generator, theta_cur[b], it.generator >= 0 ? z[b] : nullptr, theta_prop[b],
                                  log_value_prop + b, fwd + (size_t)b * n_props, bwd + (size_t)b * n_props);
    status[b] = st;
  }
  for (int b = 0; b < n_chains; ++b)
    if (status[b] != ICP_OK && status[b] != ICP_ERR_EMPTY && first_bad == ICP_OK) first_bad = status[b];
  delete tk;
  return first_bad;
}

int icp_chain_step_batched_abandon(icp_step_ticket* tk) {
  if (!tk) return ICP_ERR_INVALID_ARG;
  batch_release(*tk);  // waits for the batch's launches, gives back what they hold; nothing of the step is recorded
  delete tk;
  return ICP_OK;
}
}  // extern "C"
