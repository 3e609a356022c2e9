// abi_device_loop.inl — part of icp_abi.hip (one translation unit; included there, in order).
// C ABI: icp_chains_run_on_device (the whole MH loop on the device) and the remaining queries
// --------------------------------------------------------------------- the whole MH loop on the device, WIDE step (MhWide)
// Chains whose step is the wide one (open targets, the Hausdorff evaluator, ranks 65..256: apps/bfm/BfmFittingPartial.scala:62-96,
// apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala at rank 101).  Per step the wide step's own launches, from records that live
// in device memory and are the same every step: the current state's posteriors stay in the entries they are in — an accepted state's
// M, alpha, coefficients and correspondence records are copied there (k_mhw_adopt) —, the proposed state always has the same slot and
// entries.  What changes per step (the proposal's inputs, the proposed state's pose) is set by k_mhw_front; k_mhw_decide is
// MetropolisHastings.next.  The records are wide_issue's own, captured (WideCapture).
//   ranks <= 64: groups of chains, each in order on one stream; the warm-started Jacobi iteration in place behind the decision;
//   ranks above: ONE group; behind the summed partials the step forks — factorisations + tails | the PROPOSED states' decompositions
//   (tridiagonal route, started ahead of the decision; an accepted state's basis is copied with its posterior) | the evaluator's
//   searches — and joins at the decision (tails) and at the hand-over (decompositions).  DESIGN.md §5.3b.
namespace {
int wide_chains_run_on_device(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props_in,
                              const icp_mh_mixture* mix, const uint64_t* seeds, const int64_t* first_step, double* const* theta,
                              double* log_value, int32_t n_steps, double* const* records, int64_t* accepted) {
  struct Chain {
    icp_evaluator* e = nullptr;
    icp_proposal* props[2] = {nullptr, nullptr};
    PosteriorEntry* cur[2] = {nullptr, nullptr};
    bool busy = false;
    bool unfilled = false;  // the current state's posteriors are not on record: filled by the run's first pass
  };
  struct Group {
    int b0 = 0, B = 0;
    hipStream_t st = nullptr;
    icp_step_ticket ticket;   // (holds the captured step's reservations: state slot, the proposed state's entries)
    WideCapture cap;
    DBuf<char> wide_dev;
    DBuf<WideProposeItem> items;
    DBuf<MhWide> wide;
    DBuf<MhChain> mh;
    DBuf<MhAdopt> adopt;
    DBuf<EigenProblem> eig_rec;
    DBuf<int> eig_skip;
    std::vector<EigenRequest> rqs;   // (ranks above 64: the tridiagonal route's by-value requests …
    std::vector<const double*> spec_parts;  // … and the summed partials their matrices are assembled from)
    DBuf<double> given, normals[2], theta, rec, res;
    DBuf<int> stat;
    double* h_normals[2] = {nullptr, nullptr};
    hipEvent_t ev_copy[2] = {nullptr, nullptr};
    std::vector<std::vector<double>> theta_prop_scratch;
    // (ranks above 64) the streams of the step's independent branches — the group's first chain's own — and the events between them
    hipStream_t side[3] = {nullptr, nullptr, nullptr};
    hipEvent_t ev_sum = nullptr, ev_tails = nullptr, ev_decide = nullptr, ev_prop = nullptr, ev_inst = nullptr, ev_eig[2] = {nullptr, nullptr};
    int n_eig_streams = 0;
    bool eig_split = false;
  };
  std::vector<Chain> chains;
  std::vector<std::unique_ptr<Group>> groups;
  auto release = [&]() {
    for (auto& gp : groups)
      if (gp)
        for (auto& it : gp->ticket.items) {
          if (!it.e) continue;
          std::lock_guard<std::recursive_mutex> lk(it.e->ctx->mu);
          wide_release(it);
        }
    for (auto& ch : chains) {
      if (!ch.e) continue;
      icp_ctx& c = *ch.e->ctx;
      std::lock_guard<std::recursive_mutex> lk(c.mu);
      for (int i = 0; i < 2; ++i)
        if (ch.cur[i]) ch.cur[i]->reserved = false;
      if (ch.busy) c.batch_busy = false;
    }
  };
  struct GroupGuard {
    std::vector<std::unique_ptr<Group>>& g;
    ~GroupGuard() {
      for (auto& gp : g) {
        if (!gp) continue;
        for (int k = 0; k < 2; ++k) {
          if (gp->h_normals[k]) pinned_free(gp->h_normals[k]);
          if (gp->ev_copy[k]) (void)hipEventDestroy(gp->ev_copy[k]);
          if (gp->ev_eig[k]) (void)hipEventDestroy(gp->ev_eig[k]);
        }
        if (gp->ev_sum) (void)hipEventDestroy(gp->ev_sum);
        if (gp->ev_tails) (void)hipEventDestroy(gp->ev_tails);
        if (gp->ev_decide) (void)hipEventDestroy(gp->ev_decide);
        if (gp->ev_prop) (void)hipEventDestroy(gp->ev_prop);
        if (gp->ev_inst) (void)hipEventDestroy(gp->ev_inst);
      }
    }
  };
  int rc = guard([&] {
    require(n_chains >= 1 && evaluators && props_in && mix && seeds && first_step && theta && log_value && n_steps >= 0, "null argument");
    require(n_props >= 1 && n_props <= 2, "the on-device loop takes one or two ICP proposals per chain");
    require(mix->struct_size == sizeof(icp_mh_mixture), "icp_mh_mixture::struct_size does not match this library's header");
    require(mix->w_icp > 0.0 && mix->w_rw >= 0.0 && mix->rw_sigma > 0.0 && mix->w_pose >= 0.0, "bad mixture");
    if (mix->w_pose > 0.0)
      for (int a = 0; a < 3; ++a) require(mix->pose_rot_sigma[a] > 0.0 && mix->pose_trans_sigma[a] > 0.0, "pose walk sigmas must be positive");
    chains.resize(n_chains);
    icp_ctx& lead = *evaluators[0]->ctx;
    const int r = lead.r, P = 10 + r;
    const int root = props_in[0]->sampler == ICP_SAMPLER_CHOLESKY_ROOT;
    const bool jacobi = eigen_speculation_supported(r);
    const bool timing = dev_env("ICP_WIDE_LOOP_TIMING") != nullptr;
    auto t_phase = std::chrono::steady_clock::now();
    auto phase = [&](const char* what) {
      if (!timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[icp wide loop] %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(now - t_phase).count());
      t_phase = now;
    };
    require(jacobi || (eigen_tridiag_many_supported(r) && !root),
            "the on-device loop of the wide step covers ranks 3..256 (the Cholesky-root sampler up to rank 64)");
    // ---- claim the chains' contexts; the current state's posteriors and their bases, the ordinary way
    for (int b = 0; b < n_chains; ++b) {
      Chain& ch = chains[b];
      require(evaluators[b] && theta[b], "null argument");
      icp_ctx& c = *evaluators[b]->ctx;
      require(c.device == lead.device && c.r == r && c.Qp.p == lead.Qp.p, "chains of one run share a device, a rank and (wide step) a model");
      for (int a = 0; a < b; ++a) require(chains[a].e->ctx != &c, "every chain needs a context of its own");
      ch.e = evaluators[b];
      for (int i = 0; i < n_props; ++i) {
        ch.props[i] = props_in[(size_t)b * n_props + i];
        require(ch.props[i] && ch.props[i]->ctx == &c, "proposal belongs to another context");
        require(ch.props[i]->sampler == props_in[0]->sampler, "one sampler per run");
      }
      check_theta_finite(&c, theta[b]);
      std::lock_guard<std::recursive_mutex> lk(c.mu);
      if (c.batch_busy) fail(ICP_ERR_BUSY, "a chain's context already belongs to a batch in flight");
      require((!step_pipeline_covers(ch.e, n_props, ch.props) || !jacobi) && wide_pipeline_covers(ch.e, n_props, ch.props),
              "chains of one run take the same kind of step (this run: the wide one)");
      if (mix->w_pose > 0.0)
        require(c.rotations_mismatched == 0, "pose walks on the device: a caller-supplied rotation matrix disagreed with the library's Rz·Ry·Rx (icp_ctx_rotation_convention)");
      Bound _b(&c);
      // the context is this run's from here on (as the merged loop claims it): between this point and the captured step the locks are
      // released and taken again, and another thread's call on the context must be refused, not let in (release() clears the mark)
      c.batch_busy = true;
      ch.busy = true;
      if (ch.e->front.valid) release_front(ch.e->front);
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = ch.props[i];
        p->resolve_speculation(theta[b]);
        // Above rank 64 a state whose posterior is not on record is NOT computed here, chain after chain the per-stage way (2.5 ms each
        // at rank 200): the captured step gets an empty entry for it, and one pass of the run's own launches fills the entries of all
        // chains at once (below: "restate")
        if (!jacobi && !p->find_entry(theta[b])) { ch.unfilled = true; continue; }
        PosteriorEntry& cur = p->posterior(theta[b], false);
        p->ensure_eigen(cur);
        cur.reserved = true;
        ch.cur[i] = &cur;
      }
      if (ch.unfilled)
        for (int i = 0; i < n_props; ++i)
          if (ch.cur[i]) { ch.cur[i]->reserved = false; ch.cur[i] = nullptr; }  // (all of the chain's entries the same way)
    }
    // (… every chain's work is on its own streams by now, side by side: waited for and looked at chain by chain)
    for (int b = 0; b < n_chains; ++b) {
      Chain& ch = chains[b];
      icp_ctx& c = *ch.e->ctx;
      std::lock_guard<std::recursive_mutex> lk(c.mu);
      Bound _b(&c, false, true);
      HIP_OK(hipStreamSynchronize(c.stream));
      c.front_stream.sync();
      sync_eigen(c);
      if (ch.unfilled) continue;
      for (int i = 0; i < n_props; ++i) sync_proposal_status(ch.props[i]);
      HIP_OK(hipStreamSynchronize(c.stream));
      for (int i = 0; i < n_props; ++i) {
        ch.props[i]->check_status(*ch.cur[i]);
        if (ch.props[i]->h_eig[ch.cur[i]->status_off / 3] != 0) fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
        ch.cur[i]->eig_checked = true;
      }
    }
    phase("claim (current states' posteriors and bases)");
    // ---- groups: each its own stream, everything of a group in order on it (the first chain's three streams: see the merged loop)
    static const int forced_groups = dev_env("ICP_WIDE_LOOP_GROUPS") ? std::atoi(dev_env("ICP_WIDE_LOOP_GROUPS")) : 0;  // (developer sweep)
    // (every launch of a wide step costs about the same whatever it carries — one-workgroup factorisations and reductions, searches
    // bound by their longest candidate list —: three groups of 10 face chains measured 7.8k it/s, one group of 30 7.2k with everything in
    // stream order, tools/r5_wide_loop_rate.sh.  Ranks above 64 therefore run as ONE group whose independent branches take streams of
    // their own — see the loop —; the warm-started iteration of ranks <= 64 keeps the groups of the merged loop.)
    // (Two such groups a phase apart — one group's decompositions beside the other's searches — measured 2.4 ms per step of 30 face chains
    // against 1.7 with one: the chip-wide launches of one group hold up the other's one-workgroup kernels.)
    const int n_groups = forced_groups > 0 ? std::min(std::min(forced_groups, jacobi ? 3 : 2), n_chains) : (!jacobi ? 1 : n_chains >= 12 ? 3 : 1);
    groups.resize(n_groups);
    GroupGuard group_guard{groups};
    constexpr int kChunk = 64;  // steps per block of standard normals
    constexpr int tri_chunk = kTriManyMax;  // (requests per launch of the tridiagonal route)
    std::vector<double> zero_z(r, 0.0);
    lead.bind();
    for (int g = 0; g < n_groups; ++g) {
      groups[g].reset(new Group());
      Group& gr = *groups[g];
      gr.b0 = (int)((long long)g * n_chains / n_groups);
      gr.B = (int)((long long)(g + 1) * n_chains / n_groups) - gr.b0;
      if (jacobi) gr.st = g == 0 ? lead.stream : g == 1 ? lead.front_stream.get() : lead.eig_stream.get();
      else {
        icp_ctx& gc = *chains[gr.b0].e->ctx;
        std::lock_guard<std::recursive_mutex> glk(gc.mu);  // (LazyStream::get makes the stream on first use: not without the context's lock)
        gr.st = gc.stream;
        gr.side[0] = gc.front_stream.get(); gr.side[1] = gc.eig_stream.get(); gr.side[2] = gc.eig_stream2.get();
        HIP_OK(hipEventCreateWithFlags(&gr.ev_sum, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&gr.ev_tails, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&gr.ev_decide, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&gr.ev_prop, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&gr.ev_inst, hipEventDisableTiming));
        for (auto& e : gr.ev_eig) HIP_OK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      }
      const int B = gr.B;
      // -- the step of every chain of the group, captured: an ICP move from the current state
      icp_step_ticket& t = gr.ticket;
      t.n_chains = B; t.n_props = n_props;
      t.items.resize(B);
      t.theta_cur.resize(B); t.theta_prop.resize(B); t.z.assign(B, zero_z.data());
      gr.theta_prop_scratch.assign(B, std::vector<double>(P, 0.0));
      std::vector<std::unique_lock<std::recursive_mutex>> locks;
      for (int k = 0; k < B; ++k) {
        Chain& ch = chains[gr.b0 + k];
        BatchItem& it = t.items[k];
        it.e = ch.e;
        it.props = props_in + (size_t)(gr.b0 + k) * n_props;
        it.generator = 0;
        it.key = zero_z.data();
        it.wide = true;
        t.theta_cur[k] = theta[gr.b0 + k];
        t.theta_prop[k] = gr.theta_prop_scratch[k].data();
        locks.emplace_back(ch.e->ctx->mu);
      }
      icp_ctx& glead = *chains[gr.b0].e->ctx;
      for (int k = 0; k < B; ++k)  // (entries of a chain that is filled below: not to be handed out as another's proposed-state entry)
        if (chains[gr.b0 + k].unfilled)
          for (int i = 0; i < n_props; ++i)
            if (PosteriorEntry* en = chains[gr.b0 + k].props[i]->find_entry(theta[gr.b0 + k])) {
              en->valid = false; en->eig_valid = false; en->eig_checked = false; en->eig_event_valid = false; en->done_value = 0;
            }
      gr.cap.allow_unfilled = !jacobi;
      wide_issue(t, glead, glead, &gr.cap);
      WideCapture& cap = gr.cap;
      phase("capture");
      if (dev_env("ICP_WIDE_LOOP_TIMING")) std::fprintf(stderr, "[icp wide loop] instance head %d of %d blocks (any_split %d)\n", cap.plan.inst_head_blocks, (cap.plan.N + 63) / 64, (int)cap.any_split);
      require((int)cap.chain_args.size() == B && (int)cap.prop_items.size() == B && (int)cap.factors.size() == B * n_props &&
              (int)cap.tails.size() == 2 * B * n_props, "internal: captured wide step is incomplete");
      for (int k = 0; k < B; ++k) {
        Chain& ch = chains[gr.b0 + k];
        WideItem& w = t.items[k].W;
        for (int i = 0; i < n_props; ++i) {
          if (ch.unfilled) ch.cur[i] = w.ec[i];
          require(w.ec[i] && w.ec[i] == ch.cur[i] && w.ep[i] && w.ep[i] != w.ec[i], "internal: captured wide step lost the current state's posterior");
          ch.cur[i]->reserved = true;
        }
      }
      locks.clear();
      lead.bind();
      // -- device data
      gr.items.alloc(B); gr.wide.alloc(B); gr.mh.alloc(B);
      gr.adopt.alloc((size_t)B * n_props); gr.eig_rec.alloc((size_t)B * n_props); gr.eig_skip.alloc((size_t)B * n_props);
      gr.given.alloc((size_t)B * r); gr.theta.alloc((size_t)B * P);
      gr.res.alloc((size_t)B * 32); gr.res.fill_bytes(0);
      gr.stat.alloc((size_t)B * 16); gr.stat.fill_bytes(0);
      gr.rec.alloc(records ? (size_t)B * std::max(n_steps, 1) * (4 + P) : 1);
      for (int k = 0; k < 2; ++k) {
        gr.normals[k].alloc((size_t)B * kChunk * r);
        pinned_alloc((void**)&gr.h_normals[k], sizeof(double) * (size_t)B * kChunk * r);
        HIP_OK(hipEventCreateWithFlags(&gr.ev_copy[k], hipEventDisableTiming));
      }
      const size_t wbytes = wide_batch_bytes(B);
      gr.wide_dev.alloc(wbytes);
      std::vector<char> hw(wbytes, 0);
      wide_pack_args(B, cap.chain_args.data(), hw.data());
      std::vector<WideProposeItem> hitems(B);
      std::vector<MhWide> hwide(B);
      std::vector<MhChain> hm(B);
      std::vector<MhAdopt> had((size_t)B * n_props);
      std::vector<EigenProblem> hrec((size_t)B * n_props);
      std::vector<int> hskip((size_t)B * n_props, 1);
      std::vector<double> hth((size_t)B * P);
      gr.rqs.resize((size_t)B * n_props);
      gr.spec_parts.resize((size_t)B * n_props);
      for (size_t q = 0; q < gr.spec_parts.size(); ++q) gr.spec_parts[q] = cap.factors[q].Mpart;
      for (int k = 0; k < B; ++k) {
        Chain& ch = chains[gr.b0 + k];
        icp_ctx& c = *ch.e->ctx;
        WideItem& w = t.items[k].W;
        MhChain& m = hm[k];
        std::memset(&m, 0, sizeof(m));
        // W1: the copy of the proposed coefficients for the host (the record's last output, pinned memory) is not made
        WideProposeItem pi = cap.prop_items[k];
        pi.n_out -= 1;
        pi.kind = 0; pi.src = gr.given.p + (size_t)k * r;
        hitems[k] = pi;
        MhWide& mw = hwide[k];
        std::memset(&mw, 0, sizeof(mw));
        mw.item = gr.items.p + k;
        mw.given = gr.given.p + (size_t)k * r;
        mw.inst = wide_inst_record(gr.wide_dev.p, B, k);
        mw.search[0] = wide_search_record(gr.wide_dev.p, B, 0, k);
        mw.search[1] = wide_search_record(gr.wide_dev.p, B, 1, k);
        for (int i = 0; i < n_props; ++i) {
          icp_proposal* p = ch.props[i];
          PosteriorEntry& cur = *w.ec[i];
          PosteriorEntry& prop = *w.ep[i];
          mw.prop_in[i] = ProposeIn{cur.alpha.p, cur.V.p, cur.S.p, c.inv_sqrt_lambda.p, c.P.p, cur.coeffs.p, nullptr, kSigma2, p->prm.step_length, root};
          MhAdopt ad{};
          ad.M_from = prop.M.p; ad.M_to = cur.M.p; ad.alpha_from = prop.alpha.p; ad.alpha_to = cur.alpha.p; ad.c_from = prop.coeffs.p; ad.c_to = cur.coeffs.p;
          // Ranks up to 64: the decomposition of the current state's posterior IN PLACE, behind the decision (an accepted state has just
          // arrived there; warm start: the basis that is there).  Above: the tridiagonal route — 0.7 ms at rank 200 — on the PROPOSED
          // state's entry, started as soon as the summed partials exist, beside the factorisation, the tails and the evaluator's
          // searches (as the host-stepped wide step decomposes ahead); an accepted state's basis moves with its posterior.
          PosteriorEntry& de = jacobi ? cur : prop;
          // (ranks above 64: no host status, no completion word — the loop looks at the status words on the device, in stream order)
          EigenRequest rq{de.M.p, (root || !jacobi) ? nullptr : cur.V.p, de.V.p, de.Vt.p, de.S.p, p->work.p, p->status.p + de.status_off + 2, nullptr,
                          jacobi ? p->h_eig + de.status_off / 3 : nullptr, jacobi ? p->eig_words.p + de.status_off / 3 : nullptr, 0, c.sqrt_lambda.p};
          rq.root = root != 0;
          gr.rqs[(size_t)k * n_props + i] = rq;
          EigenProblem ep{};
          if (jacobi) ep = eigen_problem_of(r, rq);
          else {
            ep.status = p->status.p + cur.status_off + 2;  // (k_mhw_decide looks at the CURRENT state's status word only)
            ad.V_from = prop.V.p; ad.V_to = cur.V.p; ad.Vt_from = prop.Vt.p; ad.Vt_to = cur.Vt.p; ad.S_from = prop.S.p; ad.S_to = cur.S.p;
            ad.st_from = p->status.p + prop.status_off + 2; ad.st_to = p->status.p + cur.status_off + 2;
          }
          {
            const CorrBuffers cf = prop.corr(), ct = cur.corr();
            const size_t K = (size_t)std::max(p->K, 0);
            const void* from[6] = {cf.id, cf.aux, cf.pt, cf.keep, cf.nhat, cf.e};
            void* to[6] = {ct.id, ct.aux, ct.pt, ct.keep, ct.nhat, ct.e};
            const size_t bytes[6] = {sizeof(int) * K, sizeof(int) * K, sizeof(double) * 3 * K, K, sizeof(double) * 3 * K, sizeof(double) * 3 * K};
            for (int u = 0; u < 6; ++u) {
              ad.corr_from[u] = (const unsigned char*)from[u]; ad.corr_to[u] = (unsigned char*)to[u];
              ad.corr_bytes[u] = (from[u] && to[u]) ? (int)bytes[u] : 0;
            }
          }
          had[(size_t)k * n_props + i] = ad;
          hrec[(size_t)k * n_props + i] = ep;
          m.eig_alt[0][i] = ep; m.eig_alt[1][i] = ep;
          // the tails' results and status words in DEVICE memory (the decide kernel reads them)
          TransitionTailIO& fw = cap.tails[2 * ((size_t)k * n_props + i)];
          TransitionTailIO& bw = cap.tails[2 * ((size_t)k * n_props + i) + 1];
          fw.out = gr.res.p + (size_t)k * 32 + 8 + 2 * i; fw.status = gr.stat.p + (size_t)k * 16 + 2 * i;
          bw.out = gr.res.p + (size_t)k * 32 + 9 + 2 * i; bw.status = gr.stat.p + (size_t)k * 16 + 2 * i + 1;
          mw.chol[i] = cap.factors[(size_t)k * n_props + i].status;
          m.eig_seq[i] = p->eig_seq;
        }
        m.wide = gr.wide.p + k;
        m.eig_live = gr.eig_rec.p + (size_t)k * n_props;
        m.eig_skip = gr.eig_skip.p + (size_t)k * n_props;
        m.pw_id_mask = 2046;
        m.seed = seeds[gr.b0 + k];
        m.r = r; m.n_icp = n_props;
        {  // MixtureProposal weights, normalised as the harness normalises them (host/icp_host.hpp: pick_component)
          double ws = 0.0;
          for (int i = 0; i < n_props; ++i) ws += mix->icp_weight[i];
          for (int i = 0; i < n_props; ++i) m.icp_w[i] = mix->icp_weight[i] / ws;
          double raw[3];
          int no = 0;
          if (mix->w_pose > 0.0) { m.outer_kind[no] = 0; raw[no++] = mix->w_pose; }  // (BfmFittingPartial.scala:70: pose, ICP, shape walk)
          m.outer_kind[no] = 1; raw[no++] = mix->w_icp;
          if (mix->w_rw > 0.0) { m.outer_kind[no] = 2; raw[no++] = mix->w_rw; }
          double wsum = 0.0;
          for (int o = 0; o < no; ++o) wsum += raw[o];
          for (int o = 0; o < no; ++o) m.outer_w[o] = raw[o] / wsum;
          m.n_outer = no;
        }
        if (mix->w_pose > 0.0) {  // MixedProposalDistributions.scala:29-39 / host/icp_host.cpp: mixed_random_pose_proposal
          static const int param_index[6] = {6, 5, 4, 1, 2, 3};  // yaw = rotation._3, pitch = _2, roll = _1 (PoseProposals.scala:39-41); x, y, z
          m.n_pose = 6;
          double wsum = 0.0;
          for (int a = 0; a < 6; ++a) wsum += 0.5;
          for (int a = 0; a < 6; ++a) {
            const double sd = a < 3 ? mix->pose_rot_sigma[a] : mix->pose_trans_sigma[a - 3];
            m.pose_index[a] = param_index[a];
            m.pose_w[a] = 0.5 / wsum;
            m.pose_sigma[a] = sd;
            m.pose_logc[a] = std::log(std::sqrt(2.0 * M_PI)) + std::log(sd);  // breeze Gaussian(0, σ).logPdf's normaliser
          }
        }
        m.front_every_step = 1;
        m.rw_sigma = mix->rw_sigma;
        m.rw_logc = 0.5 * (r * std::log(2.0 * M_PI) + r * std::log(mix->rw_sigma * mix->rw_sigma));
        m.prior_c = 0.5 * r * std::log(2.0 * M_PI);
        const icp_evaluator_params& ep = ch.e->prm;
        m.eval_kind = ep.kind;
        m.eval_mode = ep.mode;
        m.gauss_mean = ep.gauss_mean; m.gauss_sigma = ep.gauss_sigma;
        m.gauss_logn = std::log(std::sqrt(2.0 * M_PI)) + std::log(ep.gauss_sigma);
        m.exp_rate = ep.exp_rate; m.exp_lograte = std::log(ep.exp_rate);
        m.coeff_prop = w.s->coeffs.p;
        m.red = c.d_res.p;  // (W8's reductions: device memory already)
        m.tails = gr.res.p + (size_t)k * 32 + 8;
        m.tail_status = gr.stat.p + (size_t)k * 16;
        m.chol_status = nullptr;  // (MhWide::chol: the factorisations report into their proposals' own status buffers)
        m.normals = nullptr; m.normals_first = 0; m.normals_rows = 0;
        m.records = records && records[gr.b0 + k] ? gr.rec.p + (size_t)k * n_steps * (4 + P) : nullptr;
        m.rec_first = first_step[gr.b0 + k];
        m.theta = gr.theta.p + (size_t)k * P;
        m.cur_p = log_value[gr.b0 + k];
        m.step = first_step[gr.b0 + k];
        m.accepted = 0; m.cur_sel = 0; m.gen = -1; m.leaf = -1; m.error = 0;
        std::memcpy(hth.data() + (size_t)k * P, theta[gr.b0 + k], sizeof(double) * P);
      }
      HIP_OK(hipMemcpy(gr.wide_dev.p, hw.data(), wbytes, hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.items.p, hitems.data(), sizeof(WideProposeItem) * hitems.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.wide.p, hwide.data(), sizeof(MhWide) * hwide.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.adopt.p, had.data(), sizeof(MhAdopt) * had.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.eig_rec.p, hrec.data(), sizeof(EigenProblem) * hrec.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.eig_skip.p, hskip.data(), sizeof(int) * hskip.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.theta.p, hth.data(), sizeof(double) * hth.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.mh.p, hm.data(), sizeof(MhChain) * hm.size(), hipMemcpyHostToDevice));
      HIP_OK(hipStreamSynchronize(nullptr));
      phase("records");
      // -- "restate": the current states' posteriors and bases of a group with unfilled entries, by the step's own launches — the
      // proposal is the current state itself, its posterior lands in the proposed state's entries, is decomposed there and handed over
      // like an accepted state's (every chain of the group: the values of an entry that was on record are computed again, the same)
      if (cap.any_unfilled) {
        const hipStream_t S = gr.st;
        const int nq = B * n_props;
        launch_mhw_front(S, B, gr.mh.p, 1);
        launch_wide_propose_resident(S, r, B, gr.items.p);
        launch_wide_head_resident(S, cap.plan, gr.wide_dev.p);
        launch_wide_main(S, cap.plan, gr.wide_dev.p);
        for (size_t p0 = 0; p0 < cap.sum_parts.size(); p0 += kWideMaxChains)
          launch_sum_partials_many(S, r, (int)std::min<size_t>(kWideMaxChains, cap.sum_parts.size() - p0), cap.sum_parts.data() + p0, cap.sum_splits.data() + p0);
        HIP_OK(hipEventRecord(gr.ev_sum, S));
        int used = 0;
        for (int q0 = 0; q0 < nq; q0 += tri_chunk, ++used) {
          const hipStream_t E = gr.side[1 + (used & 1)];
          if (used < 2) HIP_OK(hipStreamWaitEvent(E, gr.ev_sum, 0));
          launch_posterior_eigen_tridiag_many(E, r, std::min(tri_chunk, nq - q0), gr.rqs.data() + q0, gr.spec_parts.data() + q0, nullptr);
        }
        for (int u = 0; u < std::min(used, 2); ++u) HIP_OK(hipEventRecord(gr.ev_eig[u], gr.side[1 + u]));
        const size_t fmax0 = (size_t)posterior_factor_max();
        for (size_t p0 = 0; p0 < cap.factors.size(); p0 += fmax0)
          launch_posterior_factor(S, r, (int)std::min(fmax0, cap.factors.size() - p0), cap.factors.data() + p0);
        for (int u = 0; u < std::min(used, 2); ++u) HIP_OK(hipStreamWaitEvent(S, gr.ev_eig[u], 0));
        launch_mhw_adopt(S, r, nq, gr.adopt.p, nullptr);
        HIP_OK(hipStreamSynchronize(S));
        // the factorisations' and decompositions' status words, as the per-stage path would have looked at them
        for (int k = 0; k < B; ++k)
          for (int i = 0; i < n_props; ++i) {
            int st[3] = {0, 0, 0};
            HIP_OK(hipMemcpy(st, cap.factors[(size_t)k * n_props + i].status, sizeof(int) * 3, hipMemcpyDeviceToHost));
            if (st[0] != 0) fail(ICP_ERR_NOT_SPD, "posterior matrix is not positive definite");
            if (st[2] != 0) fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
          }
        phase("restate (current states' posteriors, all chains at once)");
      }
    }
    auto draw_block = [&](Group& gr, int blk, int buf) {
      const int s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      double* out = gr.h_normals[buf];
      for (int k = 0; k < gr.B; ++k) {
        const uint64_t seed = seeds[gr.b0 + k];
        const uint64_t f0 = (uint64_t)first_step[gr.b0 + k] + (uint64_t)s0;
        for (int s_ = 0; s_ < ns; ++s_)
          for (int j = 0; j < r; ++j) out[((size_t)k * kChunk + s_) * r + j] = harness_normal(seed, f0 + (uint64_t)s_, (uint64_t)j);
      }
    };
    const int n_blocks = (n_steps + kChunk - 1) / kChunk;
    struct ProfBind { ProfBind(icp_ctx& c) { g_prof = c.profiling ? &c.prof : nullptr; } ~ProfBind() { g_prof = nullptr; } } prof_bind(lead);
    const size_t fmax = (size_t)posterior_factor_max();
    const auto t_loop0 = std::chrono::steady_clock::now();
    for (int blk = 0; blk < n_blocks; ++blk) {
      const int buf = blk & 1, s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      for (auto& gp : groups) {
        Group& gr = *gp;
        HIP_OK(hipEventSynchronize(gr.ev_copy[buf]));  // (the staging buffer's previous upload has left it)
        draw_block(gr, blk, buf);
        HIP_OK(hipMemcpyAsync(gr.normals[buf].p, gr.h_normals[buf], sizeof(double) * (size_t)gr.B * kChunk * r, hipMemcpyHostToDevice, gr.st));
        HIP_OK(hipEventRecord(gr.ev_copy[buf], gr.st));
        launch_mh_set_normals(gr.st, gr.B, gr.mh.p, gr.normals[buf].p, kChunk * r, s0, ns);
      }
      for (int s_ = 0; s_ < ns; ++s_)
        for (auto& gp : groups) {
          Group& gr = *gp;
          const hipStream_t S = gr.st;
          const WideCapture& cap = gr.cap;
          const int nq = gr.B * n_props;
          launch_mhw_front(S, gr.B, gr.mh.p);
          launch_wide_propose_resident(S, r, gr.B, gr.items.p);
          // the proposed states' instances.  The main sequence (the proposals' K model ids against the target, their regression) reads the
          // first blocks of model points only — the ids and the corners of their triangles (plan.inst_head_blocks: wide_issue) — and goes
          // ahead after those; the rest of the 28,561 points and the spheres over them (0.1 ms for 25 chains, what the evaluator's
          // searches read) are beside it on the second stream (round 6)
          static const bool no_head = dev_env("ICP_WIDE_LOOP_HEAD_SPLIT") && std::atoi(dev_env("ICP_WIDE_LOOP_HEAD_SPLIT")) == 0;  // (A/B switch)
          const bool head_split = gr.side[0] && cap.any_split && cap.plan.inst_head_blocks > 0 && !no_head;
          if (head_split) {
            // (the rest BEHIND the head, not beside it: every wave of this launch walks the same 13 batches of basis rows, and with 1,788
            // of them resident the head's 36 finish when all do — 97 µs; alone they take a fraction of that)
            launch_wide_head_resident(S, cap.plan, gr.wide_dev.p, 1);
            HIP_OK(hipEventRecord(gr.ev_prop, S));
            HIP_OK(hipStreamWaitEvent(gr.side[0], gr.ev_prop, 0));
            launch_wide_head_resident(gr.side[0], cap.plan, gr.wide_dev.p, 2);
            HIP_OK(hipEventRecord(gr.ev_inst, gr.side[0]));
          } else launch_wide_head_resident(S, cap.plan, gr.wide_dev.p);
          launch_wide_main(S, cap.plan, gr.wide_dev.p);
          for (size_t p0 = 0; p0 < cap.sum_parts.size(); p0 += kWideMaxChains)
            launch_sum_partials_many(S, r, (int)std::min<size_t>(kWideMaxChains, cap.sum_parts.size() - p0), cap.sum_parts.data() + p0, cap.sum_splits.data() + p0);
          // what the summed partials feed does not depend on each other: the factorisations and tails (one-workgroup kernels), above rank 64
          // the proposed states' decompositions, the evaluator's searches and reductions; the decision waits for the tails, the hand-over of
          // an accepted state for the decompositions.
          // Round 6: the step's critical chain — main sequence, reduction to tridiagonal form, eigenpairs, back-transformation, hand-over —
          // stays on ONE queue: a dependency that crosses queues costs 20-30 µs between the end of one kernel and the start of the next
          // (profiles/r06_wide_loop25_step_timeline.txt before: 30 µs at the fork, 20 at the join).  The first launch of decompositions
          // goes on `S` itself, and what has slack — the evaluator's searches and the decision, 0.1 ms ahead of the reduction's end —
          // crosses instead: on the first decomposition stream of the older layout (kept below: ICP_WIDE_LOOP_CHAIN_MAIN=0, the A/B switch).
          static const bool chain_main = !(dev_env("ICP_WIDE_LOOP_CHAIN_MAIN") && std::atoi(dev_env("ICP_WIDE_LOOP_CHAIN_MAIN")) == 0) &&
                                         !(dev_env("ICP_WIDE_LOOP_EIG_SPLIT") && std::atoi(dev_env("ICP_WIDE_LOOP_EIG_SPLIT")) == 0);
          if (gr.side[0] && chain_main) {
            const hipStream_t F = gr.side[0], V = gr.side[1], E1 = gr.side[2];
            HIP_OK(hipEventRecord(gr.ev_sum, S));
            HIP_OK(hipStreamWaitEvent(F, gr.ev_sum, 0));
            HIP_OK(hipStreamWaitEvent(V, gr.ev_sum, 0));
            int used = 0;
            for (int q0 = 0; q0 < nq; q0 += tri_chunk, ++used) {  // part 1: the reductions (see below)
              if (used == 1) HIP_OK(hipStreamWaitEvent(E1, gr.ev_sum, 0));
              launch_posterior_eigen_tridiag_many((used & 1) ? E1 : S, r, std::min(tri_chunk, nq - q0), gr.rqs.data() + q0, gr.spec_parts.data() + q0, nullptr, 1);
            }
            const bool side_eig = used > 1;
            gr.n_eig_streams = 0; gr.eig_split = true;
            for (size_t p0 = 0; p0 < cap.factors.size(); p0 += fmax)
              launch_posterior_factor(F, r, (int)std::min(fmax, cap.factors.size() - p0), cap.factors.data() + p0);
            for (size_t t0 = 0; t0 < cap.tails.size(); t0 += 2 * kWideMaxChains)
              launch_transition_tails(F, r, (int)std::min<size_t>(2 * kWideMaxChains, cap.tails.size() - t0), cap.tails.data() + t0, lead.Ginv.p, kSigma2);
            HIP_OK(hipEventRecord(gr.ev_tails, F));
            if (head_split) HIP_OK(hipStreamWaitEvent(V, gr.ev_inst, 0));
            if (cap.any_split) launch_wide_eval(V, cap.plan, gr.wide_dev.p);
            HIP_OK(hipStreamWaitEvent(V, gr.ev_tails, 0));
            launch_mhw_decide(V, gr.B, r, gr.mh.p);
            HIP_OK(hipEventRecord(gr.ev_decide, V));
            HIP_OK(hipStreamWaitEvent(S, gr.ev_decide, 0));  // (long satisfied when the reduction ends)
            if (side_eig) HIP_OK(hipStreamWaitEvent(E1, gr.ev_decide, 0));
            used = 0;
            for (int q0 = 0; q0 < nq; q0 += tri_chunk, ++used)  // part 2, for the chains that moved
              launch_posterior_eigen_tridiag_many((used & 1) ? E1 : S, r, std::min(tri_chunk, nq - q0), gr.rqs.data() + q0, gr.spec_parts.data() + q0, gr.eig_skip.p + q0, 2);
            if (side_eig) {
              HIP_OK(hipEventRecord(gr.ev_eig[0], E1));
              HIP_OK(hipStreamWaitEvent(S, gr.ev_eig[0], 0));
            }
            launch_mhw_adopt(S, r, nq, gr.adopt.p, gr.eig_skip.p);
            continue;
          }
          // (the older layout, and ranks <= 64: factorisations and tails on a second stream, the decompositions on two more — kTriManyMax per
          // launch, the launches side by side —, the evaluator's sequence on `S`)
          hipStream_t S2 = S;
          if (gr.side[0]) {
            S2 = gr.side[0];
            HIP_OK(hipEventRecord(gr.ev_sum, S));
            HIP_OK(hipStreamWaitEvent(S2, gr.ev_sum, 0));
            // the proposed states' decompositions in two parts (round 6): the reduction to tridiagonal form — one workgroup per posterior,
            // 0.4 ms at rank 200 — starts here for every chain; the eigenpairs, back-transformation and refinement behind it are issued
            // BEHIND the decision (which falls while the reduction runs: tails and the evaluator's searches take 0.3 ms) and skip the
            // chains that did not move — half of them and more: the solve launch is the chip's throughput kernel (6,000 eigenpair waves
            // for 25 chains), and an accepted state's basis is all anybody will read
            static const bool no_split = dev_env("ICP_WIDE_LOOP_EIG_SPLIT") && std::atoi(dev_env("ICP_WIDE_LOOP_EIG_SPLIT")) == 0;  // (A/B switch)
            int used = 0;
            for (int q0 = 0; q0 < nq; q0 += tri_chunk, ++used) {
              const hipStream_t E = gr.side[1 + (used & 1)];
              if (used < 2) HIP_OK(hipStreamWaitEvent(E, gr.ev_sum, 0));
              launch_posterior_eigen_tridiag_many(E, r, std::min(tri_chunk, nq - q0), gr.rqs.data() + q0, gr.spec_parts.data() + q0, nullptr, no_split ? 0 : 1);
            }
            gr.n_eig_streams = std::min(used, 2);
            gr.eig_split = !no_split;
            if (no_split) for (int u = 0; u < gr.n_eig_streams; ++u) HIP_OK(hipEventRecord(gr.ev_eig[u], gr.side[1 + u]));
          }
          for (size_t p0 = 0; p0 < cap.factors.size(); p0 += fmax)
            launch_posterior_factor(S2, r, (int)std::min(fmax, cap.factors.size() - p0), cap.factors.data() + p0);
          for (size_t t0 = 0; t0 < cap.tails.size(); t0 += 2 * kWideMaxChains)
            launch_transition_tails(S2, r, (int)std::min<size_t>(2 * kWideMaxChains, cap.tails.size() - t0), cap.tails.data() + t0, lead.Ginv.p, kSigma2);
          if (S2 != S) HIP_OK(hipEventRecord(gr.ev_tails, S2));
          if (head_split) HIP_OK(hipStreamWaitEvent(S, gr.ev_inst, 0));
          if (cap.any_split) launch_wide_eval(S, cap.plan, gr.wide_dev.p);
          if (S2 != S) HIP_OK(hipStreamWaitEvent(S, gr.ev_tails, 0));
          launch_mhw_decide(S, gr.B, r, gr.mh.p);
          if (gr.n_eig_streams > 0 && gr.eig_split) {  // (ranks above 64) part 2 of the decompositions, for the chains that moved
            HIP_OK(hipEventRecord(gr.ev_decide, S));
            int used = 0;
            for (int q0 = 0; q0 < nq; q0 += tri_chunk, ++used) {
              const hipStream_t E = gr.side[1 + (used & 1)];
              if (used < 2) HIP_OK(hipStreamWaitEvent(E, gr.ev_decide, 0));
              launch_posterior_eigen_tridiag_many(E, r, std::min(tri_chunk, nq - q0), gr.rqs.data() + q0, gr.spec_parts.data() + q0, gr.eig_skip.p + q0, 2);
            }
            for (int u = 0; u < gr.n_eig_streams; ++u) HIP_OK(hipEventRecord(gr.ev_eig[u], gr.side[1 + u]));
          }
          for (int u = 0; u < gr.n_eig_streams; ++u) HIP_OK(hipStreamWaitEvent(S, gr.ev_eig[u], 0));
          launch_mhw_adopt(S, r, nq, gr.adopt.p, gr.eig_skip.p);
          if (jacobi) launch_posterior_eigen_resident(S, r, nq, gr.eig_rec.p, gr.eig_skip.p, root);
        }
    }
    // ---- results: nothing is handed out unless every chain came through
    std::vector<std::vector<MhChain>> hms(n_groups);
    int first_error = 0;
    const auto t_enq = std::chrono::steady_clock::now();
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = *groups[g];
      HIP_OK(hipStreamSynchronize(gr.st));
      if (g == n_groups - 1 && dev_env("ICP_WIDE_LOOP_TIMING"))
        std::fprintf(stderr, "[icp wide loop] %d chains x %d steps: enqueue %.1f ms, drain %.1f ms\n", n_chains, n_steps,
                     std::chrono::duration<double, std::milli>(t_enq - t_loop0).count(),
                     std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enq).count());
      hms[g].resize(gr.B);
      HIP_OK(hipMemcpy(hms[g].data(), gr.mh.p, sizeof(MhChain) * hms[g].size(), hipMemcpyDeviceToHost));
      for (const MhChain& m : hms[g])
        if (m.error != 0 && first_error == 0) first_error = m.error;
    }
    auto forget = [&](Chain& ch, int k, Group& gr) {  // whatever the entries hold now belongs to no state on record
      std::lock_guard<std::recursive_mutex> lk(ch.e->ctx->mu);
      WideItem& w = gr.ticket.items[k].W;
      for (int i = 0; i < n_props; ++i) {
        for (PosteriorEntry* en : {w.ec[i], w.ep[i]})
          if (en) { en->valid = false; en->eig_valid = false; en->eig_checked = false; en->eig_event_valid = false; en->done_value = 0; }
        ch.props[i]->warm_valid = false;
        ch.props[i]->spec_entry = nullptr;
      }
      if (w.s) w.s->valid = false;
    };
    if (first_error != 0) {
      for (int g = 0; g < n_groups; ++g)
        for (int k = 0; k < groups[g]->B; ++k) forget(chains[groups[g]->b0 + k], k, *groups[g]);
      fail(first_error == 3 ? ICP_ERR_NOT_SPD : first_error == 5 ? ICP_ERR_EMPTY : ICP_ERR_NOT_FINITE,
           first_error == 2 ? "on-device loop: a transition tail did not contract (step these chains through icp_chain_step_batched)"
           : first_error == 6 ? "on-device loop: posterior eigen-decomposition did not converge"
                              : "on-device loop: a chain stopped on a non-finite, empty or non-positive-definite result");
    }
    // the decompositions behind the LAST step's decisions have no decide kernel behind them: their status (pinned, written by the
    // decomposition itself) is looked at here
    for (int g = 0; g < n_groups; ++g)
      for (int k = 0; k < groups[g]->B; ++k) {
        Chain& ch = chains[groups[g]->b0 + k];
        for (int i = 0; i < n_props; ++i) {
          int st = ch.props[i]->h_eig[ch.cur[i]->status_off / 3];
          if (!jacobi)  // (decomposed in the proposed state's entry: the status word came over with the basis, in device memory)
            HIP_OK(hipMemcpy(&st, ch.props[i]->status.p + ch.cur[i]->status_off + 2, sizeof(int), hipMemcpyDeviceToHost));
          if (st != 0) {
            forget(ch, k, *groups[g]);
            fail(ICP_ERR_NOT_FINITE, "on-device loop: posterior eigen-decomposition did not converge");
          }
        }
      }
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = *groups[g];
      std::vector<double> hth((size_t)gr.B * P);
      HIP_OK(hipMemcpy(hth.data(), gr.theta.p, sizeof(double) * hth.size(), hipMemcpyDeviceToHost));
      for (int k = 0; k < gr.B; ++k) {
        const int b = gr.b0 + k;
        Chain& ch = chains[b];
        icp_ctx& c = *ch.e->ctx;
        const MhChain& m = hms[g][k];
        WideItem& w = gr.ticket.items[k].W;
        std::memcpy(theta[b], hth.data() + (size_t)k * P, sizeof(double) * P);
        log_value[b] = m.cur_p;
        if (accepted) accepted[b] = m.accepted;
        if (records && records[b] && n_steps > 0)
          HIP_OK(hipMemcpy(records[b], gr.rec.p + (size_t)k * n_steps * (4 + P), sizeof(double) * (size_t)n_steps * (4 + P), hipMemcpyDeviceToHost));
        // the contexts' own bookkeeping: the current state's entries are on record again, decomposed; nothing else of the run is
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        for (int i = 0; i < n_props; ++i) {
          icp_proposal* p = ch.props[i];
          PosteriorEntry& cur = *w.ec[i];
          cur.valid = true; cur.eig_valid = true; cur.eig_checked = true; cur.eig_event_valid = false; cur.done_value = 0;
          cur.theta.assign(theta[b], theta[b] + P);
          cur.stamp = ++p->clock;
          p->h_eig[cur.status_off / 3] = 0;
          p->warm_ptr = cur.V.p;
          p->warm_valid = true;
          PosteriorEntry& pr = *w.ep[i];
          pr.valid = false; pr.eig_valid = false; pr.eig_checked = false; pr.eig_event_valid = false; pr.done_value = 0;
          if (jacobi) p->eig_seq = m.eig_seq[i];
          p->spec_entry = nullptr;
        }
        w.s->valid = false;
        ch.e->last_prop.clear();
        c.paths.n[3] += n_steps; g_step_paths.n[3] += n_steps;
      }
    }
  });
  release();
  return rc;
}
}  // namespace

extern "C" {
// --------------------------------------------------------------------- the whole MH loop on the device (MhChain, kernels_step.hip)
int icp_chains_run_on_device(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props_in,
                             const icp_mh_mixture* mix, const uint64_t* seeds, const int64_t* first_step, double* const* theta,
                             double* log_value, int32_t n_steps, double* const* records, int64_t* accepted) {
  struct Chain {
    icp_evaluator* e = nullptr;
    icp_proposal* props[2] = {nullptr, nullptr};
    PosteriorEntry* set[2][2] = {{nullptr, nullptr}, {nullptr, nullptr}};  // [sel][proposal]
    StateSlot* slot = nullptr;
    std::unique_lock<std::recursive_mutex> lk;
    bool busy = false;
  };
  // chains whose step is the wide one (an open target, the Hausdorff evaluator, a rank above 64): the loop of their own
  if (n_chains >= 1 && evaluators && evaluators[0] && props_in && props_in[0] && n_props >= 1 && n_props <= 2) {
    bool wide = false, ok = true;
    for (int i = 0; i < n_props; ++i) ok = ok && props_in[i] && props_in[i]->ctx == evaluators[0]->ctx;
    if (ok) {
      std::lock_guard<std::recursive_mutex> lk(evaluators[0]->ctx->mu);
      // (… or the five merged launches at a rank whose decomposition is not the one-workgroup Jacobi iteration — ranks 65..116,
      // apps/femur/StdIcpVsChainICPrandomInitComparisonAll.scala at rank 101: the wide step's kernels compute the same step)
      wide = (!step_pipeline_covers(evaluators[0], n_props, props_in) || !eigen_speculation_supported(evaluators[0]->ctx->r)) &&
             wide_pipeline_covers(evaluators[0], n_props, props_in);
    }
    if (wide)
      return wide_chains_run_on_device(n_chains, evaluators, n_props, props_in, mix, seeds, first_step, theta, log_value, n_steps, records, accepted);
  }
  std::vector<Chain> chains;
  auto release = [&]() {
    for (auto& ch : chains) {
      if (!ch.e) continue;
      icp_ctx& c = *ch.e->ctx;
      if (!ch.lk.owns_lock()) ch.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      for (int sel = 0; sel < 2; ++sel)
        for (int i = 0; i < 2; ++i)
          if (ch.set[sel][i]) ch.set[sel][i]->reserved = false;
      if (ch.slot) ch.slot->reserved = false;
      if (ch.busy) c.batch_busy = false;
      ch.lk.unlock();
    }
  };
  int rc = guard([&] {
    require(n_chains >= 1 && evaluators && props_in && mix && seeds && first_step && theta && log_value && n_steps >= 0, "null argument");
    require(n_props >= 1 && n_props <= 2, "the on-device loop takes one or two ICP proposals per chain");
    require(mix->struct_size == sizeof(icp_mh_mixture), "icp_mh_mixture::struct_size does not match this library's header");
    require(mix->w_icp > 0.0 && mix->w_rw >= 0.0 && mix->rw_sigma > 0.0 && mix->w_pose >= 0.0, "bad mixture");
    if (mix->w_pose > 0.0)
      for (int a = 0; a < 3; ++a) require(mix->pose_rot_sigma[a] > 0.0 && mix->pose_trans_sigma[a] > 0.0, "pose walk sigmas must be positive");
    chains.resize(n_chains);
    icp_ctx& lead = *evaluators[0]->ctx;
    const int r = lead.r, P = 10 + r;
    require(eigen_speculation_supported(r), "the on-device loop covers ranks 3..64");
    // ---- claim the chains' contexts, fix their posterior entries and state slot
    for (int b = 0; b < n_chains; ++b) {
      Chain& ch = chains[b];
      require(evaluators[b] && theta[b], "null argument");
      icp_ctx& c = *evaluators[b]->ctx;
      require(c.device == lead.device && c.r == r, "chains of one run share a device and a rank");
      for (int a = 0; a < b; ++a) require(chains[a].e->ctx != &c, "every chain needs a context of its own");
      ch.e = evaluators[b];
      for (int i = 0; i < n_props; ++i) {
        ch.props[i] = props_in[(size_t)b * n_props + i];
        require(ch.props[i] && ch.props[i]->ctx == &c, "proposal belongs to another context");
        require(ch.props[i]->sampler == ch.props[0]->sampler && ch.props[i]->sampler == chains[0].props[0]->sampler, "one sampler per run");
      }
      check_theta_finite(&c, theta[b]);
      ch.lk = std::unique_lock<std::recursive_mutex>(c.mu);
      if (c.batch_busy) fail(ICP_ERR_BUSY, "a chain's context already belongs to a batch in flight");
      require(step_pipeline_covers(ch.e, n_props, ch.props), "configuration not covered by the merged launches");
      // (the device makes the proposed pose's matrix with the library's own convention: open to a host whose supplied matrices have
      // all agreed with it — icp_ctx_set_rotation's check —, closed to one whose convention is another)
      if (mix->w_pose > 0.0)
        require(c.rotations_mismatched == 0, "pose walks on the device: a caller-supplied rotation matrix disagreed with the library's Rz·Ry·Rx (icp_ctx_rotation_convention)");
      Bound _b(&c);
      if (ch.e->front.valid) release_front(ch.e->front);
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = ch.props[i];
        p->resolve_speculation(theta[b]);
        PosteriorEntry& cur = p->posterior(theta[b], false);  // the current state's posterior and its basis, the ordinary way
        p->ensure_eigen(cur);
        cur.reserved = true;
        ch.set[0][i] = &cur;
        PosteriorEntry& other = p->fresh_entry();
        other.reserved = true;
        ch.set[1][i] = &other;
      }
      HIP_OK(hipStreamSynchronize(c.stream));
      c.front_stream.sync();
      sync_eigen(c);
      for (int i = 0; i < n_props; ++i) sync_proposal_status(ch.props[i]);
      HIP_OK(hipStreamSynchronize(c.stream));
      for (int i = 0; i < n_props; ++i) {
        ch.props[i]->check_status(*ch.set[0][i]);
        if (ch.props[i]->h_eig[ch.set[0][i]->status_off / 3] != 0) fail(ICP_ERR_NOT_FINITE, "posterior eigen-decomposition did not converge");
      }
      StateSlot& s = c.fresh_state();
      s.reserved = true;
      s.pose = c.pose_of(theta[b]);
      ch.slot = &s;
      c.batch_busy = true;
      ch.busy = true;
      ch.lk.unlock();
    }
    // ---- groups: each its own stream, everything of a group in order on it; the groups overlap each other's launches
    // (three since the token moved forward, §5.1c: 64 chains 202k it/s in two groups, 211k in three; 128 chains 251k / 257k; 32 chains 130k /
    // 135k; 24 chains 104k / 108k.  Four — a fourth stream made with every context — measured 218k / 266k / 137k / 109k, but the extra
    // stream shifts every context's streams over the runtime's hardware queues, and the host-stepped lockstep path's decompositions, which
    // wait on the device for launches of other streams, then ran into their time-outs: 15 of them in a 50-chain test.  Not adopted.)
    const int n_groups = n_chains >= 16 ? 3 : 1;
    struct Group {
      int b0 = 0, B = 0;
      hipStream_t st = nullptr;
      int grid[5] = {0, 0, 0, 0, 0};
      DBuf<StepBeginArgs> begin_alt, begin_live;
      DBuf<StepSearchArgs> search_alt, search_live;
      bool filter_prepared = true;  // (step_filter_prepared of every captured step)
      bool reg_folded = false;      // (the captured regressions fold their leaves: regression_fold)
      DBuf<StepRegressionArgs> regression_alt, regression_live;
      DBuf<StepFinishArgs> finish_alt, finish_live;
      DBuf<MhChain> mh;
      DBuf<EigenProblem> eig_live;
      DBuf<int> eig_skip;
      DBuf<double> normals[2], theta, rec;
      DBuf<double> res;   // per chain 32 doubles: [0..7] launch 4's reductions, [8..11] the tails fwd_i / bwd_i — in DEVICE memory
      DBuf<int> stat;     // per chain 16 ints: [0..3] the tails' status, [8..9] the factorisations' (the decide kernel reads them: pinned
                          // host memory, where the host-stepped paths want them, would cost it a bus round trip per number)
      double* h_normals[2] = {nullptr, nullptr};
      hipEvent_t ev_copy[2] = {nullptr, nullptr};
      hipEvent_t ev_big = nullptr;  // recorded behind launch 4: the next group's chip-wide launches may start
    };
    std::vector<Group> groups(n_groups);
    constexpr int kChunk = 64;  // steps per block of standard normals
    const int root = chains[0].props[0]->sampler == ICP_SAMPLER_CHOLESKY_ROOT;
    struct GroupGuard {
      std::vector<Group>& g;
      ~GroupGuard() {
        for (auto& gr : g) {
          for (int k = 0; k < 2; ++k) {
            if (gr.h_normals[k]) pinned_free(gr.h_normals[k]);
            if (gr.ev_copy[k]) (void)hipEventDestroy(gr.ev_copy[k]);
          }
          if (gr.ev_big) (void)hipEventDestroy(gr.ev_big);
        }
      }
    } group_guard{groups};
    lead.bind();
    std::vector<double> zero_key(P, 0.0);
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      gr.b0 = (int)((long long)g * n_chains / n_groups);
      gr.B = (int)((long long)(g + 1) * n_chains / n_groups) - gr.b0;
      // the first chain's three streams: created one after the other with the context, they sit on different hardware queues and run
      // beside each other (streams of DIFFERENT contexts, or streams made later, may share a queue: the runtime multiplexes streams
      // onto a handful of them, and two groups on one queue alternate in ~55 µs slices — every kernel of the step then "takes" a
      // multiple of that: eight streams made for the purpose on first use ran two groups at 116k it/s instead of 202k)
      gr.st = g == 0 ? lead.stream : g == 1 ? lead.front_stream.get() : lead.eig_stream.get();
      const int B = gr.B;
      struct FoldScope { FoldScope(int n) { tl_regression_posteriors = n; } ~FoldScope() { tl_regression_posteriors = 1; } } fold_scope(std::max(1, B * n_props));
    struct SearchHintScope2 { SearchHintScope2(int n) { search_chains_hint(n); } ~SearchHintScope2() { search_chains_hint(1); } } search_hint_scope2(B);
      gr.begin_alt.alloc(2 * B); gr.begin_live.alloc(B);
      gr.search_alt.alloc(2 * B); gr.search_live.alloc(B);
      gr.regression_alt.alloc(2 * B); gr.regression_live.alloc(B);
      gr.finish_alt.alloc(2 * B); gr.finish_live.alloc(B);
      gr.mh.alloc(B);
      gr.eig_live.alloc((size_t)B * n_props);
      gr.eig_skip.alloc((size_t)B * n_props);
      gr.theta.alloc((size_t)B * P);
      gr.res.alloc((size_t)B * 32); gr.res.fill_bytes(0);
      gr.stat.alloc((size_t)B * 16); gr.stat.fill_bytes(0);
      HIP_OK(hipEventCreateWithFlags(&gr.ev_big, hipEventDisableTiming));
      gr.rec.alloc(records ? (size_t)B * std::max(n_steps, 1) * (4 + P) : 1);
      for (int k = 0; k < 2; ++k) {
        gr.normals[k].alloc((size_t)B * kChunk * r);
        pinned_alloc((void**)&gr.h_normals[k], sizeof(double) * (size_t)B * kChunk * r);
        HIP_OK(hipEventCreateWithFlags(&gr.ev_copy[k], hipEventDisableTiming));
      }
      std::vector<StepBeginArgs> hb(2 * B);
      std::vector<StepSearchArgs> hs(2 * B);
      std::vector<StepRegressionArgs> hr(2 * B);
      std::vector<StepFinishArgs> hf(2 * B);
      std::vector<MhChain> hm(B);
      std::vector<int> hskip((size_t)B * n_props, 1);
      std::vector<double> hth((size_t)B * P);
      for (int k = 0; k < B; ++k) {
        Chain& ch = chains[gr.b0 + k];
        icp_ctx& c = *ch.e->ctx;
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        Bound _b(&c, true, true);
        MhChain& m = hm[k];
        std::memset(&m, 0, sizeof(m));
        for (int sel = 0; sel < 2; ++sel) {
          // the step's launches with the current state in set `sel` and the proposed one in the other, captured
          StepCapture cap;
          std::memset(cap.grid, 0, sizeof(cap.grid));
          StepFront F;
          F.n_props = n_props; F.generator = -1; F.parity = 0; F.stream = c.stream; F.s = ch.slot;
          for (int i = 0; i < n_props; ++i) { F.props[i] = ch.props[i]; F.ec[i] = ch.set[sel][i]; F.ep[i] = ch.set[1 - sel][i]; }
          {
            struct CaptureScope { CaptureScope(StepCapture* cp) { step_capture(cp); } ~CaptureScope() { step_capture(nullptr); } } scope(&cap);
            front_launches(ch.e, n_props, ch.props, -1, zero_key.data(), F, true, false);
            StepFinishArgs f{};
            f.n = n_props; f.r = r; f.Ginv = c.Ginv.p; f.sigma2 = kSigma2;
            for (int i = 0; i < n_props; ++i) {
              icp_proposal* p = ch.props[i];
              f.Mpart[i] = F.mpart[i]; f.splits[i] = F.splits[i];
              f.M[i] = F.ep[i]->M.p; f.alpha[i] = F.ep[i]->alpha.p;
              f.status[i] = p->status.p + F.ep[i]->status_off;
              f.host_status[i] = gr.stat.p + (size_t)k * 16 + 8 + i;
              f.fwd[i] = TransitionTailIO{F.ec[i]->alpha.p, F.ec[i]->M.p, F.ec[i]->coeffs.p, F.ep[i]->coeffs.p, p->prm.step_length,
                                          gr.res.p + (size_t)k * 32 + 8 + 2 * i, gr.stat.p + (size_t)k * 16 + 2 * i};
              f.bwd[i] = TransitionTailIO{F.ep[i]->alpha.p, F.ep[i]->M.p, F.ep[i]->coeffs.p, F.ec[i]->coeffs.p, p->prm.step_length,
                                          gr.res.p + (size_t)k * 32 + 9 + 2 * i, gr.stat.p + (size_t)k * 16 + 2 * i + 1};
            }
            f.done_counter = c.d_done.p; f.host_flag = c.h_flag; f.seq = 0;
            f.ready_flag = nullptr;
            launch_step_finish(c.stream, f);  // (captured; finalised by the launcher)
          }
          cap.begin.wait_flag = nullptr; cap.begin.wait2_flag = nullptr; cap.begin.wait_ticks = nullptr; cap.begin.hold_regs = 0;
          cap.regression.red_out = gr.res.p + (size_t)k * 32;  // (device memory instead of the context's pinned area)
          {  // (launch 1's matvec layout: set by enqueue_front only when it knows the generator)
            int t = 0;
            while (t < 6 && (r << (t + 1)) <= 256 && (r >> (t + 1)) >= 8) ++t;
            cap.begin.tpr_log2 = t;
          }
          hb[(size_t)sel * B + k] = cap.begin; hs[(size_t)sel * B + k] = cap.search; hr[(size_t)sel * B + k] = cap.regression;
          hf[(size_t)sel * B + k] = cap.finish;
          for (int q = 0; q < 5; ++q) gr.grid[q] = std::max(gr.grid[q], cap.grid[q]);
          gr.filter_prepared = gr.filter_prepared && step_filter_prepared(cap.search);
          for (int i = 0; i < cap.regression.n && i < 2; ++i)  // (any folded record of any chain: the folded kernel takes both kinds)
            gr.reg_folded = gr.reg_folded || cap.regression.fold[i] > 1;
          m.begin_alt[sel] = gr.begin_alt.p + (size_t)sel * B + k;
          m.search_alt[sel] = gr.search_alt.p + (size_t)sel * B + k;
          m.regression_alt[sel] = gr.regression_alt.p + (size_t)sel * B + k;
          m.finish_alt[sel] = gr.finish_alt.p + (size_t)sel * B + k;
          for (int i = 0; i < n_props; ++i) {
            icp_proposal* p = ch.props[i];
            PosteriorEntry& cur = *ch.set[sel][i];
            m.prop_alt[sel][i] = ProposeIn{cur.alpha.p, cur.V.p, cur.S.p, c.inv_sqrt_lambda.p, c.P.p, cur.coeffs.p, nullptr, kSigma2,
                                           p->prm.step_length, root};
            // the decomposition of set `sel`'s posterior (an accepted state arrives there), warm-started from the other set's basis
            PosteriorEntry& other = *ch.set[1 - sel][i];
            EigenRequest rq{cur.M.p, root ? nullptr : other.V.p, cur.V.p, cur.Vt.p, cur.S.p, p->work.p, p->status.p + cur.status_off + 2, nullptr,
                            p->h_eig + cur.status_off / 3, p->eig_words.p + cur.status_off / 3, 0, c.sqrt_lambda.p};
            rq.root = root != 0;
            m.eig_alt[sel][i] = eigen_problem_of(r, rq);
          }
        }
        m.begin_live = gr.begin_live.p + k; m.search_live = gr.search_live.p + k;
        m.regression_live = gr.regression_live.p + k; m.finish_live = gr.finish_live.p + k;
        m.eig_live = gr.eig_live.p + (size_t)k * n_props;
        m.eig_skip = gr.eig_skip.p + (size_t)k * n_props;
        m.pw_id_mask = 2046;
        m.seed = seeds[gr.b0 + k];
        m.r = r; m.n_icp = n_props;
        {  // MixtureProposal weights, normalised as the harness normalises them (host/icp_host.hpp: pick_component)
          double ws = 0.0;
          for (int i = 0; i < n_props; ++i) ws += mix->icp_weight[i];
          for (int i = 0; i < n_props; ++i) m.icp_w[i] = mix->icp_weight[i] / ws;
          double raw[3];
          int no = 0;
          if (mix->w_pose > 0.0) { m.outer_kind[no] = 0; raw[no++] = mix->w_pose; }  // (BfmFittingPartial.scala:70: pose, ICP, shape walk)
          m.outer_kind[no] = 1; raw[no++] = mix->w_icp;
          if (mix->w_rw > 0.0) { m.outer_kind[no] = 2; raw[no++] = mix->w_rw; }
          double wsum = 0.0;
          for (int o = 0; o < no; ++o) wsum += raw[o];
          for (int o = 0; o < no; ++o) m.outer_w[o] = raw[o] / wsum;
          m.n_outer = no;
        }
        if (mix->w_pose > 0.0) {  // MixedProposalDistributions.scala:29-39 / host/icp_host.cpp: mixed_random_pose_proposal
          static const int param_index[6] = {6, 5, 4, 1, 2, 3};  // yaw = rotation._3, pitch = _2, roll = _1 (PoseProposals.scala:39-41); x, y, z
          m.n_pose = 6;
          double wsum = 0.0;
          for (int a = 0; a < 6; ++a) wsum += 0.5;
          for (int a = 0; a < 6; ++a) {
            const double sd = a < 3 ? mix->pose_rot_sigma[a] : mix->pose_trans_sigma[a - 3];
            m.pose_index[a] = param_index[a];
            m.pose_w[a] = 0.5 / wsum;
            m.pose_sigma[a] = sd;
            m.pose_logc[a] = std::log(std::sqrt(2.0 * M_PI)) + std::log(sd);  // breeze Gaussian(0, σ).logPdf's normaliser
          }
          m.front_every_step = 1;
        }
        m.rw_sigma = mix->rw_sigma;
        m.rw_logc = 0.5 * (r * std::log(2.0 * M_PI) + r * std::log(mix->rw_sigma * mix->rw_sigma));
        m.prior_c = 0.5 * r * std::log(2.0 * M_PI);
        const icp_evaluator_params& ep = ch.e->prm;
        m.eval_kind = ep.kind == ICP_EVAL_INDEPENDENT_POINT_DISTANCE ? 0 : 2;
        m.eval_mode = ep.mode;
        m.gauss_mean = ep.gauss_mean; m.gauss_sigma = ep.gauss_sigma;
        m.gauss_logn = std::log(std::sqrt(2.0 * M_PI)) + std::log(ep.gauss_sigma);
        m.exp_rate = ep.exp_rate; m.exp_lograte = std::log(ep.exp_rate);
        m.coeff_prop = ch.slot->coeffs.p;
        m.red = gr.res.p + (size_t)k * 32;
        m.tails = gr.res.p + (size_t)k * 32 + 8;
        m.tail_status = gr.stat.p + (size_t)k * 16;
        m.chol_status = gr.stat.p + (size_t)k * 16 + 8;
        m.normals = nullptr; m.normals_first = 0; m.normals_rows = 0;
        m.records = records && records[gr.b0 + k] ? gr.rec.p + (size_t)k * n_steps * (4 + P) : nullptr;
        m.rec_first = first_step[gr.b0 + k];
        m.theta = gr.theta.p + (size_t)k * P;
        m.cur_p = log_value[gr.b0 + k];
        m.step = first_step[gr.b0 + k];
        m.accepted = 0; m.cur_sel = 0; m.gen = -1; m.leaf = -1; m.error = 0;
        for (int i = 0; i < n_props; ++i) m.eig_seq[i] = ch.props[i]->eig_seq;
        std::memcpy(hth.data() + (size_t)k * P, theta[gr.b0 + k], sizeof(double) * P);
        for (int i = 0; i < 16; ++i) { c.h_res[i] = 0.0; c.h_status[i] = 0; }
      }
      HIP_OK(hipMemcpy(gr.begin_alt.p, hb.data(), sizeof(StepBeginArgs) * hb.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.search_alt.p, hs.data(), sizeof(StepSearchArgs) * hs.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.regression_alt.p, hr.data(), sizeof(StepRegressionArgs) * hr.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.finish_alt.p, hf.data(), sizeof(StepFinishArgs) * hf.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.eig_skip.p, hskip.data(), sizeof(int) * hskip.size(), hipMemcpyHostToDevice));
      HIP_OK(hipMemcpy(gr.theta.p, hth.data(), sizeof(double) * hth.size(), hipMemcpyHostToDevice));
      // (normals: the two buffers are addressed through MhChain::normals / normals_first, re-pointed per block of steps below)
      HIP_OK(hipMemcpy(gr.mh.p, hm.data(), sizeof(MhChain) * hm.size(), hipMemcpyHostToDevice));
      HIP_OK(hipStreamSynchronize(nullptr));  // (as DBuf::upload: the copies have reached the device before a non-blocking stream's launch reads them)
    }
    // ---- the loop: per block of kChunk steps the chains' standard normals (the harness' own expression, drawn here on the host
    // while the device works on the block before), then per step and group eight launches, nothing waited for
    auto draw_block = [&](Group& gr, int blk, int buf) {
      const int s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      double* out = gr.h_normals[buf];
      for (int k = 0; k < gr.B; ++k) {
        const uint64_t seed = seeds[gr.b0 + k];
        const uint64_t f0 = (uint64_t)first_step[gr.b0 + k] + (uint64_t)s0;
        for (int s_ = 0; s_ < ns; ++s_)
          for (int j = 0; j < r; ++j) out[((size_t)k * kChunk + s_) * r + j] = harness_normal(seed, f0 + (uint64_t)s_, (uint64_t)j);
      }
    };
    const int n_blocks = (n_steps + kChunk - 1) / kChunk;
    // (per-kernel event timing, if the first chain's context is being profiled: icp_ctx_profile_start — its event pool, both streams)
    struct ProfBind { ProfBind(icp_ctx& c) { g_prof = c.profiling ? &c.prof : nullptr; } ~ProfBind() { g_prof = nullptr; } } prof_bind(lead);
    for (int blk = 0; blk < n_blocks; ++blk) {
      const int buf = blk & 1, s0 = blk * kChunk, ns = std::min(kChunk, n_steps - s0);
      for (auto& gr : groups) {
        HIP_OK(hipEventSynchronize(gr.ev_copy[buf]));  // (the staging buffer's previous upload has left it)
        draw_block(gr, blk, buf);
        HIP_OK(hipMemcpyAsync(gr.normals[buf].p, gr.h_normals[buf], sizeof(double) * (size_t)gr.B * kChunk * r, hipMemcpyHostToDevice, gr.st));
        HIP_OK(hipEventRecord(gr.ev_copy[buf], gr.st));
        launch_mh_set_normals(gr.st, gr.B, gr.mh.p, gr.normals[buf].p, kChunk * r, s0, ns);
      }
      // The chip-wide launches of two groups side by side slow each other down more than the overlap gains; what
      // should run beside a group's chip-wide launches is the OTHER group's small ones (launch 5 on four CUs per chain, the
      // decide kernel, the decompositions on three CUs each).  So the launches pass a token from group to group: a group's first
      // launch waits for an event of the previous group's.  Which one: behind launch 4 while the filter launch held five workgroups per
      // CU; with eight (§5.1c) behind launch 2 — the other group's begin and filter beside this group's resolve and regression, two
      // chains of latencies that leave the CUs room — measures best (two groups of 32 chains: 202.0k it/s against 197.3k behind launch
      // 4, 199.0k without a token; four groups of 16, on a build with a fourth stream: 216.2k against 215.0k behind launch 1, 206.3k
      // without, 199.6k behind launch 3, 167.9k behind launch 4); behind launch 1 with 64 per group (two groups of 64: 249.5k against 244.7k / 241.5k).
      for (int s_ = 0; s_ < ns; ++s_)
        for (int g = 0; g < n_groups; ++g) {
          Group& gr = groups[g];
          if (n_groups > 1) {
            Group& prev = groups[(g + n_groups - 1) % n_groups];
            if (blk > 0 || s_ > 0 || g > 0) HIP_OK(hipStreamWaitEvent(gr.st, prev.ev_big, 0));
          }
          // (later steps of a block: prepared by the decide kernel of the step before — except with pose walks, whose proposed pose is
          // made by the front kernel)
          if (s_ == 0 || mix->w_pose > 0.0) launch_mh_front(gr.st, gr.B, gr.mh.p);
          const int token_at = gr.B >= 64 ? 1 : 2;
          {
            int ga[5] = {0, 0, 0, 0, 0}, gb[5] = {0, 0, 0, 0, 0};
            for (int q = 0; q < 4; ++q) (q < token_at ? ga : gb)[q] = gr.grid[q];
            if (token_at > 0)
              launch_step_batch_resident(gr.st, gr.B, ga, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p, gr.filter_prepared, gr.reg_folded);
            if (n_groups > 1) HIP_OK(hipEventRecord(gr.ev_big, gr.st));
            if (token_at < 4)
              launch_step_batch_resident(gr.st, gr.B, gb, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p, gr.filter_prepared, gr.reg_folded);
          }
          int g5[5] = {0, 0, 0, 0, gr.grid[4]};
          launch_step_batch_resident(gr.st, gr.B, g5, r, gr.begin_live.p, gr.search_live.p, gr.regression_live.p, gr.finish_live.p);
          launch_mh_decide(gr.st, gr.B, gr.mh.p);
          launch_posterior_eigen_resident(gr.st, r, gr.B * n_props, gr.eig_live.p, gr.eig_skip.p, root);
        }
    }
    // ---- results: nothing is handed out unless every chain came through
    std::vector<std::vector<MhChain>> hms(n_groups);
    int first_error = 0;
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      HIP_OK(hipStreamSynchronize(gr.st));
      hms[g].resize(gr.B);
      HIP_OK(hipMemcpy(hms[g].data(), gr.mh.p, sizeof(MhChain) * hms[g].size(), hipMemcpyDeviceToHost));
      for (const MhChain& m : hms[g])
        if (m.error != 0 && first_error == 0) first_error = m.error;
    }
    if (first_error != 0) {
      for (auto& ch : chains) {  // whatever the sets hold now belongs to no state on record
        std::lock_guard<std::recursive_mutex> lk(ch.e->ctx->mu);
        for (int i = 0; i < n_props; ++i) {
          for (int sel = 0; sel < 2; ++sel) { ch.set[sel][i]->valid = false; ch.set[sel][i]->eig_valid = false; ch.set[sel][i]->eig_checked = false; }
          ch.props[i]->warm_valid = false;
          ch.props[i]->spec_entry = nullptr;
        }
        ch.slot->valid = false;
      }
      fail(first_error == 3 ? ICP_ERR_NOT_SPD : first_error == 5 ? ICP_ERR_EMPTY : ICP_ERR_NOT_FINITE,
           first_error == 2 ? "on-device loop: a transition tail did not contract (step these chains through icp_chain_step_batched)"
           : first_error == 6 ? "on-device loop: posterior eigen-decomposition did not converge"
                              : "on-device loop: a chain stopped on a non-finite, empty or non-positive-definite result");
    }
    // the decompositions behind the LAST step's decisions have no decide kernel behind them: their status (pinned, written by the
    // decomposition itself) is looked at here, before the sets are booked as decomposed and checked
    for (int g = 0; g < n_groups; ++g)
      for (int k = 0; k < groups[g].B; ++k) {
        Chain& ch = chains[groups[g].b0 + k];
        const MhChain& m = hms[g][k];
        for (int i = 0; i < n_props; ++i) {
          const int st = ch.props[i]->h_eig[ch.set[m.cur_sel][i]->status_off / 3];
          if (st != 0) {
            std::lock_guard<std::recursive_mutex> lk(ch.e->ctx->mu);
            for (int sel = 0; sel < 2; ++sel) { ch.set[sel][i]->valid = false; ch.set[sel][i]->eig_valid = false; ch.set[sel][i]->eig_checked = false; }
            ch.props[i]->warm_valid = false;
            ch.props[i]->spec_entry = nullptr;
            fail(ICP_ERR_NOT_FINITE, "on-device loop: posterior eigen-decomposition did not converge");
          }
        }
      }
    for (int g = 0; g < n_groups; ++g) {
      Group& gr = groups[g];
      std::vector<double> hth((size_t)gr.B * P);
      HIP_OK(hipMemcpy(hth.data(), gr.theta.p, sizeof(double) * hth.size(), hipMemcpyDeviceToHost));
      for (int k = 0; k < gr.B; ++k) {
        const int b = gr.b0 + k;
        Chain& ch = chains[b];
        icp_ctx& c = *ch.e->ctx;
        const MhChain& m = hms[g][k];
        std::memcpy(theta[b], hth.data() + (size_t)k * P, sizeof(double) * P);
        log_value[b] = m.cur_p;
        if (accepted) accepted[b] = m.accepted;
        if (records && records[b] && n_steps > 0)
          HIP_OK(hipMemcpy(records[b], gr.rec.p + (size_t)k * n_steps * (4 + P), sizeof(double) * (size_t)n_steps * (4 + P), hipMemcpyDeviceToHost));
        // the contexts' own bookkeeping: the set that holds the current state is on record again, decomposed
        std::lock_guard<std::recursive_mutex> lk(c.mu);
        for (int i = 0; i < n_props; ++i) {
          icp_proposal* p = ch.props[i];
          for (int sel = 0; sel < 2; ++sel) {
            PosteriorEntry& en = *ch.set[sel][i];
            const bool cur = sel == m.cur_sel;
            en.valid = cur; en.eig_valid = cur; en.eig_checked = cur; en.eig_event_valid = false; en.done_value = 0;
            if (cur) {
              en.theta.assign(theta[b], theta[b] + P);
              en.stamp = ++p->clock;
              p->h_eig[en.status_off / 3] = 0;
              p->warm_ptr = en.V.p;
              p->warm_valid = true;
            }
          }
          p->eig_seq = m.eig_seq[i];
          p->spec_entry = nullptr;
        }
        ch.slot->valid = false;
        ch.e->last_prop.clear();
        c.paths.n[3] += n_steps; g_step_paths.n[3] += n_steps;
      }
    }
  });
  release();
  return rc;
}

int icp_proposal_basis_state(icp_proposal* p, const double* theta) {
  if (!p || !theta) return ICP_ERR_INVALID_ARG;
  icp_ctx& c = *p->ctx;
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  if (c.batch_busy) return 1;  // (its context is part of a batch in flight: whatever it is doing, it is not ready)
  PosteriorEntry* e = p->find_entry(theta);
  if (!e) return 0;
  if (!e->eig_valid) return 0;
  return *(volatile int*)(p->h_eig + e->status_off / 3) == -1 ? 1 : 2;
}

int icp_chain_step_path(icp_evaluator* e, int32_t n_props, icp_proposal* const* props) {
  if (!e || n_props < 0 || (n_props > 0 && !props)) return ICP_ERR_INVALID_ARG;
  for (int i = 0; i < n_props; ++i)
    if (!props[i] || props[i]->ctx != e->ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(e->ctx->mu);
  if (step_pipeline_covers(e, n_props, props)) return 0;
  if (n_props >= 1 && n_props <= 2 && wide_pipeline_covers(e, n_props, props)) return 1;
  return 2;
}

int icp_chain_step_batched(int32_t n_chains, icp_evaluator* const* evaluators, int32_t n_props, icp_proposal* const* props,
                           const int32_t* generator, const double* const* theta_cur, const double* const* z,
                           double* const* theta_prop, double* log_value_prop, double* fwd, double* bwd, int32_t* status) {
  icp_step_ticket* tk = nullptr;
  const int rc = icp_chain_step_batched_issue(n_chains, evaluators, n_props, props, generator, theta_cur, z, theta_prop, log_value_prop, fwd,
                                              bwd, status, nullptr, &tk);
  if (rc != ICP_OK) return rc;
  return icp_chain_step_batched_collect(tk);
}



}  // extern "C"
