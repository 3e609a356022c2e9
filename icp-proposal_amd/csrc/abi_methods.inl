// abi_methods.inl — part of icp_abi.hip (one translation unit; included there, in order).
// C ABI: per-method entry points — proposals, evaluators, deterministic fit, variability maps, metrics, icp_chain_eval_step
namespace {
// ---- icp_chain_bind: a whole step submitted on behalf of ONE per-method call (include/icp_proposal.h).  Scalismo's
// MetropolisHastings.next (built at api/sampling/SamplingRegistration.scala:54) asks for a step's numbers one method at
// a time — propose, logValue(proposal), and through MixtureProposal.logTransitionProbability (App. B2) every ICP proposal's density
// both ways: six device round trips.  Bound, the first of these calls submits what icp_chain_step submits and parks the rest.
thread_local int tl_bound_depth = 0;  // > 0: inside such a step — the entry points it calls itself compute as if unbound
struct BoundStepScope {
  BoundStepScope() { ++tl_bound_depth; }
  ~BoundStepScope() { --tl_bound_depth; }
};

void unbind_chain(icp_evaluator* e) {
  for (int i = 0; i < e->bind.n; ++i)
    if (e->bind.props[i]) { e->bind.props[i]->bound_eval = nullptr; e->bind.props[i]->bound_index = -1; }
  e->bind = icp_evaluator::ChainBinding{};
}

void park_bound_step(icp_evaluator* e, const double* cur, const double* prop, const double* fwd, const double* bwd) {
  const size_t P = 10 + (size_t)e->ctx->r;
  icp_evaluator::ChainBinding::Parked& k = e->bind.parked[e->bind.next];
  e->bind.next ^= 1;
  k.cur.assign(cur, cur + P);
  k.prop.assign(prop, prop + P);
  for (int i = 0; i < e->bind.n; ++i) { k.fwd[i] = fwd[i]; k.bwd[i] = bwd[i]; }
  k.valid = true;
}

// logTransitionProbability(from, to) of bound proposal `index`, if a parked step holds it
const double* parked_transition(icp_evaluator* e, int index, const double* from, const double* to) {
  const size_t bytes = sizeof(double) * (10 + (size_t)e->ctx->r);
  for (auto& k : e->bind.parked) {
    if (!k.valid) continue;
    if (std::memcmp(k.cur.data(), from, bytes) == 0 && std::memcmp(k.prop.data(), to, bytes) == 0) return &k.fwd[index];
    if (std::memcmp(k.prop.data(), from, bytes) == 0 && std::memcmp(k.cur.data(), to, bytes) == 0) return &k.bwd[index];
  }
  return nullptr;
}

// propose() of a bound proposal: the chain's whole step with this proposal as the generator.  false: not taken (the step failed — the
// per-method call then reports, or survives, on its own).
bool bound_propose(icp_proposal* p, const double* theta, const double* z, double* theta_out) {
  icp_ctx& c = *p->ctx;
  std::lock_guard<std::recursive_mutex> lk(c.mu);
  icp_evaluator* e = p->bound_eval;
  if (!e || c.batch_busy) return false;
  double lv = 0.0, fwd[8], bwd[8];
  int rc;
  {
    BoundStepScope _s;
    rc = icp_chain_step(e, e->bind.n, e->bind.props, p->bound_index, theta, z, theta_out, &lv, fwd, bwd);
  }
  if (rc != ICP_OK && rc != ICP_ERR_EMPTY) return false;  // (EMPTY is the likelihood's: memoised with the state, logValue reports it)
  park_bound_step(e, theta, theta_out, fwd, bwd);
  ++e->bind.steps_from_propose;
  return true;
}
}  // namespace

extern "C" {
// --------------------------------------------------------------------- chain binding

int icp_chain_bind(icp_evaluator* e, int32_t n_props, icp_proposal* const* props) {
  return guard([&] {
    require(e != nullptr, "null argument");
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || props), "bad proposal list");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) {
      require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
      for (int j = 0; j < i; ++j) require(props[j] != props[i], "a proposal is listed twice");
    }
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    unbind_chain(e);
    for (int i = 0; i < n_props; ++i) {
      if (props[i]->bound_eval) unbind_chain(props[i]->bound_eval);  // (a proposal is a component of ONE chain's mixture)
      props[i]->bound_eval = e;
      props[i]->bound_index = i;
      e->bind.props[i] = props[i];
    }
    e->bind.n = n_props;
  });
}

int icp_chain_bind_stats(const icp_evaluator* e, int64_t out[3]) {
  if (!e || !out) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(e->ctx->mu);
  out[0] = e->bind.steps_from_propose;
  out[1] = e->bind.steps_from_log_value;
  out[2] = e->bind.parked_hits;
  return ICP_OK;
}

// --------------------------------------------------------------------- proposal

int icp_proposal_create(icp_ctx* ctx, const icp_proposal_params* params, icp_proposal** out) {
  if (out) *out = nullptr;
  icp_proposal* p = nullptr;
  int rc = guard([&] {
    require(ctx && params && out, "null argument");
    require(params->direction == ICP_MODEL_SAMPLING || params->direction == ICP_TARGET_SAMPLING, "unknown direction");
    require(params->step_length != 0.0 && std::isfinite(params->step_length), "step_length must be finite and non-zero");
    require(params->tangential_noise > 0.0 && params->noise_along_normal > 0.0, "noise standard deviations must be positive");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    NullStreamBatch _fills;  // (nothing is launched in here: the uploads and fills are waited for together, when the call returns)
    p = new icp_proposal();
    p->ctx = ctx;
    p->prm = *params;
    if (params->direction == ICP_TARGET_SAMPLING) {
      require(params->n_target_points >= 0 && (params->target_points || params->n_target_points == 0), "bad target points");
      p->K = params->n_target_points;
      p->target_pts.upload(params->target_points, 3 * (size_t)p->K);
      p->hint_nn.alloc(std::max(p->K, 1));
      p->hint_nn.fill_bytes(0xFF);
      p->nn_id.alloc(std::max(p->K, 1));
    } else {
      require(params->n_model_ids >= 0 && params->n_model_ids <= ctx->N, "n_model_ids out of range");
      p->K = params->n_model_ids;
    }
    p->prm.target_points = nullptr;  // caller memory is not retained
    p->work.alloc(eigen_work_doubles(ctx->r));
    p->work.fill_bytes(0);  // holds the completion counter of the eigenvector replay kernel
    if (ctx->eig_stream2) {  // ranks above 64: the second eigen stream's work area
      p->work2.alloc(eigen_work_doubles(ctx->r));
      p->work2.fill_bytes(0);
    }
    p->mpart_half_doubles = (size_t)regression_splits(std::max(p->K, 1)) * (ctx->r + 1) * (ctx->r + 1);
    p->Mpart.alloc(icp_proposal::kMpartRing * p->mpart_half_doubles);
    p->fscratch.alloc((size_t)(ctx->r + 1) * ctx->r + 8);
    pinned_alloc((void**)&p->h_cancel, sizeof(int) * 16);
    for (int i = 0; i < 16; ++i) p->h_cancel[i] = 0;
    pinned_alloc((void**)&p->h_eig, sizeof(int) * kPosteriorMemo);
    for (int i = 0; i < kPosteriorMemo; ++i) p->h_eig[i] = 0;
    p->status.alloc(3 * kPosteriorMemo);
    p->status.fill_bytes(0);
    p->eig_words.alloc(kPosteriorMemo);
    p->eig_words.fill_bytes(0);
    p->h_status.assign(3 * kPosteriorMemo, 0);
    p->memo.reset(new PosteriorEntry[kPosteriorMemo]);
    for (int i = 0; i < kPosteriorMemo; ++i) p->alloc_entry(p->memo[i]);
    ctx->proposals.push_back(p);
    *out = p;
  });
  if (rc != ICP_OK && p) delete p;
  return rc;
}

namespace { void release_front(StepFront& F); }

void icp_proposal_destroy(icp_proposal* p) {
  if (!p) return;
  {
    std::lock_guard<std::recursive_mutex> lk(p->ctx->mu);
    (void)hipSetDevice(p->ctx->device);
    (void)hipStreamSynchronize(p->ctx->stream);
    p->ctx->front_stream.sync_quiet();
    try { sync_eigen(*p->ctx); } catch (...) {}
    DeviceQuiesce _q;
    if (p->bound_eval) unbind_chain(p->bound_eval);
    for (icp_evaluator* ev : p->ctx->evaluators)  // a half step launched ahead with this proposal holds entries of it
      if (ev->front.valid && (ev->front.props[0] == p || ev->front.props[1] == p)) release_front(ev->front);
    if (g_host_timing.on && eigen_speculation_supported(p->ctx->r)) eigen_debug_dump(p->work.p, p->ctx->r);
    if (p->h_cancel) pinned_free(p->h_cancel);
    if (p->h_eig) pinned_free(p->h_eig);
    auto& live = p->ctx->proposals;
    live.erase(std::remove(live.begin(), live.end(), p), live.end());
    delete p;
  }
}

int icp_proposal_num_candidates(const icp_proposal* p) { return p ? p->K : ICP_ERR_INVALID_ARG; }

int icp_proposal_set_sampler(icp_proposal* p, int32_t sampler) {
  return guard([&] {
    require(p != nullptr, "null argument");
    require(sampler == ICP_SAMPLER_EIGEN || sampler == ICP_SAMPLER_CHOLESKY_ROOT, "unknown sampler");
    icp_ctx& c = *p->ctx;
    require(sampler == ICP_SAMPLER_EIGEN || c.r <= kCholMaxRankAbi, "the Cholesky-root sampler covers ranks up to 256");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (p->sampler == sampler) return;
    Bound _b(&c);
    // whatever was decomposed (or is being decomposed) the other way is dropped: its V / S mean something else
    HIP_OK(hipStreamSynchronize(c.stream));
    sync_eigen(c);
    for (icp_evaluator* ev : c.evaluators)
      if (ev->front.valid && (ev->front.props[0] == p || ev->front.props[1] == p)) release_front(ev->front);
    // Ranks above 64 have no decomposition of the root kind: there the posterior's own factorisation hands the factor out
    // (PosteriorFactorIO::Lout / Sout, written only when the posterior is computed), and the eigen route has no `root` form.  A
    // memoised posterior would be a memo hit that never rewrites V / S the new way — the sampler would draw from one kind of
    // buffer read as the other.  Those entries are forgotten altogether: the next use recomputes the posterior under the new sampler.
    const bool refactor = !eigen_speculation_supported(c.r);
    for (int i = 0; i < kPosteriorMemo; ++i) {
      p->memo[i].eig_valid = false; p->memo[i].eig_checked = false; p->memo[i].eig_event_valid = false;
      if (refactor) p->memo[i].valid = false;
    }
    p->side_parts = nullptr; p->side_parts_entry = nullptr;
    p->spec_entry = nullptr;
    p->warm_valid = false;
    p->sampler = sampler;
  });
}

int icp_proposal_propose(icp_proposal* p, const double* theta, const double* z, double* theta_out, int32_t* corr_id_out) {
  // (icp_chain_bind) the whole step in this call; a caller that also wants the correspondence ids gets them from the per-method path
  // below — a memo hit on the posterior, the same z: the same numbers
  if (p && p->bound_eval && tl_bound_depth == 0 && theta && z && theta_out) {
    bool finite = true;
    for (int i = 0; i < 10 + p->ctx->r && finite; ++i) finite = std::isfinite(theta[i]);
    for (int j = 0; j < p->ctx->r && finite; ++j) finite = std::isfinite(z[j]);
    if (finite && bound_propose(p, theta, z, theta_out) && !corr_id_out) return ICP_OK;
  }
  return guard([&] {
    require(p && z && theta_out, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r;
    const double* dz = c.stage(z, r);                 // :55 the caller's standard normals (on their way before the wait for the basis)
    PosteriorEntry& e = p->posterior(theta, false);  // NonRigidIcpProposal.scala:54
    p->ensure_eigen(e);
    p->await_eigen(e);
    // (the entry's status words travel with the proposal: one result copy, not two)
    launch_propose(c.stream, r, e.alpha.p, e.V.p, e.S.p, c.inv_sqrt_lambda.p, c.P.p, kSigma2, e.coeffs.p, dz,
                   p->prm.step_length, c.d_res.p, p->sampler == ICP_SAMPLER_CHOLESKY_ROOT, p->status.p + e.status_off, c.d_status.p);
    std::vector<int> ids;
    std::vector<uint8_t> keep;
    if (corr_id_out && p->K > 0) {
      ids.resize(p->K);
      keep.resize(p->K);
      HIP_OK(hipMemcpyAsync(ids.data(), e.id.p, sizeof(int) * p->K, hipMemcpyDeviceToHost, c.stream));
      HIP_OK(hipMemcpyAsync(keep.data(), e.keep.p, p->K, hipMemcpyDeviceToHost, c.stream));
    }
    c.finish(r, 3);
    for (int k = 0; k < 3; ++k) p->h_status.data()[e.status_off + k] = c.h_status[k];
    p->check_status(e);
    static const bool dbg = dev_env("ICP_DEBUG_EIGEN") != nullptr;
    if (dbg) std::fprintf(stderr, "eigen sweeps %d\n", p->h_status[e.status_off + 1]);
    std::memcpy(theta_out, theta, sizeof(double) * 10);
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(c.h_res[j])) fail(ICP_ERR_NOT_FINITE, "proposed coefficients are not finite");
      theta_out[10 + j] = c.h_res[j];
    }
    if (corr_id_out)
      for (int k = 0; k < p->K; ++k) corr_id_out[k] = keep[k] ? ids[k] : -1;
  });
}

int icp_proposal_log_transition(icp_proposal* p, const double* theta_from, const double* theta_to, double* out) {
  return guard([&] {
    require(p && out, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta_from);
    check_theta_finite(&c, theta_to);
    if (!pose_equal(theta_from, theta_to)) {  // NonRigidIcpProposal.scala:72-74
      *out = -INFINITY;
      return;
    }
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (p->bound_eval && tl_bound_depth == 0) {  // (icp_chain_bind) the step that proposed `to` from `from`, or the reverse, has it
      if (const double* v = parked_transition(p->bound_eval, p->bound_index, theta_from, theta_to)) {
        ++p->bound_eval->bind.parked_hits;
        if (std::isnan(*v)) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
        *out = *v;
        return;
      }
    }
    Bound _b(&c);
    PosteriorEntry& e = p->posterior(theta_from, false);  // :76
    const double* dto = c.stage(theta_to + 10, c.r);
    TransitionTailIO io{e.alpha.p, e.M.p, e.coeffs.p, dto, p->prm.step_length, c.d_res.p, c.d_status.p};
    launch_transition_tails(c.stream, c.r, 1, &io, c.Ginv.p, kSigma2);
    sync_proposal_status(p);
    c.finish(1, 1);
    p->check_status(e);
    if (c.h_status[0] != 0) {  // the fixed-point form did not contract for this model/noise: direct factorisation
      const double* dto2 = c.stage(theta_to + 10, c.r);
      io.c_to = dto2;
      sync_eigen(c);  // (the direct form borrows the eigen work buffer)
      launch_transition_tail_direct(c.stream, c.r, io, c.G.p, kSigma2, p->work.p);
      c.finish(1, 1);
      if (c.h_status[0] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
    }
    if (std::isnan(c.h_res[0])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
    *out = c.h_res[0];
  });
}

int icp_proposal_posterior(icp_proposal* p, const double* theta, icp_posterior_view* view) {
  return guard([&] {
    require(p && view, "null argument");
    icp_ctx& c = *p->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r, K = p->K;
    PosteriorEntry& e = p->posterior(theta, view->corr_aux != nullptr);
    if (view->V || view->S) { p->ensure_eigen(e); p->await_eigen(e); }
    view->n_candidates = K;
    auto d2h = [&](void* dst, const void* src, size_t bytes) {
      if (dst && bytes) HIP_OK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c.stream));
    };
    d2h(view->corr_id, e.id.p, sizeof(int) * K);
    d2h(view->corr_aux, e.aux.p, sizeof(int) * K);
    d2h(view->corr_point, e.pt.p, sizeof(double) * 3 * K);
    d2h(view->keep, e.keep.p, K);
    d2h(view->alpha, e.alpha.p, sizeof(double) * r);
    d2h(view->M, e.M.p, sizeof(double) * r * r);
    d2h(view->V, e.V.p, sizeof(double) * r * r);
    d2h(view->S, e.S.p, sizeof(double) * r);
    sync_proposal_status(p);
    c.finish(0, 0);
    p->check_status(e);
  });
}

// --------------------------------------------------------------------- evaluators

int icp_evaluator_create(icp_ctx* ctx, const icp_evaluator_params* params, icp_evaluator** out) {
  if (out) *out = nullptr;
  icp_evaluator* ev = nullptr;
  int rc = guard([&] {
    require(ctx && params && out, "null argument");
    require(params->kind >= 0 && params->kind <= 2, "unknown evaluator kind");
    require(params->kind == ICP_EVAL_HAUSDORFF || (params->mode >= 0 && params->mode <= 2), "unknown evaluation mode");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    NullStreamBatch _fills;  // (nothing is launched in here: the uploads and fills are waited for together, when the call returns)
    ev = new icp_evaluator();
    ev->ctx = ctx;
    ev->prm = *params;
    if (params->kind == ICP_EVAL_HAUSDORFF) {
      require(params->exp_rate > 0.0, "exp_rate must be positive");
      ev->Kt = ctx->target.V;  // MeshMetrics.hausdorffDistance: every target vertex against the model surface
      ev->d_tpts = ctx->target.verts.p;
    } else {
      require(params->gauss_sigma > 0.0, "gauss_sigma must be positive");
      require(params->kind != ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE || params->exp_rate > 0.0, "exp_rate must be positive");
      require(params->n_model_ids >= 0 && params->n_model_ids <= ctx->N, "n_model_ids out of range");
      require(params->n_target_points >= 0 && (params->target_points || params->n_target_points == 0), "bad target points");
      ev->Kt = params->n_target_points;
      ev->target_pts.upload(params->target_points, 3 * (size_t)ev->Kt);
      ev->d_tpts = ev->target_pts.p;
      ev->points_hash = hash_words(0x9abc, params->target_points, sizeof(double) * 3 * (size_t)ev->Kt) | 1;
    }
    ev->prm.target_points = nullptr;
    const size_t Ka = std::max(ev->Kt, 1);
    ev->hint_tri.alloc(Ka);
    ev->hint_nnv.alloc(Ka);
    seed_evaluator_hints(ev);  // (another evaluator's winners for the same points against the same pair, or none)
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ev->t2m_tri.alloc(Ka); ev->t2m_nnv.alloc(Ka);
    ev->t2m_cp.alloc(3 * Ka); ev->t2m_d2.alloc(Ka);
    ctx->evaluators.push_back(ev);
    *out = ev;
  });
  if (rc != ICP_OK && ev) delete ev;
  return rc;
}

void icp_evaluator_destroy(icp_evaluator* e) {
  if (!e) return;
  std::lock_guard<std::recursive_mutex> lk(e->ctx->mu);
  (void)hipSetDevice(e->ctx->device);
  (void)hipStreamSynchronize(e->ctx->stream);
  DeviceQuiesce _q;
  unbind_chain(e);
  // a pre-launched half step holds a state slot of the context and memo entries of its proposals
  if (e->front.valid) release_front(e->front);
  auto& evs = e->ctx->evaluators;
  evs.erase(std::remove(evs.begin(), evs.end(), e), evs.end());
  delete e;
}

int icp_evaluator_log_value(icp_evaluator* e, const double* theta, double* out, double* aux) {
  int status = ICP_OK;
  int rc = guard([&] {
    require(e && out, "null argument");
    icp_ctx& c = *e->ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    if (c.batch_busy) fail(ICP_ERR_BUSY, "the context belongs to a batch in flight (icp_chain_step_batched_issue): collect or abandon it first");
    icp_evaluator::Memo* m = eval_lookup(e, theta);  // evaluators/EvaluationCaching.scala:32-36 (a hit touches no stream)
    const size_t P = 10 + (size_t)c.r;
    const bool bound = e->bind.n > 0 && tl_bound_depth == 0;
    if (!m && bound && !e->bind.last_theta.empty() && std::memcmp(e->bind.last_theta.data(), theta, sizeof(double) * P) != 0) {
      // (icp_chain_bind) a state made on the host — a random-walk or pose proposal — from the state of the previous logValue call
      // (MetropolisHastings.next evaluates the current state, proposes, evaluates the proposal): the chain's whole step now
      std::vector<double> cur(e->bind.last_theta), prop(theta, theta + P);
      double lv = 0.0, fwd[8], bwd[8];
      int st;
      {
        BoundStepScope _s;
        st = icp_chain_step(e, e->bind.n, e->bind.props, -1, cur.data(), nullptr, prop.data(), &lv, fwd, bwd);
      }
      if (st == ICP_OK || st == ICP_ERR_EMPTY) {
        park_bound_step(e, cur.data(), theta, fwd, bwd);
        ++e->bind.steps_from_log_value;
        m = eval_lookup(e, theta);
      }
    }
    if (bound) e->bind.last_theta.assign(theta, theta + P);
    if (!m) {
      Bound _b(&c);
      adopt_hints(e);  // (hints another chain of this model and target has filed since this context was made)
      StateSlot& s = c.state(theta);
      enqueue_eval(e, s, 0);
      c.finish(8, 0);
      file_hints(e);  // (once per context / evaluator: the first completed evaluation's winners start the next chains' searches)
      m = eval_store(e, theta);
      m->status = finish_eval(e, c.h_res, &m->value, m->aux);
    }
    *out = m->value;
    if (aux) std::memcpy(aux, m->aux, sizeof(double) * 4);
    status = m->status;
    if (status != ICP_OK) g_err = icp_status_string(status);
  });
  return rc != ICP_OK ? rc : status;
}

int icp_prior_log_value(int32_t rank, const double* theta, double* out) {
  return guard([&] {
    require(rank > 0 && theta && out, "bad argument");
    double nn = 0.0;
    for (int j = 0; j < rank; ++j) nn += theta[10 + j] * theta[10 + j];
    *out = -0.5 * nn - 0.5 * rank * std::log(2.0 * M_PI);  // MultivariateNormalDistribution(0, I).logpdf
  });
}

// --------------------------------------------------------------------- deterministic non-rigid ICP (next row 1)

int icp_fit_deterministic(icp_ctx* ctx, const icp_fit_params* prm, const double* theta_init, int32_t n_iterations, int32_t n_sigma,
                          const double* sigma2_seq, double* theta_out) {
  return guard([&] {
    require(ctx && prm && theta_out && sigma2_seq, "null argument");
    require(n_iterations >= 0 && n_sigma >= 0, "negative iteration count");
    require(prm->direction == ICP_MODEL_SAMPLING || prm->direction == ICP_TARGET_SAMPLING, "unknown direction");
    require(std::isfinite(prm->step_length), "step_length must be finite");
    icp_ctx& c = *ctx;
    check_theta_finite(&c, theta_init);
    const bool model_side = prm->direction == ICP_MODEL_SAMPLING;
    const int K = model_side ? prm->n_model_ids : prm->n_target_points;
    require(K >= 0 && (K == 0 || (model_side ? (const void*)prm->model_ids : (const void*)prm->target_points)), "bad sample list");
    if (model_side)
      for (int k = 0; k < K; ++k) require(prm->model_ids[k] >= 0 && prm->model_ids[k] < c.N, "model id out of range");
    for (int i = 0; i < n_sigma; ++i) require(sigma2_seq[i] > 0.0 && std::isfinite(sigma2_seq[i]), "sigma2 must be positive");
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r, Ka = std::max(K, 1);
    const Pose pose = c.pose_of(theta_init);
    DBuf<double> coeffs, x, P, cp, pts, Mpart, M, alpha, e, nhat, pt;
    DBuf<int> ids, nn, hint, corr_id, aux, status;
    DBuf<uint8_t> keep;
    coeffs.upload(theta_init + 10, r);
    x.alloc(3 * (size_t)c.N);
    P.alloc(3 * (size_t)Ka); cp.alloc(3 * (size_t)Ka);
    hint.alloc(Ka); hint.fill_bytes(0xFF);
    nn.alloc(Ka); corr_id.alloc(Ka); aux.alloc(Ka); keep.alloc(Ka);
    e.alloc(3 * (size_t)Ka); nhat.alloc(3 * (size_t)Ka); pt.alloc(3 * (size_t)Ka);
    if (model_side) ids.upload(prm->model_ids, K);
    else pts.upload(prm->target_points, 3 * (size_t)K);
    Mpart.alloc((size_t)regression_splits(Ka) * (r + 1) * (r + 1));
    M.alloc((size_t)r * r); alpha.alloc(r);
    DBuf<double> fscratch;
    fscratch.alloc((size_t)(r + 1) * r + 8);
    status.alloc(4); status.fill_bytes(0);
    const CorrBuffers cb{corr_id.p, aux.p, pt.p, keep.p, nhat.p, e.p};
    for (int si = 0; si < n_sigma; ++si) {
      const double wt = 1.0 / sigma2_seq[si];                                     // isotropic noise N(0, sigma2·I) (:81)
      for (int it = 0; it <= n_iterations; ++it) {                                // nbIterations = numIterations .. 0 (:55-104)
        launch_instance(c.stream, c.N, r, c.Qp.p, c.ref.p, c.mean.p, pose, coeffs.p, x.p);      // :61
        if (K > 0) {
          if (model_side) {                                                       // :72-74
            launch_gather_points(c.stream, K, x.p, ids.p, P.p);
            QueryBuffers qb = c.query_scratch(K, c.target.T);
            launch_surface_query(c.stream, c.target.T, c.target.verts.p, c.target.tris.p, c.target.spheres.p, K, P.p, hint.p, qb, cp.p,
                                 nullptr, nullptr);
            launch_correspond_plain(c.stream, K, ids.p, cp.p, c.ref.p, c.mean.p, cb);
          } else {                                                                // :76-78
            QueryBuffers qb = c.query_scratch(K, c.N);
            launch_vertex_query(c.stream, c.N, x.p, K, pts.p, hint.p, qb, nullptr, nn.p);
            launch_correspond_plain(c.stream, K, nn.p, pts.p, c.ref.p, c.mean.p, cb);
          }
        }
        int splits = 1;
        launch_regression(c.stream, K, r, c.Q.p, cb, wt, 0.0, Mpart.p, &splits);   // model.posterior(corr, sigma2) (:81)
        PosteriorFactorIO io{Mpart.p, splits, M.p, alpha.p, status.p, fscratch.p};
        launch_posterior_factor(c.stream, r, 1, &io);                              // posterior.mean (:82)
        launch_mean_step(c.stream, r, alpha.p, c.P.p, kSigma2, prm->step_length, coeffs.p);   // :84-85
      }
    }
    HIP_OK(hipMemcpyAsync(c.h_res, coeffs.p, sizeof(double) * r, hipMemcpyDeviceToHost, c.stream));
    HIP_OK(hipMemcpyAsync(c.h_status, status.p, sizeof(int), hipMemcpyDeviceToHost, c.stream));
    c.finish(0, 0);
    if (c.h_status[0] != 0) fail(ICP_ERR_NOT_SPD, "regression normal equations are not positive definite");
    std::memcpy(theta_out, theta_init, sizeof(double) * 10);
    for (int j = 0; j < r; ++j) {
      if (!std::isfinite(c.h_res[j])) fail(ICP_ERR_NOT_FINITE, "fitted coefficients are not finite");
      theta_out[10 + j] = c.h_res[j];
    }
  });
}

// --------------------------------------------------------------------- posterior variability maps (next row 3)

int icp_posterior_variability(icp_ctx* ctx, int32_t n_samples, const double* thetas, int32_t mode, const double* theta_ref, double* out) {
  return guard([&] {
    require(ctx && thetas && out, "null argument");
    require(n_samples >= 2, "at least two samples are needed");
    require(mode >= 0 && mode <= 2, "unknown mode");
    require(mode != 1 || theta_ref, "theta_ref is null");
    icp_ctx& c = *ctx;
    const size_t P = 10 + (size_t)c.r, n3 = 3 * (size_t)c.N;
    for (int s = 0; s < n_samples; ++s) check_theta_finite(&c, thetas + s * P);
    if (mode == 1) check_theta_finite(&c, theta_ref);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    DBuf<double> X, coeffs, nrm, tmp, res;
    X.alloc((size_t)n_samples * n3);
    std::vector<double> hc((size_t)(n_samples + 1) * c.r);
    for (int s = 0; s < n_samples; ++s) std::memcpy(&hc[(size_t)s * c.r], thetas + s * P + 10, sizeof(double) * c.r);
    if (mode == 1) std::memcpy(&hc[(size_t)n_samples * c.r], theta_ref + 10, sizeof(double) * c.r);
    coeffs.upload(hc.data(), hc.size());
    nrm.alloc(n3); tmp.alloc(n3); res.alloc(c.N);
    for (int s = 0; s < n_samples; ++s)   // ModelFittingParameters.transformedMesh of every sample (LogHelper.logSamples2shapes)
      launch_instance(c.stream, c.N, c.r, c.Qp.p, c.ref.p, c.mean.p, c.pose_of(thetas + s * P), coeffs.p + (size_t)s * c.r,
                      X.p + (size_t)s * n3);
    if (mode == 1) {
      launch_instance(c.stream, c.N, c.r, c.Qp.p, c.ref.p, c.mean.p, c.pose_of(theta_ref), coeffs.p + (size_t)n_samples * c.r, tmp.p);
      launch_vertex_normals(c.stream, c.N, tmp.p, c.tris.p, c.adj_off.p, c.adj.p, nrm.p);
    } else if (mode == 2) {
      HIP_OK(hipMemsetAsync(nrm.p, 0, sizeof(double) * n3, c.stream));
      for (int s = 0; s < n_samples; ++s) {
        launch_vertex_normals(c.stream, c.N, X.p + (size_t)s * n3, c.tris.p, c.adj_off.p, c.adj.p, tmp.p);
        launch_accumulate(c.stream, (int)n3, tmp.p, s == n_samples - 1 ? 1.0 / n_samples : 0.0, nrm.p);
      }
    }
    launch_variability(c.stream, c.N, n_samples, X.p, mode, nrm.p, res.p);
    HIP_OK(hipMemcpyAsync(out, res.p, sizeof(double) * c.N, hipMemcpyDeviceToHost, c.stream));
    c.finish(0, 0);
  });
}

// --------------------------------------------------------------------- registration metrics (next row 4)

int icp_mesh_metrics(icp_ctx* ctx, const double* theta, double* out) {
  return guard([&] {
    require(ctx && out, "null argument");
    icp_ctx& c = *ctx;
    check_theta_finite(&c, theta);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    StateSlot& s = c.state(theta);
    double* res = c.d_res.p;
    HIP_OK(hipMemsetAsync(res, 0, sizeof(double) * 16, c.stream));
    // reconstruction -> target: every model vertex against the target surface (shared with the proposals/evaluators of the state)
    c.ensure_surface_prefix(s, c.N);
    launch_dist_stats(c.stream, c.N, s.surf_d2.p, nullptr, nullptr, 0, res + 0);                       // avgDistance, one-sided max
    const bool flags = c.target.n_boundary > 0;
    if (flags) c.ensure_nnv_prefix(s, c.N);
    launch_dist_stats(c.stream, c.N, s.surf_d2.p, flags ? c.target.boundary.p : nullptr, flags ? s.surf_nnv.p : nullptr, c.target.V,
                      res + 4);                                                                        // boundary-aware (:31-42)
    // target -> reconstruction: every target vertex against the current model surface (hausdorffDistance is symmetric)
    DBuf<double> d2;
    DBuf<int> hint;
    d2.alloc(c.target.V); hint.alloc(c.target.V); hint.fill_bytes(0xFF);
    c.ensure_model_spheres(s);
    QueryBuffers qb = c.query_scratch(c.target.V, c.T);
    launch_surface_query(c.stream, c.T, s.x.p, c.tris.p, s.spheres.p, c.target.V, c.target.verts.p, hint.p, qb, nullptr, d2.p, nullptr);
    launch_dist_stats(c.stream, c.target.V, d2.p, nullptr, nullptr, 0, res + 8);
    c.finish(12, 0);
    const double* h = c.h_res;
    out[0] = h[0] / h[2];
    out[1] = std::max(h[1], h[9]);
    out[2] = h[6] > 0.0 ? h[4] / h[6] : NAN;
    out[3] = h[6] > 0.0 ? h[5] : NAN;
    out[4] = h[6];
  });
}

// --------------------------------------------------------------------- fused chain step

} // extern "C" (helper)
namespace {
// 0: never, 1: always, 2: adaptive (ICP_SPECULATION / ICP_NO_SPECULATION)
int speculation_mode() {
  static const int mode = [] {
    if (std::getenv("ICP_NO_SPECULATION")) return 0;
    const char* v = std::getenv("ICP_SPECULATION");
    return v ? (std::atoi(v) != 0 ? 1 : 0) : 2;
  }();
  return mode;
}
}  // namespace
extern "C" {
int icp_chain_eval_step(icp_evaluator* e, int32_t n_props, icp_proposal* const* props, const double* theta_cur,
                        const double* theta_prop, double* log_value_prop, double* fwd, double* bwd) {
  int status = ICP_OK;
  int rc = guard([&] {
    require(e && theta_cur && theta_prop && log_value_prop, "null argument");
    require(n_props >= 0 && n_props <= 8 && (n_props == 0 || (props && fwd && bwd)), "bad proposal list");
    icp_ctx& c = *e->ctx;
    for (int i = 0; i < n_props; ++i) require(props[i] && props[i]->ctx == &c, "proposal belongs to another context");
    check_theta_finite(&c, theta_cur);
    check_theta_finite(&c, theta_prop);
    std::lock_guard<std::recursive_mutex> lk(c.mu);
    Bound _b(&c);
    const int r = c.r;
    icp_evaluator::Memo* m = eval_lookup(e, theta_prop);
    const bool need_eval = m == nullptr;
    const bool shape_only = pose_equal(theta_cur, theta_prop);
    // Ranks above 64 (one workgroup factors, one reduces to tridiagonal form: 0.2 + 0.7 ms at rank 200 with most of the chip idle):
    // the posteriors go first and the proposed state's decomposition starts at once on the eigen stream, BESIDE the evaluator's
    // searches on this one — if the state is accepted, the next proposal finds its basis done or under way; if not, the work
    // was done on CUs nobody needed.  (Acceptance tracked as in the merged step; not worth it when next to nothing is accepted.)
    if (!e->last_prop.empty()) {
      const bool accepted = std::memcmp(e->last_prop.data(), theta_cur, sizeof(double) * (10 + (size_t)r)) == 0;
      e->acc_ema = 0.9 * e->acc_ema + (accepted ? 0.1 : 0.0);
    }
    const int spec_mode = speculation_mode();
    const bool spec_ok = r > 64 && n_props == 1 && need_eval && !c.speculation_off &&
                         (spec_mode == 1 || (spec_mode == 2 && e->acc_ema >= 0.1));
    const bool spec_big = spec_ok && shape_only;
    // a pose move changes the state too: if it is kept, the next ICP proposal draws from the posterior at the NEW state — which
    // nothing on this path computes (the transition probabilities across a pose change are zero).  Started here, ahead, it is
    // done or under way by then: search (the evaluator's own, for a model-sampling proposal), regression, factorisation and
    // decomposition, all beside the evaluator.
    const bool spec_pose = spec_ok && !shape_only;
    if (need_eval && !spec_big && !spec_pose) {
      StateSlot& s = c.state(theta_prop);
      enqueue_eval(e, s, 0);
    }
    PosteriorEntry* ec[8];
    PosteriorEntry* ep[8];
    TransitionTailIO tails[16];
    int n_tails = 0;
    bool eval_enqueued = false;
    // (… and so do the one-workgroup factorisations and the tails: they go to a stream of their own, behind the regression; the
    // decomposition follows them on the eigen stream; this stream goes on with the searches and waits for the tails before the
    // results are copied)
    const hipStream_t side = spec_big || spec_pose ? c.front_stream.get() : nullptr;  // (the merged step's second stream: idle on this path)
    // the decomposition of a posterior whose factorisation has just gone to the side stream: behind that — or, if the posterior was
    // computed just now, beside it: M = I + the summed partials is written by a launch at the head of the decomposition as well
    // (the same values the factorisation's assembly writes)
    hipStream_t es_ahead = nullptr;  // the stream part 1 of a decomposition went to (part 2 follows it there)
    auto decompose_ahead = [&](icp_proposal* p, PosteriorEntry& en, int part = 0) {
      if (part == 2) {
        if (es_ahead) p->ensure_eigen_on(en, es_ahead, 2);
        return;
      }
      if (en.eig_valid) return;
      if (c.eig_last && c.eig_last != c.eig_stream.get()) (void)eigen_stream_for(c, c.eig_stream.get());  // (a batch's stream was in use: drained)
      c.eig_last = c.eig_stream.get();
      const hipStream_t es = (c.eig_stream2 && (p->eig_flip++ & 1)) ? c.eig_stream2.get() : c.eig_stream.get();  // two under way at a time
      es_ahead = es;
      if (p->side_parts && p->side_parts_entry == &en) {
        // (computed just now, all of it on the side stream: nothing of this entry is on the context stream — which carries the
        // evaluator's searches by now, and the decomposition must not wait for those)
        HIP_OK(hipStreamWaitEvent(es, c.ev_sum, 0));
        launch_assemble_posterior_matrix(es, r, p->side_parts, en.M.p);
        HIP_OK(hipEventRecord(c.ev_asm, es));
        p->side_asm_pending = true;
      } else {
        HIP_OK(hipEventRecord(c.ev_ready, c.stream));  // (whatever of this entry is still in flight on the context stream)
        HIP_OK(hipStreamWaitEvent(es, c.ev_ready, 0));
        HIP_OK(hipStreamWaitEvent(es, c.ev_side, 0));
      }
      p->ensure_eigen_on(en, es, part);
    };
    // Order of issue on this path (the host needs 3-6 µs per launch, the context stream is idle until it gets the evaluator's):
    // the proposed state's instance -> the evaluator's searches (the model ids the posterior's own searches will cover — 0..K — left
    // out: StateSlot's detached ranges) and its target-to-model half -> the posterior on the side stream (searches of ids 0..K,
    // correspondences, regression, factorisation) -> the head of the decomposition -> the evaluator's model-to-target reductions,
    // behind the side stream's searches -> the decomposition's other launches -> the tails.
    // the two coefficient vectors the tails read: the states' own copies on the device (the current state's slot is kept from
    // being recycled for the proposed one), staged from the host only for a current state that has no slot any more — here, ahead of
    // the evaluator's launches: ev_inst covers them
    const double *d_cur = nullptr, *d_prop = nullptr;
    if (shape_only && n_props > 0) {
      StateSlot* sc = c.find_state(theta_cur);
      if (sc) sc->stamp = ++c.clock;
      d_cur = sc ? sc->coeffs.p : c.stage(theta_cur + 10, r);
      d_prop = c.state(theta_prop).coeffs.p;
    }
    bool split_eval = false;
    if (side && need_eval && n_props == 1) {
      icp_proposal* p0 = props[0];
      PosteriorEntry* known = p0->find_entry(theta_prop);
      if (known) known->stamp = ++p0->clock;  // (not the one a posterior of the current state, computed first, recycles)
      const int R = (!known && p0->prm.direction == ICP_MODEL_SAMPLING) ? p0->K : 0;
      const int Rn = (R > 0 && p0->prm.boundary_aware && c.target.n_boundary > 0) ? R : 0;
      StateSlot& s = c.state(theta_prop);
      HIP_OK(hipEventRecord(c.ev_inst, c.stream));
      c.ev_inst_slot = &s;
      enqueue_eval_searches(e, s, 0, R, Rn);
      split_eval = true;
    }
    struct InstGuard { icp_ctx& c; ~InstGuard() { c.ev_inst_slot = nullptr; } } inst_guard{c};
    auto eval_reductions = [&] {
      if (split_eval && !eval_enqueued) { enqueue_eval_reductions(e, c.state(theta_prop), 0); eval_enqueued = true; }
    };
    PosteriorEntry* pose_entry = nullptr;
    if (spec_pose) {
      PosteriorEntry& en = props[0]->posterior(theta_prop, false, side);
      decompose_ahead(props[0], en, 1);  // (its other launches: behind the evaluator's, below)
      pose_entry = &en;
      eval_reductions();
    }
    if (shape_only && n_props > 0) {
      for (int i = 0; i < n_props; ++i) {
        icp_proposal* p = props[i];
        ec[i] = &p->posterior(theta_cur, false, side);
        ep[i] = &p->posterior(theta_prop, false, side);
        // (each tail passes its posterior's status words on to d_status[16 + 3·tail …]: they come back with the step's results)
        tails[n_tails] = TransitionTailIO{ec[i]->alpha.p, ec[i]->M.p, d_cur, d_prop, p->prm.step_length, c.d_res.p + 8 + n_tails,
                                          c.d_status.p + n_tails, p->status.p + ec[i]->status_off, c.d_status.p + 16 + 3 * n_tails};
        ++n_tails;
        tails[n_tails] = TransitionTailIO{ep[i]->alpha.p, ep[i]->M.p, d_prop, d_cur, p->prm.step_length, c.d_res.p + 8 + n_tails,
                                          c.d_status.p + n_tails, p->status.p + ep[i]->status_off, c.d_status.p + 16 + 3 * n_tails};
        ++n_tails;
      }
      if (side) {
        // (the staged coefficients and — for entries found in the memo — everything else the tails read: behind ev_inst, which was
        // recorded ahead of the evaluator's searches, or behind ev_ready)
        if (split_eval) {
          HIP_OK(hipStreamWaitEvent(side, c.ev_inst, 0));
        } else {
          HIP_OK(hipEventRecord(c.ev_ready, c.stream));
          HIP_OK(hipStreamWaitEvent(side, c.ev_ready, 0));
        }
        // Order of issue (the host needs 3-6 µs per launch): factorisation (inside posterior) -> the HEAD of the decomposition (the
        // reduction to tridiagonal form, 0.48 ms on one workgroup) -> the evaluator's ten launches -> the decomposition's other ten
        // launches (they run behind the reduction whenever they are issued) -> the tails (behind the factorisation: 0.1-0.3 ms of
        // slack).  The evaluator's searches used to start 70-120 µs after the regression had ended because they were issued last.
        decompose_ahead(props[0], *ep[0], 1);
        eval_reductions();
        if (need_eval && !eval_enqueued) {
          StateSlot& s = c.state(theta_prop);
          enqueue_eval(e, s, 0);
          eval_enqueued = true;
        }
        decompose_ahead(props[0], *ep[0], 2);
      }
      for (int t0 = 0; t0 < n_tails; t0 += 8)
        launch_transition_tails(side ? side : c.stream, r, std::min(8, n_tails - t0), tails + t0, c.Ginv.p, kSigma2);
      if (side) {
        HIP_OK(hipEventRecord(c.ev_side, side));
        props[0]->side_factor_pending = false;  // (this stream waits for ev_side below)
      }
    }
    eval_reductions();
    if (need_eval && (spec_big || spec_pose) && !eval_enqueued) {
      StateSlot& s = c.state(theta_prop);
      enqueue_eval(e, s, 0);
    }
    if (pose_entry) decompose_ahead(props[0], *pose_entry, 2);
    if (side && shape_only && n_props > 0) HIP_OK(hipStreamWaitEvent(c.stream, c.ev_side, 0));
    e->last_prop.assign(theta_prop, theta_prop + 10 + r);
    c.finish(8 + (size_t)n_tails, (size_t)n_tails);
    for (int t = 0; t < n_tails; ++t)
      if (c.h_status[t] != 0) {  // rare: fixed-point tail did not contract -> direct kernel, one at a time
        std::vector<double> saved(c.h_res, c.h_res + 8 + n_tails);
        icp_proposal* p = props[t / 2];
        TransitionTailIO io = tails[t];
        io.c_from = c.stage((t % 2 == 0 ? theta_cur : theta_prop) + 10, r);
        io.c_to = c.stage((t % 2 == 0 ? theta_prop : theta_cur) + 10, r);
        io.out = c.d_res.p;
        io.status = c.d_status.p + 64;
        io.relay_in = nullptr; io.relay_out = nullptr;
        sync_eigen(c);  // (the direct form borrows the eigen work buffer)
        launch_transition_tail_direct(c.stream, r, io, c.G.p, kSigma2, p->work.p);
        c.finish(1, 96);
        if (c.h_status[64] != 0) fail(ICP_ERR_NOT_SPD, "G + sigma^2 M is not positive definite");
        saved[8 + t] = c.h_res[0];
        std::memcpy(c.h_res, saved.data(), sizeof(double) * saved.size());
      }
    if (need_eval) {
      m = eval_store(e, theta_prop);
      m->status = finish_eval(e, c.h_res, &m->value, m->aux);
    }
    *log_value_prop = m->value;
    status = m->status;
    for (int i = 0; i < n_props; ++i) {
      if (!shape_only) { fwd[i] = -INFINITY; bwd[i] = -INFINITY; continue; }
      for (int k = 0; k < 3; ++k) {
        props[i]->h_status.data()[ec[i]->status_off + k] = c.h_status[16 + 3 * (2 * i) + k];
        props[i]->h_status.data()[ep[i]->status_off + k] = c.h_status[16 + 3 * (2 * i + 1) + k];
      }
      props[i]->check_status(*ec[i]);
      props[i]->check_status(*ep[i]);
      fwd[i] = c.h_res[8 + 2 * i];
      bwd[i] = c.h_res[9 + 2 * i];
      if (std::isnan(fwd[i]) || std::isnan(bwd[i])) fail(ICP_ERR_NOT_FINITE, "NaN transition probability");
    }
  });
  return rc != ICP_OK ? rc : status;
}

// --------------------------------------------------------------------- one Metropolis–Hastings step, one submission

namespace {

// the merged-launch pipeline covers the configurations of the reference's experiments that run on closed targets
// (apps/femur/*): one proposal per ICP direction, model-to-target likelihood; everything else takes the per-stage path
bool step_pipeline_covers(icp_evaluator* e, int n_props, icp_proposal* const* props) {
  icp_ctx& c = *e->ctx;
  if (n_props < 1 || n_props > 2) return false;
  const icp_evaluator_params& ep = e->prm;
  if (ep.kind == ICP_EVAL_HAUSDORFF) return false;
  // (the boundary-aware collective evaluator on a target WITH boundary needs the nearest-vertex pass behind the searches)
  if (ep.kind == ICP_EVAL_COLLECTIVE_AVG_HAUSDORFF_BOUNDARY_AWARE && c.target.n_boundary > 0) return false;
  const bool m2t = ep.mode != ICP_TARGET_TO_MODEL, t2m = ep.mode != ICP_MODEL_TO_TARGET;
  if (m2t && ep.n_model_ids < 1) return false;
  if (t2m && (e->Kt < 1 || c.T < 1 || (size_t)(e->Kt + 4) * (size_t)cand_stride(c.T) > kMaxCandidates)) return false;
  int n_model = 0, n_target = 0, ksurf = m2t ? ep.n_model_ids : 0;
  for (int i = 0; i < n_props; ++i) {
    const icp_proposal* p = props[i];
    if (p->K < 1) return false;
    // (the Cholesky-root sampler above rank 64 gets its factor from the per-stage factorisation: icp_proposal::posterior)
    if (p->sampler == ICP_SAMPLER_CHOLESKY_ROOT && !eigen_speculation_supported(c.r)) return false;
    if (p->prm.direction == ICP_MODEL_SAMPLING) {
      if (p->prm.boundary_aware && c.target.n_boundary > 0) return false;  // needs the nearest-vertex pass (:98-99)
      ++n_model;
      ksurf = std::max(ksurf, p->K);
    } else {
      ++n_target;
    }
  }
  if (n_model > 1 || n_target > 1) return false;
  if (ksurf < 1) return false;  // (a TargetToModel evaluator beside a TargetSampling proposal alone: no model-side surface query at all)
  if (c.target.T < 1 || (size_t)(ksurf + 4) * (size_t)cand_stride(c.target.T) > kMaxCandidates) return false;
  if (n_target && (size_t)(props[0]->K + props[n_props - 1]->K + 8) * (size_t)cand_stride(c.N) > kMaxCandidates) return false;
  return step_finish_supported(c.r);
}

}  // namespace

}  // extern "C"
