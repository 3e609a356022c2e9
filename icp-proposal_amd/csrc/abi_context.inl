// abi_context.inl — part of icp_abi.hip (one translation unit; included there, in order).
// C ABI: contexts — create / destroy / set_target / set_rotation, counters, profiling, geometry entry points
namespace {
// the target's immutable device data (vertices, triangles, boundary flags, bounding spheres in patch order): shared by every context of
// the device made from the same arrays (g_shared_targets).  The caller holds g_shared_mu and has bound the device.
void attach_target(icp_ctx* ctx, const icp_mesh_desc* target, int device) {
  uint64_t th = hash_words(0x5678, target->points, sizeof(double) * 3 * (size_t)target->n_points);
  th = hash_words(th, target->triangles, sizeof(int32_t) * 3 * (size_t)target->n_triangles);
  const SharedKey tkey{device, target->n_points, target->n_triangles, 0, th};
  // (a batch registration attaches one target after the other: the entries of targets nobody holds any more are dropped)
  for (auto it = g_shared_targets.begin(); it != g_shared_targets.end();)
    it = it->second.expired() ? g_shared_targets.erase(it) : std::next(it);
  std::shared_ptr<SharedTarget> stg = g_shared_targets[tkey].lock();
  if (!stg) {
    stg = std::make_shared<SharedTarget>();
    DeviceMesh& tg = stg->mesh;
    tg.V = target->n_points; tg.T = target->n_triangles;
    std::vector<uint8_t> tb;
    boundary_flags(tg.V, tg.T, target->triangles, tb);
    tg.n_boundary = (int)std::count(tb.begin(), tb.end(), (uint8_t)1);
    tg.verts.upload(target->points, (size_t)3 * tg.V);
    tg.tris.upload(target->triangles, (size_t)3 * tg.T);
    tg.boundary.upload(tb.data(), tb.size());
    tg.spheres.alloc(sphere_floats4(tg.T));
    {
      const std::vector<int> order = coherent_triangle_order(tg.V, tg.T, target->points, target->triangles);
      tg.tri_order.upload(order.data(), order.size());
    }
    launch_tri_spheres(ctx->stream, tg.T, tg.verts.p, tg.tris.p, tg.tri_order.p, tg.spheres.p);
    HIP_OK(hipStreamSynchronize(ctx->stream));  // (complete before another context may find it)
    g_shared_targets[tkey] = stg;
  }
  ctx->shared_target = stg;
  DeviceMesh& tg = ctx->target;
  const DeviceMesh& o = stg->mesh;
  tg.V = o.V; tg.T = o.T; tg.n_boundary = o.n_boundary;
  tg.verts.alias(o.verts); tg.tris.alias(o.tris); tg.tri_order.alias(o.tri_order); tg.spheres.alias(o.spheres); tg.boundary.alias(o.boundary);
}
}  // namespace

// ===================================================================== C ABI

extern "C" {

const char* icp_status_string(int status) {
  switch (status) {
    case ICP_OK: return "ok";
    case ICP_ERR_INVALID_ARG: return "invalid argument";
    case ICP_ERR_DEVICE: return "HIP device error";
    case ICP_ERR_NOT_FINITE: return "non-finite result";
    case ICP_ERR_NOT_SPD: return "matrix not positive definite";
    case ICP_ERR_EMPTY: return "no points left after the boundary filter";
    case ICP_ERR_BUSY: return "context busy: part of a batch in flight";
    default: return "unknown status";
  }
}

const char* icp_last_error(void) { return g_err.c_str(); }

int icp_ctx_rank(const icp_ctx* ctx) { return ctx ? ctx->r : ICP_ERR_INVALID_ARG; }
int icp_ctx_device(const icp_ctx* ctx) { return ctx ? ctx->device : ICP_ERR_INVALID_ARG; }

int icp_ctx_expect(int device, int32_t n_contexts) {
  return guard([&] {
    require(n_contexts >= 0, "negative count");
    if (device < 0) {
      const char* lr = std::getenv("LOCAL_RANK");
      device = lr ? std::atoi(lr) : 0;
    }
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device >= n_dev) fail(ICP_ERR_DEVICE, "no such HIP device");
    prewarm_streams(device, std::min(n_contexts, 64));
  });
}

int icp_ctx_create(const icp_model_desc* model, const icp_mesh_desc* target, int device, icp_ctx** out) {
  return icp_ctx_create_keyed(model, target, device, 0, out);
}

int icp_ctx_create_keyed(const icp_model_desc* model, const icp_mesh_desc* target, int device, uint64_t model_key, icp_ctx** out) {
  if (out) *out = nullptr;
  icp_ctx* ctx = nullptr;
  int rc = guard([&] {
    require(model && target && out, "null argument");
    require(model->n_points > 0 && model->n_triangles >= 0 && model->rank > 0 && model->rank <= kMaxRank,
            "model sizes out of range (rank must be in [1,500])");
    require(model->ref_points && model->basis && model->variance && (model->triangles || model->n_triangles == 0),
            "model arrays missing");
    require(target->n_points > 0 && target->n_triangles >= 0 && target->points &&
                (target->triangles || target->n_triangles == 0),
            "target arrays missing");
    const int N = model->n_points, T = model->n_triangles, r = model->rank;
    check_triangles(N, T, model->triangles, "model");
    check_triangles(target->n_points, target->n_triangles, target->triangles, "target");
    for (int j = 0; j < r; ++j) require(model->variance[j] > 0.0 && std::isfinite(model->variance[j]), "variance must be positive");

    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0)
      fail(ICP_ERR_DEVICE, std::string("no usable HIP device (this library has no CPU fallback): ") + hipGetErrorString(e));
    if (device < 0) {
      const char* lr = std::getenv("LOCAL_RANK");
      device = lr ? std::atoi(lr) % ndev : 0;
    }
    require(device < ndev, "device ordinal out of range");

    static const bool create_timing = std::getenv("ICP_CREATE_TIMING") != nullptr;  // (operational: where a context's creation goes)
    auto t_mark = std::chrono::steady_clock::now();
    auto mark = [&](const char* what) {
      if (!create_timing) return;
      const auto now = std::chrono::steady_clock::now();
      std::fprintf(stderr, "[icp create timing] context: %s %.0f us\n", what, std::chrono::duration<double, std::micro>(now - t_mark).count());
      t_mark = now;
    };
    ctx = new icp_ctx();
    ctx->device = device;
    ctx->N = N; ctx->T = T; ctx->r = r;
    ctx->bind();
    mark("checks, bind");
    // The runtime multiplexes streams onto a small pool of hardware queues PER PRIORITY (four by default), and two
    // streams on one hardware queue run one kernel at a time: with other streams alive in the process (torch's,
    // RCCL's: default priority) the two step streams ended up sharing a queue and a step cost 15 % more (measured under
    // torch.distributed.run).  The streams of the first context of a process — the one-chain-per-GPU layout — are
    // therefore created at the greatest priority: a pool of their own.  Further contexts (several chains on one GPU
    // from one process) take the default priority: with every stream in the greatest-priority pool their aggregate
    // rate fell from 22k to 15k it/s (tools/multichain.py, 4-16 contexts).
    int prio_least = 0, prio_greatest = 0;
    HIP_OK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    if (g_live_contexts.load(std::memory_order_relaxed) > 0) prio_greatest = 0;
    if (const char* sp = dev_env("ICP_STREAM_PRIORITY")) {  // A/B switch: 0 = default priority everywhere
      if (std::atoi(sp) == 0) prio_greatest = 0;
    }
    // (out of the pool of streams of destroyed contexts where it has any of that priority class: take_stream)
    const bool greatest = prio_greatest != 0;
    ctx->stream = take_stream(device, greatest, prio_greatest);
    // The side streams: made HERE, next to the context stream, for the first two contexts alive in the process (the runtime maps streams
    // to its hardware queues in creation order, and a latecomer shared one with the context stream); from the third context on — the
    // members of a batch registration's pool, the chains of a many-chains job, all stepped through a launch context's streams — on
    // first use (LazyStream).  ICP_EAGER_STREAMS=1: always here.
    static const bool eager_always = std::getenv("ICP_EAGER_STREAMS") != nullptr;
    const bool now = eager_always || g_live_contexts.load(std::memory_order_relaxed) < 2;
    ctx->front_stream.arm(device, greatest, prio_greatest, false, now);
    ctx->eig_stream.arm(device, greatest, prio_greatest, true, now);
    // (ranks above 64 only: a stream costs a few MB of the runtime's own memory)
    if (ctx->r > 64) ctx->eig_stream2.arm(device, greatest, prio_greatest, false, now);
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_ready, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_side, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_sum, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_asm, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_join, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_inst, hipEventDisableTiming));
    HIP_OK(hipEventCreateWithFlags(&ctx->ev_front, hipEventDisableTiming));
    pinned_alloc((void**)&ctx->h_wait_error, sizeof(int) * 16);
    ctx->h_wait_error[0] = 0;

    mark("streams, events");
    // ---- model and target: the immutable device data is shared between the contexts of a device made from the same arrays
    std::lock_guard<std::mutex> shared_lk(g_shared_mu);
    // Which model this is: a hash of its arrays — 137 MB of basis at the face model's size, 6.6 ms of every context's creation (20
    // contexts of a batch registration: 130 ms) —, or, if the caller vouches for it (model_key != 0: equal keys mean equal arrays,
    // icp_ctx_create_keyed), the key, the small arrays and a few thousand basis values spread over the array
    uint64_t mh = hash_words(0x1234, model->ref_points, sizeof(double) * 3 * N);
    if (model_key == 0) {
      mh = hash_words(mh, model->basis, sizeof(double) * 3 * N * r);
    } else {
      mh = hash_words(mh ^ 0x6b657965646d6f64ull, &model_key, sizeof(model_key));
      const size_t nb = (size_t)3 * N * r, stride = std::max<size_t>(1, nb / 4096);
      std::vector<double> sample;
      sample.reserve(4100);
      for (size_t i = 0; i < nb; i += stride) sample.push_back(model->basis[i]);
      sample.push_back(model->basis[nb - 1]);
      mh = hash_words(mh, sample.data(), sizeof(double) * sample.size());
    }
    mh = hash_words(mh, model->variance, sizeof(double) * r);
    if (model->mean_deformation) mh = hash_words(mh, model->mean_deformation, sizeof(double) * 3 * N);
    mh = hash_words(mh, model->triangles, sizeof(int32_t) * 3 * T);
    mark("model hash");
    const SharedKey mkey{device, N, T, r, mh};
    std::shared_ptr<SharedModel> sm = g_shared_models[mkey].lock();
    if (!sm) {
      sm = std::make_shared<SharedModel>();
      sm->device = device;
      static uint64_t next_uid = 0;  // (under g_shared_mu)
      sm->uid = ++next_uid;
      // Q = Φ·diag(√λ) in two layouts, Gram matrix G = QᵀQ and chol(G + σ²I) (one-off host work)
      std::vector<double> Q((size_t)3 * N * r), Qp((size_t)3 * N * r), sl(r), isl(r);
      for (int j = 0; j < r; ++j) { sl[j] = std::sqrt(model->variance[j]); isl[j] = 1.0 / sl[j]; }
      for (size_t row = 0; row < (size_t)3 * N; ++row)
        for (int j = 0; j < r; ++j) {
          double q = model->basis[row * r + j] * sl[j];
          Q[row * r + j] = q;
          size_t i = row / 3, d = row % 3;
          Qp[((size_t)j * 3 + d) * N + i] = q;
        }
      std::vector<double> G((size_t)r * r, 0.0);
      for (size_t row = 0; row < (size_t)3 * N; ++row) {
        const double* q = &Q[row * r];
        for (int a = 0; a < r; ++a) {
          double qa = q[a];
          double* g = &G[(size_t)a * r];
          for (int b = 0; b <= a; ++b) g[b] += qa * q[b];
        }
      }
      for (int a = 0; a < r; ++a)
        for (int b = a + 1; b < r; ++b) G[(size_t)a * r + b] = G[(size_t)b * r + a];
      std::vector<double> Gs = G, Ginv, Pinv;
      for (int a = 0; a < r; ++a) Gs[(size_t)a * r + a] += kSigma2;
      if (!host_spd_inverse(r, Gs, Pinv)) fail(ICP_ERR_NOT_SPD, "Q^T Q + sigma^2 I is not positive definite");
      if (!host_spd_inverse(r, G, Ginv)) fail(ICP_ERR_NOT_SPD, "Q^T Q is not positive definite (linearly dependent basis functions)");
      std::vector<double> mean((size_t)3 * N, 0.0);
      if (model->mean_deformation) std::memcpy(mean.data(), model->mean_deformation, sizeof(double) * 3 * N);
      std::vector<uint8_t> mb;
      boundary_flags(N, T, model->triangles, mb);
      std::vector<int> off, adj;
      vertex_adjacency(N, T, model->triangles, off, adj);
      sm->n_boundary = (int)std::count(mb.begin(), mb.end(), (uint8_t)1);
      {
        sm->ring_prefix_max.assign((size_t)N, 0);
        for (int v = 0; v < N; ++v) {
          int reach = v;
          for (int a = off[v]; a < off[v + 1]; ++a)
            for (int q = 0; q < 3; ++q) reach = std::max(reach, (int)model->triangles[3 * (size_t)adj[a] + q]);
          sm->ring_prefix_max[v] = v > 0 ? std::max(reach, sm->ring_prefix_max[v - 1]) : reach;
        }
      }
      sm->ref.upload(model->ref_points, (size_t)3 * N);
      sm->mean.upload(mean.data(), mean.size());
      sm->Q.upload(Q.data(), Q.size());
      sm->Qp.upload(Qp.data(), Qp.size());
      sm->sqrt_lambda.upload(sl.data(), r);
      sm->inv_sqrt_lambda.upload(isl.data(), r);
      sm->G.upload(G.data(), G.size());
      sm->Ginv.upload(Ginv.data(), Ginv.size());
      sm->P.upload(Pinv.data(), Pinv.size());
      sm->tris.upload(model->triangles, (size_t)3 * T);
      {
        const std::vector<int> order = coherent_triangle_order(N, T, model->ref_points, model->triangles);
        sm->tri_order.upload(order.data(), order.size());
      }
      sm->adj_off.upload(off.data(), off.size());
      sm->adj.upload(adj.data(), adj.size());
      sm->boundary.upload(mb.data(), mb.size());
      g_shared_models[mkey] = sm;
    }
    ctx->shared_model = sm;
    if (g_model_keep[0] != sm && g_model_keep[1] != sm) {
      std::shared_ptr<SharedModel> evicted = std::move(g_model_keep[g_model_keep_next]);
      g_model_keep[g_model_keep_next] = sm;
      g_model_keep_next ^= 1;
      if (evicted && evicted.use_count() == 1 && evicted->device != device) {  // its last owner: freed under its own device
        (void)hipSetDevice(evicted->device);
        evicted.reset();
        (void)hipSetDevice(device);
      }
    }
    ctx->n_boundary = sm->n_boundary;
    ctx->ref.alias(sm->ref); ctx->mean.alias(sm->mean); ctx->Q.alias(sm->Q); ctx->Qp.alias(sm->Qp);
    ctx->sqrt_lambda.alias(sm->sqrt_lambda); ctx->inv_sqrt_lambda.alias(sm->inv_sqrt_lambda);
    ctx->G.alias(sm->G); ctx->Ginv.alias(sm->Ginv); ctx->P.alias(sm->P);
    ctx->tris.alias(sm->tris); ctx->tri_order.alias(sm->tri_order); ctx->adj_off.alias(sm->adj_off); ctx->adj.alias(sm->adj);
    ctx->boundary.alias(sm->boundary);

    mark("shared model");
    attach_target(ctx, target, device);
    mark("target");

    ctx->hint_surf.alloc(N);
    ctx->hint_nnv.alloc(N);
    seed_context_hints(*ctx, true);  // (the pair's filed hints, or none; g_shared_mu is held here)
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ctx->stage_cap = 64 * (size_t)(10 + r) + 4096;
    pinned_alloc((void**)&ctx->h_stage, sizeof(double) * ctx->stage_cap);
    ctx->d_stage.alloc(ctx->stage_cap);
    const size_t res_cap = std::max<size_t>(2048, 3 * (size_t)N + 64);
    pinned_alloc((void**)&ctx->h_out, sizeof(double) * (icp_ctx::kStatusDoubles + res_cap));
    ctx->d_out.alloc(icp_ctx::kStatusDoubles + res_cap);
    ctx->h_status = (int*)ctx->h_out;
    ctx->h_res = ctx->h_out + icp_ctx::kStatusDoubles;
    ctx->d_status.p = (int*)ctx->d_out.p; ctx->d_status.n = 2 * icp_ctx::kStatusDoubles; ctx->d_status.owned = false;
    ctx->d_res.p = ctx->d_out.p + icp_ctx::kStatusDoubles; ctx->d_res.n = res_cap; ctx->d_res.owned = false;
    pinned_alloc((void**)&ctx->h_flag, sizeof(int) * 16);
    ctx->h_flag[0] = 0;
    ctx->d_done.alloc(4);
    ctx->d_done.fill_bytes(0);
    ctx->d_wait_ticks.alloc(2);
    ctx->d_wait_ticks.fill_bytes(0);
    mark("hints, pinned result areas");
    for (auto& sl : ctx->slots) ctx->alloc_slot(sl);
    HIP_OK(hipStreamSynchronize(ctx->stream));
    mark("state slots");
    ++g_live_contexts;
    ctx->counted = true;
    *out = ctx;
  });
  if (rc != ICP_OK && ctx) {
    icp_ctx_destroy(ctx);
  }
  return rc;
}

void icp_release_cached_models(void) {
  std::lock_guard<std::mutex> lk(g_shared_mu);
  int caller_device = -1;  // (a model's buffers are freed under ITS device; the caller's current device is put back afterwards)
  const bool have_device = hipGetDevice(&caller_device) == hipSuccess;
  for (int i = 0; i < 2; ++i) {
    if (g_model_keep[i]) (void)hipSetDevice(g_model_keep[i]->device);
    g_model_keep[i].reset();
  }
  if (have_device) (void)hipSetDevice(caller_device);
  drain_pools();
}

void icp_ctx_destroy(icp_ctx* ctx) {
  if (!ctx) return;
  (void)hipSetDevice(ctx->device);
  DeviceQuiesce _q;  // (its device buffers go back to the pool: device_free)
  if (ctx->eig_last && ctx->eig_last != ctx->eig_stream.peek()) {  // decompositions of this context on a batch's stream
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(ctx->eig_last)) (void)hipStreamSynchronize(ctx->eig_last);
  }
  if (ctx->eig_last2 && ctx->eig_last2 != ctx->eig_stream2.peek()) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    if (g_eig_streams.count(ctx->eig_last2)) (void)hipStreamSynchronize(ctx->eig_last2);
  }
  for (hipStream_t* pool : {ctx->batch_eig, ctx->batch_eig2})
    for (int k = 0; k < icp_ctx::kBatchRing; ++k)
      if (pool[k]) {
        std::lock_guard<std::mutex> lk(g_eig_streams_mu);
        g_eig_streams.erase(pool[k]);
        give_stream(pool[k]);
        pool[k] = nullptr;
      }
  if (hipStream_t s2 = ctx->eig_stream2.release()) give_stream(s2);
  if (hipStream_t s1 = ctx->eig_stream.release()) {
    std::lock_guard<std::mutex> lk(g_eig_streams_mu);
    g_eig_streams.erase(s1);
    (void)hipStreamSynchronize(s1);
    library_release_stream(s1);
    give_stream(s1);
  }
  if (ctx->ev_ready) (void)hipEventDestroy(ctx->ev_ready);
  if (ctx->ev_side) (void)hipEventDestroy(ctx->ev_side);
  if (ctx->ev_sum) (void)hipEventDestroy(ctx->ev_sum);
  if (ctx->ev_asm) (void)hipEventDestroy(ctx->ev_asm);
  if (hipStream_t fs = ctx->front_stream.release()) {
    (void)hipStreamSynchronize(fs);
    library_release_stream(fs);
    give_stream(fs);
  }
  if (ctx->stream) {
    (void)hipStreamSynchronize(ctx->stream);
    library_release_stream(ctx->stream);
    give_stream(ctx->stream);
  }
  if (ctx->ev_join) (void)hipEventDestroy(ctx->ev_join);
  if (ctx->ev_inst) (void)hipEventDestroy(ctx->ev_inst);
  if (ctx->ev_front) (void)hipEventDestroy(ctx->ev_front);
  if (ctx->h_wait_error) pinned_free(ctx->h_wait_error);
  if (ctx->h_wide_z) pinned_free(ctx->h_wide_z);
  for (void* bp : ctx->wide_pinned)
    if (bp) pinned_free(bp);
  for (hipEvent_t ev : ctx->ev_wide_sum)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_fac)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_head)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : ctx->ev_wide_eval)
    if (ev) (void)hipEventDestroy(ev);
  if (ctx->h_gate_error) pinned_free(ctx->h_gate_error);
  for (void* bp : ctx->batch_eig_rec)
    if (bp) pinned_free(bp);
  g_host_timing.report();
  g_batch_timing.report();
  for (auto& r : ctx->prof.pool) { (void)hipEventDestroy(r.a); (void)hipEventDestroy(r.b); }
  if (ctx->h_stage) pinned_free(ctx->h_stage);
  if (ctx->h_out) pinned_free(ctx->h_out);
  if (ctx->h_flag) pinned_free(ctx->h_flag);
  for (void* bp : ctx->batch_pinned)
    if (bp) pinned_free(bp);
  if (ctx->counted) --g_live_contexts;
  delete ctx;
}

int icp_ctx_set_target(icp_ctx* ctx, const icp_mesh_desc* target) {
  return guard([&] {
    require(ctx && target, "null argument");
    require(target->n_points > 0 && target->n_triangles >= 0 && target->points && (target->triangles || target->n_triangles == 0),
            "target arrays missing");
    check_triangles(target->n_points, target->n_triangles, target->triangles, "target");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    require(ctx->proposals.empty() && ctx->evaluators.empty(), "the context still has proposals or evaluators made for its present target");
    Bound _b(ctx);
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ctx->front_stream.sync();
    sync_eigen(*ctx);
    {
      std::lock_guard<std::mutex> shared_lk(g_shared_mu);
      attach_target(ctx, target, ctx->device);
    }
    // what was cached against the old target: the states' surface points and nearest vertices, the search hints
    for (auto& s : ctx->slots) { s.valid = false; s.defo_valid = false; s.spheres_valid = false; s.n_surf = s.n_nnv = 0; s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0; }
    seed_context_hints(*ctx);  // (the new pair's filed hints, or none)
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ctx->stage_used = 0;
  });
}

namespace { void release_front(StepFront& F); }

int icp_ctx_set_rotation(icp_ctx* ctx, const double* angles, const double* R) {
  return guard([&] {
    require(ctx && angles, "null argument");
    for (int k = 0; k < 3; ++k) require(std::isfinite(angles[k]), "angles must be finite");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    icp_ctx::RotationEntry* slot = nullptr;
    for (auto& e : ctx->rotations)
      if (e.valid && e.angles[0] == angles[0] && e.angles[1] == angles[1] && e.angles[2] == angles[2]) { slot = &e; break; }
    // Everything cached under a theta with these angles was posed with the matrix in force so far: whenever that changes — a first
    // registration (the library's own convention until now), a replacement, a withdrawal, an eviction — the state slots, the
    // posterior memo entries of every proposal and the evaluators' memoised values of such thetas are dropped.  (Pending half steps
    // and decompositions in flight are drained first: they hold such entries.)
    auto forget = [&](const double* a) {
      auto same = [&](const std::vector<double>& th) { return th.size() >= 7 && th[4] == a[0] && th[5] == a[1] && th[6] == a[2]; };
      bool any = false;
      for (auto& sl : ctx->slots) any = any || (sl.valid && same(sl.theta));
      for (icp_proposal* p : ctx->proposals)
        for (int i = 0; i < kPosteriorMemo; ++i) any = any || (p->memo[i].valid && same(p->memo[i].theta));
      for (icp_evaluator* ev : ctx->evaluators) {
        for (auto& m : ev->memo) any = any || (m.valid && same(m.theta));
        any = any || (ev->front.valid && same(ev->front.theta_cur));
      }
      if (!any) return;
      if (ctx->batch_busy) throw IcpError{ICP_ERR_BUSY, "the context belongs to a batch in flight"};
      ctx->bind();
      HIP_OK(hipStreamSynchronize(ctx->stream));
      ctx->front_stream.sync();
      sync_eigen(*ctx);
      for (icp_evaluator* ev : ctx->evaluators) {
        if (ev->front.valid) release_front(ev->front);
        for (auto& m : ev->memo)
          if (m.valid && same(m.theta)) m.valid = false;
        ev->last_prop.clear();
      }
      for (auto& sl : ctx->slots)
        if (sl.valid && same(sl.theta)) sl.valid = false;
      for (icp_proposal* p : ctx->proposals) {
        for (int i = 0; i < kPosteriorMemo; ++i) {
          PosteriorEntry& en = p->memo[i];
          if (en.valid && same(en.theta)) { en.valid = false; en.eig_valid = false; en.eig_checked = false; }
        }
        p->spec_entry = nullptr;
      }
    };
    if (!R) {  // withdraw the entry
      if (slot) { slot->valid = false; forget(angles); }
      return;
    }
    // the matrix must be a rotation (orthonormal to 1e-9, determinant +1): a wrong layout would otherwise pass silently
    for (int a = 0; a < 3; ++a)
      for (int b = 0; b < 3; ++b) {
        double d = 0.0;
        for (int k = 0; k < 3; ++k) d += R[3 * a + k] * R[3 * b + k];
        require(std::fabs(d - (a == b ? 1.0 : 0.0)) <= 1e-9, "R is not orthonormal (row-major 3x3 rotation expected)");
      }
    const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
    require(det > 0.0, "R is a reflection, not a rotation");
    {  // the convention check: does the caller's matrix agree with the library's Rz·Ry·Rx for these angles?
      double own[9];
      icp_rotation_matrix(angles[0], angles[1], angles[2], own);
      double dmax = 0.0;
      for (int k = 0; k < 9; ++k) dmax = std::max(dmax, std::fabs(own[k] - R[k]));
      if (dmax <= icp_ctx::kRotationTol) ++ctx->rotations_verified; else ++ctx->rotations_mismatched;
    }
    if (slot) {  // replacement: only if the matrix really differs
      bool differs = false;
      for (int k = 0; k < 9; ++k) differs = differs || slot->R[k] != R[k];
      if (differs) forget(angles);
    } else {
      slot = &ctx->rotations[0];
      for (auto& e : ctx->rotations) {
        if (!e.valid) { slot = &e; break; }
        if (e.stamp < slot->stamp) slot = &e;
      }
      if (slot->valid) forget(slot->angles);  // eviction: that triple falls back to the library's convention
      forget(angles);
    }
    for (int k = 0; k < 3; ++k) slot->angles[k] = angles[k];
    for (int k = 0; k < 9; ++k) slot->R[k] = R[k];
    slot->valid = true;
    slot->stamp = ++ctx->rotation_clock;
  });
}

int icp_ctx_rotation_convention(icp_ctx* ctx, int64_t* verified, int64_t* mismatched) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  if (verified) *verified = ctx->rotations_verified;
  if (mismatched) *mismatched = ctx->rotations_mismatched;
  return ICP_OK;
}

int icp_ctx_runtime_stats(const icp_ctx* ctx, icp_runtime_stats* out) {
  if (!out) return ICP_ERR_INVALID_ARG;
  const RuntimeStats& s = ctx ? ctx->stats : g_runtime_stats;
  std::memset(out, 0, sizeof(*out));
  out->wait_timeouts = s.wait_timeouts.load();
  out->speculation_giveups = s.speculation_giveups.load();
  out->pipeline_fallbacks = s.pipeline_fallbacks.load();
  out->step_redos = s.step_redos.load();
  out->gate_timeouts = s.gate_timeouts.load();
  return ICP_OK;
}

int icp_ctx_step_paths(const icp_ctx* ctx, int64_t* out) {
  if (!out) return ICP_ERR_INVALID_ARG;
  const StepPaths& p = ctx ? ctx->paths : g_step_paths;
  for (int k = 0; k < 4; ++k) out[k] = p.n[k].load(std::memory_order_relaxed);
  return ICP_OK;
}

int icp_ctx_set_idle_hook(icp_ctx* ctx, icp_idle_fn fn, void* arg) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->idle_fn = fn;
  ctx->idle_arg = fn ? arg : nullptr;
  return ICP_OK;
}

int icp_ctx_profile_start(icp_ctx* ctx, int32_t max_launches) {
  return guard([&] {
    require(ctx && max_launches > 0, "bad argument");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->bind();
    HIP_OK(hipStreamSynchronize(ctx->stream));
    while (ctx->prof.pool.size() < (size_t)max_launches) {
      Profiler::Rec r;
      HIP_OK(hipEventCreate(&r.a));
      HIP_OK(hipEventCreate(&r.b));
      r.id = 0;
      ctx->prof.pool.push_back(r);
    }
    ctx->prof.used = 0;
    ctx->prof.overflow = false;
    ctx->d_wait_ticks.fill_bytes(0);
    if (!ctx->d_search_counters.p) ctx->d_search_counters.alloc(kSearchCounters);
    ctx->d_search_counters.fill_bytes(0);
    ctx->prof.counters = ctx->count_searches ? ctx->d_search_counters.p : nullptr;
    ctx->profiling = true;
  });
}

int icp_ctx_profile_search_counters(icp_ctx* ctx, int32_t on) {
  if (!ctx) return ICP_ERR_INVALID_ARG;
  std::lock_guard<std::recursive_mutex> lk(ctx->mu);
  ctx->count_searches = on != 0;
  return ICP_OK;
}

int icp_ctx_profile_stop(icp_ctx* ctx, icp_kernel_stat* stats, int32_t capacity, int32_t* n_out) {
  return guard([&] {
    require(ctx && stats && n_out && capacity >= KID_COUNT, "bad argument (capacity must be >= 32)");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    ctx->bind();
    HIP_OK(hipStreamSynchronize(ctx->stream));
    ctx->front_stream.sync();
    sync_eigen(*ctx);
    for (hipStream_t bs : ctx->batch_eig)
      if (bs) HIP_OK(hipStreamSynchronize(bs));  // (decompositions of batches this context carried)
    for (hipStream_t bs : ctx->batch_eig2)
      if (bs) HIP_OK(hipStreamSynchronize(bs));
    ctx->profiling = false;
    std::vector<icp_kernel_stat> acc(KID_COUNT);
    for (int i = 0; i < KID_COUNT; ++i) {
      std::memset(&acc[i], 0, sizeof(icp_kernel_stat));
      std::strncpy(acc[i].name, kKernelNames[i], sizeof(acc[i].name) - 1);
      acc[i].min_ms = 1e300;
    }
    for (size_t i = 0; i < ctx->prof.used; ++i) {
      float ms = 0.f;
      HIP_OK(hipEventElapsedTime(&ms, ctx->prof.pool[i].a, ctx->prof.pool[i].b));
      icp_kernel_stat& a = acc[ctx->prof.pool[i].id];
      a.calls++;
      a.total_ms += ms;
      a.min_ms = std::min(a.min_ms, (double)ms);
      a.max_ms = std::max(a.max_ms, (double)ms);
    }
    int n = 0;
    for (int i = 0; i < KID_COUNT; ++i)
      if (acc[i].calls > 0) stats[n++] = acc[i];
    {  // how much of k_step_begin's time was spent waiting ON THE DEVICE for the previous step / the decomposition it draws from
      long long both[2] = {0, 0};
      HIP_OK(hipMemcpy(both, ctx->d_wait_ticks.p, sizeof(both), hipMemcpyDeviceToHost));
      const long long ticks = both[0];
      if (both[1] > 0 && n < capacity) {  // … and of the speculative decompositions' time waiting for their input (EigenSpec::wait_ticks)
        icp_kernel_stat w;
        std::memset(&w, 0, sizeof(w));
        std::strncpy(w.name, "k_posterior_eigen.device_wait", sizeof(w.name) - 1);
        w.calls = acc[KID_EIGEN].calls;
        w.total_ms = (double)both[1] * 1e-5;
        stats[n++] = w;
      }
      if (ticks > 0 && n < capacity) {
        icp_kernel_stat w;
        std::memset(&w, 0, sizeof(w));
        std::strncpy(w.name, "k_step_begin.device_wait", sizeof(w.name) - 1);
        w.calls = acc[KID_STEP_BEGIN].calls;
        w.total_ms = (double)ticks * 1e-5;  // 100 MHz ticks
        stats[n++] = w;
      }
    }
    {  // executed tests of the searches (counted per wave while profiling): rows "count.*", the number in `calls`
      unsigned long long cnt[kSearchCounters] = {};
      if (ctx->d_search_counters.p && ctx->prof.counters) HIP_OK(hipMemcpy(cnt, ctx->d_search_counters.p, sizeof(cnt), hipMemcpyDeviceToHost));
      static const char* names[5] = {"count.surface_ball_tests", "count.surface_sphere_tests", "count.surface_exact_tests",
                                     "count.vertex_filter_tests", "count.vertex_exact_tests"};
      for (int k = 0; k < 5; ++k)
        if (cnt[k] > 0 && n < capacity) {
          icp_kernel_stat w;
          std::memset(&w, 0, sizeof(w));
          std::strncpy(w.name, names[k], sizeof(w.name) - 1);
          w.calls = (int64_t)cnt[k];
          stats[n++] = w;
        }
    }
    *n_out = n;
    if (ctx->prof.overflow) fail(ICP_ERR_INVALID_ARG, "profiler event pool too small: raise max_launches");
  });
}

int icp_transformed_mesh(icp_ctx* ctx, const double* theta, double* points_out) {
  return guard([&] {
    require(ctx && points_out, "null argument");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    HIP_OK(hipMemcpyAsync(points_out, s.x.p, sizeof(double) * 3 * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    ctx->finish(0, 0);
  });
}

int icp_vertex_normals(icp_ctx* ctx, const double* theta, double* normals_out) {
  return guard([&] {
    require(ctx && normals_out, "null argument");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    DBuf<double> nrm;
    nrm.alloc(3 * (size_t)ctx->N);
    launch_vertex_normals(ctx->stream, ctx->N, s.x.p, ctx->tris.p, ctx->adj_off.p, ctx->adj.p, nrm.p);
    HIP_OK(hipMemcpyAsync(normals_out, nrm.p, sizeof(double) * 3 * ctx->N, hipMemcpyDeviceToHost, ctx->stream));
    ctx->finish(0, 0);
  });
}

namespace {
// shared body of the four stand-alone search entry points
void run_search(icp_ctx* ctx, bool surface, int V, int T, const double* verts, const int* tris, const float4* spheres,
                int32_t n, const double* queries, double* points_out, int32_t* index_out, double* dist2_out) {
  require(n >= 0 && (queries || n == 0), "bad query array");
  if (n == 0) return;
  DBuf<double> q, cp, d2;
  DBuf<int> idx;
  q.upload(queries, 3 * (size_t)n);
  cp.alloc(3 * (size_t)n);
  d2.alloc(n);
  idx.alloc(n);
  QueryBuffers qb = ctx->query_scratch(n, surface ? T : V);
  if (surface) launch_surface_query(ctx->stream, T, verts, tris, spheres, n, q.p, nullptr, qb, cp.p, d2.p, idx.p);
  else launch_vertex_query(ctx->stream, V, verts, n, q.p, nullptr, qb, d2.p, idx.p);
  if (points_out && surface) HIP_OK(hipMemcpyAsync(points_out, cp.p, sizeof(double) * 3 * n, hipMemcpyDeviceToHost, ctx->stream));
  if (index_out) HIP_OK(hipMemcpyAsync(index_out, idx.p, sizeof(int) * n, hipMemcpyDeviceToHost, ctx->stream));
  if (dist2_out) HIP_OK(hipMemcpyAsync(dist2_out, d2.p, sizeof(double) * n, hipMemcpyDeviceToHost, ctx->stream));
  ctx->finish(0, 0);
}
}  // namespace

int icp_closest_point_on_target(icp_ctx* ctx, int32_t n, const double* queries, double* points_out, int32_t* triangle_out,
                                double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    const DeviceMesh& t = ctx->target;
    run_search(ctx, true, t.V, t.T, t.verts.p, t.tris.p, t.spheres.p, n, queries, points_out, triangle_out, dist2_out);
  });
}

int icp_closest_target_vertex(icp_ctx* ctx, int32_t n, const double* queries, int32_t* id_out, double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    const DeviceMesh& t = ctx->target;
    run_search(ctx, false, t.V, t.T, t.verts.p, t.tris.p, t.spheres.p, n, queries, nullptr, id_out, dist2_out);
  });
}

int icp_closest_model_vertex(icp_ctx* ctx, const double* theta, int32_t n, const double* queries, int32_t* id_out,
                             double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    run_search(ctx, false, ctx->N, ctx->T, s.x.p, ctx->tris.p, nullptr, n, queries, nullptr, id_out, dist2_out);
  });
}

int icp_closest_point_on_model(icp_ctx* ctx, const double* theta, int32_t n, const double* queries, double* points_out,
                               int32_t* triangle_out, double* dist2_out) {
  return guard([&] {
    require(ctx, "null context");
    check_theta_finite(ctx, theta);
    std::lock_guard<std::recursive_mutex> lk(ctx->mu);
    Bound _b(ctx);
    StateSlot& s = ctx->state(theta);
    ctx->ensure_model_spheres(s);
    run_search(ctx, true, ctx->N, ctx->T, s.x.p, ctx->tris.p, s.spheres.p, n, queries, points_out, triangle_out, dist2_out);
  });
}
}  // extern "C"
