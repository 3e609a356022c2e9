// abi_hints.inl — part of icp_abi.hip (one translation unit; included there, in order).
// search hints handed from the first evaluation against a (model, target) pair to the contexts and evaluators that follow (HintSeed)
namespace {

HintSeed* find_seed(icp_ctx& c) {  // (g_shared_mu held)
  if (!c.shared_target || !c.shared_model) return nullptr;
  for (auto& s : c.shared_target->seeds)
    if (s->model_uid == c.shared_model->uid) return s.get();
  return nullptr;
}

// a context that has just been given its target: start from the pair's filed hints, if any (else: none, 0xFF); enqueued on the context stream
void seed_context_hints(icp_ctx& c, bool shared_mu_held = false) {
  std::unique_lock<std::mutex> lk(g_shared_mu, std::defer_lock);
  if (!shared_mu_held) lk.lock();
  HintSeed* s = find_seed(c);
  if (s && s->surf.p) {
    HIP_OK(hipMemcpyAsync(c.hint_surf.p, s->surf.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
    HIP_OK(hipMemcpyAsync(c.hint_nnv.p, s->nnv.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
    c.hints_filed = true;
  } else {
    HIP_OK(hipMemsetAsync(c.hint_surf.p, 0xFF, sizeof(int) * c.N, c.stream));
    HIP_OK(hipMemsetAsync(c.hint_nnv.p, 0xFF, sizeof(int) * c.N, c.stream));
    c.hints_filed = false;
  }
}

void seed_evaluator_hints(icp_evaluator* ev) {
  icp_ctx& c = *ev->ctx;
  const size_t Ka = std::max(ev->Kt, 1);
  std::lock_guard<std::mutex> lk(g_shared_mu);
  HintSeed* s = find_seed(c);
  if (s && ev->Kt > 0 && ev->points_hash)
    for (auto& e : s->evals)
      if (e->Kt == ev->Kt && e->points_hash == ev->points_hash) {
        HIP_OK(hipMemcpyAsync(ev->hint_tri.p, e->tri.p, sizeof(int) * Ka, hipMemcpyDeviceToDevice, c.stream));
        HIP_OK(hipMemcpyAsync(ev->hint_nnv.p, e->nnv.p, sizeof(int) * Ka, hipMemcpyDeviceToDevice, c.stream));
        ev->hints_filed = true;
        return;
      }
  HIP_OK(hipMemsetAsync(ev->hint_tri.p, 0xFF, sizeof(int) * Ka, c.stream));
  HIP_OK(hipMemsetAsync(ev->hint_nnv.p, 0xFF, sizeof(int) * Ka, c.stream));
  ev->hints_filed = false;
}

// in front of an evaluation: a context / evaluator that has neither filed hints nor been given any takes the pair's, if somebody has
// filed them meanwhile (a batch registration makes its contexts and hands them their target BEFORE the first chain's initial
// evaluation: at that time there was nothing to copy)
void adopt_hints(icp_evaluator* ev) {
  icp_ctx& c = *ev->ctx;
  if (c.hints_filed && ev->hints_filed) return;
  std::lock_guard<std::mutex> lk(g_shared_mu);
  HintSeed* s = find_seed(c);
  if (!s) return;
  if (!c.hints_filed && s->surf.p) {
    HIP_OK(hipMemcpyAsync(c.hint_surf.p, s->surf.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
    HIP_OK(hipMemcpyAsync(c.hint_nnv.p, s->nnv.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
    c.hints_filed = true;
  }
  if (!ev->hints_filed && ev->Kt > 0 && ev->points_hash)
    for (auto& e : s->evals)
      if (e->Kt == ev->Kt && e->points_hash == ev->points_hash) {
        HIP_OK(hipMemcpyAsync(ev->hint_tri.p, e->tri.p, sizeof(int) * ev->Kt, hipMemcpyDeviceToDevice, c.stream));
        HIP_OK(hipMemcpyAsync(ev->hint_nnv.p, e->nnv.p, sizeof(int) * ev->Kt, hipMemcpyDeviceToDevice, c.stream));
        ev->hints_filed = true;
        break;
      }
}

// behind an evaluation that has COMPLETED on the context stream (the caller has synchronised): what its searches left as hints is
// filed for the pair, once
void file_hints(icp_evaluator* ev) {
  icp_ctx& c = *ev->ctx;
  if (c.hints_filed && ev->hints_filed) return;
  std::lock_guard<std::mutex> lk(g_shared_mu);
  if (!c.shared_target || !c.shared_model) return;
  HintSeed* s = find_seed(c);
  if (!s) {
    c.shared_target->seeds.emplace_back(new HintSeed());
    s = c.shared_target->seeds.back().get();
    s->model_uid = c.shared_model->uid;
  }
  bool copied = false;
  if (!c.hints_filed) {
    if (!s->surf.p) {
      s->surf.alloc(c.N); s->nnv.alloc(c.N);
      HIP_OK(hipMemcpyAsync(s->surf.p, c.hint_surf.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
      HIP_OK(hipMemcpyAsync(s->nnv.p, c.hint_nnv.p, sizeof(int) * c.N, hipMemcpyDeviceToDevice, c.stream));
      copied = true;
    }
    c.hints_filed = true;
  }
  if (!ev->hints_filed) {
    if (ev->Kt > 0 && ev->points_hash) {
      bool have = false;
      for (auto& e : s->evals) have = have || (e->Kt == ev->Kt && e->points_hash == ev->points_hash);
      if (!have) {
        std::unique_ptr<HintSeed::Eval> e(new HintSeed::Eval());
        e->Kt = ev->Kt; e->points_hash = ev->points_hash;
        e->tri.alloc(ev->Kt); e->nnv.alloc(ev->Kt);
        HIP_OK(hipMemcpyAsync(e->tri.p, ev->hint_tri.p, sizeof(int) * ev->Kt, hipMemcpyDeviceToDevice, c.stream));
        HIP_OK(hipMemcpyAsync(e->nnv.p, ev->hint_nnv.p, sizeof(int) * ev->Kt, hipMemcpyDeviceToDevice, c.stream));
        s->evals.push_back(std::move(e));
        copied = true;
      }
    }
    ev->hints_filed = true;
  }
  if (copied) HIP_OK(hipStreamSynchronize(c.stream));  // (filed = complete: another context may copy from it at once)
}

}  // namespace
