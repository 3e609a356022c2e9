// abi_state.inl — part of icp_abi.hip (one translation unit; included there, in order).
// icp_ctx: state slots (instances) and the search prefixes shared by proposals and evaluators
Pose icp_ctx::pose_of(const double* theta) {
  Pose p = pose_from_theta(theta);
  for (auto& e : rotations)
    if (e.valid && e.angles[0] == theta[4] && e.angles[1] == theta[5] && e.angles[2] == theta[6]) {
      for (int k = 0; k < 9; ++k) p.R[k] = e.R[k];
      e.stamp = ++rotation_clock;
      break;
    }
  return p;
}

StateSlot* icp_ctx::find_state(const double* theta) {
  const size_t P = 10 + (size_t)r;
  for (auto& s : slots)
    if (s.valid && std::memcmp(s.theta.data(), theta, sizeof(double) * P) == 0) return &s;
  return nullptr;
}

// least recently used slot, emptied (buffers allocated on first use); the caller fills it and sets `valid`
// device buffers of a state slot (all slots at context creation: an allocation is a synchronising runtime call of 50-100 µs,
// which a chain's first steps would otherwise pay one slot at a time)
void icp_ctx::alloc_slot(StateSlot& s) {
  if (s.x.p) return;
  s.coeffs.alloc(r);
  s.x.alloc(3 * (size_t)N);
  s.defo.alloc(3 * (size_t)N);
  s.spheres.alloc(sphere_floats4(T));
  s.surf_cp.alloc(3 * (size_t)N);
  s.surf_d2.alloc(N);
  s.surf_tri.alloc(N);
  s.surf_nnv.alloc(N);
}

StateSlot& icp_ctx::fresh_state() {
  StateSlot* lru = nullptr;
  for (auto& s : slots) {
    if (s.reserved) continue;
    if (!lru) { lru = &s; continue; }
    if (!s.valid) { if (lru->valid) lru = &s; }
    else if (lru->valid && s.stamp < lru->stamp) lru = &s;
  }
  if (!lru) fail(ICP_ERR_DEVICE, "internal: every state slot is reserved");
  StateSlot& s = *lru;
  alloc_slot(s);
  s.valid = false;
  s.defo_valid = false;
  s.spheres_valid = false;
  s.n_surf = 0;
  s.n_nnv = 0;
  s.lo_surf = s.hi_surf = s.lo_nnv = s.hi_nnv = 0;
  return s;
}

StateSlot& icp_ctx::state(const double* theta) {
  const size_t P = 10 + (size_t)r;
  if (StateSlot* hit = find_state(theta)) {
    hit->stamp = ++clock;
    return *hit;
  }
  // a pose move (PoseProposals.scala: 0.4 of the configs[3]/[4] mixture) leaves the coefficients alone: the points are the kept
  // deformations of a state with the same coefficients under the new pose — 0.7 MB instead of the basis' 137 MB at N = 28,561, rank 200
  StateSlot* same = nullptr;
  for (auto& o : slots)
    if (o.valid && o.defo_valid && std::memcmp(o.theta.data() + 10, theta + 10, sizeof(double) * r) == 0) { same = &o; break; }
  if (same) same->stamp = ++clock;  // (not the one recycled below, unless every other slot is reserved: in place works, too)
  StateSlot& s = fresh_state();
  s.theta.assign(theta, theta + P);
  s.valid = true;
  s.stamp = ++clock;
  s.pose = pose_of(theta);
  stage_to(s.coeffs.p, theta + 10, r);
  if (same) launch_instance_pose(stream, N, ref.p, s.pose, same->defo.p, s.x.p, s.defo.p);
  else launch_instance_keep(stream, N, r, Qp.p, ref.p, mean.p, s.pose, s.coeffs.p, s.x.p, s.defo.p);  // ModelFittingParameters.scala:108-110
  s.defo_valid = true;
  return s;
}

void icp_ctx::ensure_model_spheres(StateSlot& s) {
  if (s.spheres_valid) return;
  launch_tri_spheres(stream, T, s.x.p, tris.p, tri_order.p, s.spheres.p);
  s.spheres_valid = true;
}

// target.operations.closestPointOnSurface(currentMesh.point(id)) for id in [0, K) (NonRigidIcpProposal.scala:96-97,
// IndependentPointDistanceEvaluator.scala:41-43): shared by every proposal / evaluator of this context.
void icp_ctx::ensure_surface_prefix(StateSlot& s, int K, hipStream_t st, int which, int reserve) {
  if (K > N) fail(ICP_ERR_INVALID_ARG, "model id count exceeds the number of model points");
  auto join = [&] { if (s.hi_surf > 0 && s.n_surf >= s.lo_surf) { s.n_surf = std::max(s.n_surf, s.hi_surf); s.lo_surf = s.hi_surf = 0; } };
  auto search = [&](int k0, int k1) {
    const int n = k1 - k0;
    QueryBuffers qb = query_scratch(n, target.T, which);
    launch_surface_query(st ? st : stream, target.T, target.verts.p, target.tris.p, target.spheres.p, n, s.x.p + 3 * (size_t)k0,
                         hint_surf.p + k0, qb, s.surf_cp.p + 3 * (size_t)k0, s.surf_d2.p + k0, s.surf_tri.p + k0);
  };
  join();
  if (K <= s.n_surf) return;
  if (reserve > s.n_surf) {  // ids [prefix, reserve) are somebody else's: [reserve, K) detached (once)
    if (s.hi_surf == 0 && reserve < K) { search(reserve, K); s.lo_surf = reserve; s.hi_surf = K; }
    return;
  }
  while (s.n_surf < K) {
    const int k1 = s.hi_surf > 0 ? std::min(K, s.lo_surf) : K;  // (up to a detached range, which then joins)
    if (k1 > s.n_surf) search(s.n_surf, k1);
    s.n_surf = std::max(s.n_surf, k1);
    join();
  }
}

// target.pointSet.findClosestPoint(targetPoint).id (NonRigidIcpProposal.scala:98)
void icp_ctx::ensure_nnv_prefix(StateSlot& s, int K, hipStream_t st, int which, int reserve) {
  ensure_surface_prefix(s, K, st, which, reserve);
  auto join = [&] { if (s.hi_nnv > 0 && s.n_nnv >= s.lo_nnv) { s.n_nnv = std::max(s.n_nnv, s.hi_nnv); s.lo_nnv = s.hi_nnv = 0; } };
  auto search = [&](int k0, int k1) {
    const int n = k1 - k0;
    QueryBuffers qb = query_scratch(n, target.V, which);
    launch_vertex_query(st ? st : stream, target.V, target.verts.p, n, s.surf_cp.p + 3 * (size_t)k0, hint_nnv.p + k0, qb, nullptr,
                        s.surf_nnv.p + k0);
  };
  join();
  if (K <= s.n_nnv) return;
  if (reserve > s.n_nnv) {
    if (s.hi_nnv == 0 && reserve < K) { search(reserve, K); s.lo_nnv = reserve; s.hi_nnv = K; }
    return;
  }
  while (s.n_nnv < K) {
    const int k1 = s.hi_nnv > 0 ? std::min(K, s.lo_nnv) : K;
    if (k1 > s.n_nnv) search(s.n_nnv, k1);
    s.n_nnv = std::max(s.n_nnv, k1);
    join();
  }
}
