// kernels_geometry.hip — instance synthesis and the two brute-force searches (gfx950).
//
// Search design (both K3 point×vertex and K4 point×triangle):
//   * the SEARCHED set (vertices / triangle bounding spheres) is streamed from HBM exactly once per launch:
//     one element per lane, coalesced, held in VGPRs for the whole kernel;
//   * the QUERY batch is tiny (≤ a few thousand points) and wave-uniform, so it is read through the scalar
//     unit (s_load, scalar cache) — no LDS staging or barriers are needed and VALU ops take it as SGPR operands;
//   * every (query, element) pair is visited (brute force), but the expensive exact test runs only where a
//     conservative bound cannot rule the element out.  The bound comes from the previous winner of the same
//     query (`hint`): any element yields a valid upper bound, so a stale hint costs time, never correctness;
//   * the exact minimum is taken with a 64-bit integer atomicMin on the bit pattern of the squared distance
//     (pass A); the lowest index attaining exactly that value is taken with a second atomicMin (pass B).
//     Result = lexicographic min over (d², index) of ALL elements — identical to a sequential scan with `<`.
#include "icp_kernels.hpp"

namespace icp {

thread_local Profiler* g_prof = nullptr;
const char* const kKernelNames[KID_COUNT] = {
    "k_instance", "k_surface_init", "k_surface_pass<0>", "k_bound_from_best", "k_surface_pass<1>", "k_surface_final",
    "k_vertex_init", "k_vertex_pass<0>", "k_vertex_pass<1>", "k_vertex_final", "k_tri_spheres", "k_correspond",
    "k_regression", "k_posterior_factor", "k_transition_tail", "k_posterior_eigen", "k_propose", "k_reduce"};

void Profiler::begin(hipStream_t st, int id) {
  if (used >= pool.size()) { overflow = true; return; }
  pool[used].id = id;
  (void)hipEventRecord(pool[used].a, st);
}
void Profiler::end(hipStream_t st) {
  if (used >= pool.size()) return;
  (void)hipEventRecord(pool[used].b, st);
  ++used;
}

namespace {

constexpr int kBlock = 256;
constexpr double kSlack = 1.0 + 1e-9;     // relative slack on the squared pruning bound
constexpr double kThrInfl = 1.0 + 1e-12;  // inflation of sqrt(d²) bounds

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- K1 instance

__global__ void __launch_bounds__(kBlock) k_instance(int N, int r, const double* __restrict__ Qp,
                                                      const double* __restrict__ ref, const double* __restrict__ mean,
                                                      Pose pose, const double* __restrict__ coeffs, double* __restrict__ x) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  double a0 = mean[3 * i], a1 = mean[3 * i + 1], a2 = mean[3 * i + 2];
  const double* q = Qp + i;
  for (int j = 0; j < r; ++j) {
    double c = coeffs[j];
    a0 = a0 + q[(size_t)(3 * j) * N] * c;
    a1 = a1 + q[(size_t)(3 * j + 1) * N] * c;
    a2 = a2 + q[(size_t)(3 * j + 2) * N] * c;
  }
  double u0 = ref[3 * i] + a0, u1 = ref[3 * i + 1] + a1, u2 = ref[3 * i + 2] + a2;
  double v0 = u0 - pose.ctr[0], v1 = u1 - pose.ctr[1], v2 = u2 - pose.ctr[2];
  double w0 = (pose.R[0] * v0 + pose.R[1] * v1) + pose.R[2] * v2;
  double w1 = (pose.R[3] * v0 + pose.R[4] * v1) + pose.R[5] * v2;
  double w2 = (pose.R[6] * v0 + pose.R[7] * v1) + pose.R[8] * v2;
  x[3 * i] = pose.s * ((w0 + pose.ctr[0]) + pose.t[0]);
  x[3 * i + 1] = pose.s * ((w1 + pose.ctr[1]) + pose.t[1]);
  x[3 * i + 2] = pose.s * ((w2 + pose.ctr[2]) + pose.t[2]);
}

__global__ void __launch_bounds__(kBlock) k_vertex_normals(int N, const double* __restrict__ x, const int* __restrict__ tris,
                                                            const int* __restrict__ adj_off, const int* __restrict__ adj,
                                                            double* __restrict__ normals) {
  int v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= N) return;
  d3 n = vertex_normal(x, tris, adj_off, adj, v);
  normals[3 * v] = n.x; normals[3 * v + 1] = n.y; normals[3 * v + 2] = n.z;
}

__global__ void __launch_bounds__(kBlock) k_tri_spheres(int T, const double* __restrict__ verts, const int* __restrict__ tris,
                                                         double4* __restrict__ spheres) {
  int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= T) return;
  d3 a = ld3(verts + 3 * tris[3 * t]), b = ld3(verts + 3 * tris[3 * t + 1]), c = ld3(verts + 3 * tris[3 * t + 2]);
  d3 m = {(a.x + b.x + c.x) / 3.0, (a.y + b.y + c.y) / 3.0, (a.z + b.z + c.z) / 3.0};
  d3 da = sub(a, m), db = sub(b, m), dc = sub(c, m);
  double r2 = fmax(dot(da, da), fmax(dot(db, db), dot(dc, dc)));
  // inflate: relative for the sqrt/arith rounding, absolute for the rounding of the centre itself
  double R = sqrt(r2) * (1.0 + 1e-10) + 1e-12 * (fabs(m.x) + fabs(m.y) + fabs(m.z));
  spheres[t] = make_double4(m.x, m.y, m.z, R);
}

// ---------------------------------------------------------------- K4 closest point on surface

__global__ void __launch_bounds__(kBlock) k_surface_init(int K, const double* __restrict__ P, int T,
                                                          const double* __restrict__ verts, const int* __restrict__ tris,
                                                          const int* __restrict__ hint, double* __restrict__ thr,
                                                          unsigned long long* __restrict__ best_d2, int* __restrict__ best_idx) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int h = hint ? hint[k] : -1;
  double d2 = __builtin_inf();
  if (h >= 0 && h < T) d2 = tri_dist2(ld3(P + 3 * k), verts, tris, h, nullptr);
  if (!(d2 == d2)) d2 = __builtin_inf();  // degenerate hint triangle
  best_d2[k] = d2bits(d2);
  best_idx[k] = kNoIndex;
  thr[k] = sqrt(d2) * kThrInfl;
}

__global__ void __launch_bounds__(kBlock) k_bound_from_best(int K, const unsigned long long* __restrict__ best_d2,
                                                             double* __restrict__ thr) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  thr[k] = sqrt(bits2d(best_d2[k])) * kThrInfl;
}

// PASS 0 (A): exact minimum squared distance.  PASS 1 (B): lowest triangle index attaining it.
template <int PASS>
__global__ void __launch_bounds__(kBlock) k_surface_pass(int T, const double4* __restrict__ spheres,
                                                          const double* __restrict__ verts, const int* __restrict__ tris,
                                                          int K, int kchunk, const double* __restrict__ P,
                                                          const double* __restrict__ thr,
                                                          const unsigned long long* __restrict__ ref_d2,
                                                          unsigned long long* __restrict__ best_d2,
                                                          int* __restrict__ best_idx) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  double cx = 0.0, cy = 0.0, cz = 0.0, R = -__builtin_inf();  // out-of-range lanes can never pass the bound
  if (t < T) {
    double4 s = spheres[t];
    cx = s.x; cy = s.y; cz = s.z; R = s.w;
  }
  const int k0 = blockIdx.y * kchunk;
  const int k1 = min(K, k0 + kchunk);
  for (int k = k0; k < k1; ++k) {
    const double px = P[3 * k], py = P[3 * k + 1], pz = P[3 * k + 2];  // wave-uniform: scalar loads
    const double tt = thr[k] + R;
    const double dx = px - cx, dy = py - cy, dz = pz - cz;
    const double dc2 = fma(dz, dz, fma(dy, dy, dx * dx));
    if (tt >= 0.0 && dc2 <= tt * tt * kSlack) {
      d3 p = {px, py, pz};
      double d2 = tri_dist2(p, verts, tris, t, nullptr);
      if (d2 == d2) {
        if (PASS == 0) {
          atomicMin(&best_d2[k], d2bits(d2));
        } else {
          if (d2bits(d2) == ref_d2[k]) atomicMin(&best_idx[k], t);
        }
      }
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_surface_final(int K, const double* __restrict__ P,
                                                           const double* __restrict__ verts, const int* __restrict__ tris,
                                                           const unsigned long long* __restrict__ best_d2,
                                                           const int* __restrict__ best_idx, int* __restrict__ hint,
                                                           double* __restrict__ cp, double* __restrict__ d2out,
                                                           int* __restrict__ triout) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int t = best_idx[k];
  d3 c = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
  if (t != kNoIndex) tri_dist2(ld3(P + 3 * k), verts, tris, t, &c);
  if (cp) { cp[3 * k] = c.x; cp[3 * k + 1] = c.y; cp[3 * k + 2] = c.z; }
  if (d2out) d2out[k] = bits2d(best_d2[k]);
  if (triout) triout[k] = (t == kNoIndex) ? -1 : t;
  if (hint) hint[k] = (t == kNoIndex) ? -1 : t;
}

// ---------------------------------------------------------------- K3 nearest vertex

__global__ void __launch_bounds__(kBlock) k_vertex_init(int K, const double* __restrict__ P, int V,
                                                         const double* __restrict__ verts, const int* __restrict__ hint,
                                                         double* __restrict__ thr, unsigned long long* __restrict__ best_d2,
                                                         int* __restrict__ best_idx) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int h = hint ? hint[k] : -1;
  double d2 = __builtin_inf();
  if (h >= 0 && h < V) {
    d3 d = sub(ld3(P + 3 * k), ld3(verts + 3 * h));
    d2 = dot(d, d);
  }
  if (!(d2 == d2)) d2 = __builtin_inf();
  best_d2[k] = d2bits(d2);
  best_idx[k] = kNoIndex;
  thr[k] = d2;  // squared bound, same expression as the pass -> the hint vertex itself always passes
}

template <int PASS>
__global__ void __launch_bounds__(kBlock) k_vertex_pass(int V, const double* __restrict__ verts, int K, int kchunk,
                                                         const double* __restrict__ P, const double* __restrict__ thr,
                                                         const unsigned long long* __restrict__ ref_d2,
                                                         unsigned long long* __restrict__ best_d2,
                                                         int* __restrict__ best_idx) {
  const int v = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = v < V;
  d3 q = {0.0, 0.0, 0.0};
  if (valid) q = ld3(verts + 3 * v);
  const int k0 = blockIdx.y * kchunk;
  const int k1 = min(K, k0 + kchunk);
  for (int k = k0; k < k1; ++k) {
    d3 p = {P[3 * k], P[3 * k + 1], P[3 * k + 2]};  // wave-uniform
    d3 d = sub(p, q);
    double d2 = dot(d, d);  // (dx·dx + dy·dy) + dz·dz, unfused — the value the argmin is defined on
    if (PASS == 0) {
      if (valid && d2 <= thr[k]) atomicMin(&best_d2[k], d2bits(d2));
    } else {
      if (valid && d2bits(d2) == ref_d2[k]) atomicMin(&best_idx[k], v);
    }
  }
}

__global__ void __launch_bounds__(kBlock) k_vertex_final(int K, const unsigned long long* __restrict__ best_d2,
                                                          const int* __restrict__ best_idx, int* __restrict__ hint,
                                                          double* __restrict__ d2out, int* __restrict__ idxout) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int v = best_idx[k];
  if (d2out) d2out[k] = bits2d(best_d2[k]);
  if (idxout) idxout[k] = (v == kNoIndex) ? -1 : v;
  if (hint) hint[k] = (v == kNoIndex) ? -1 : v;
}

// enough waves to fill 256 CUs, but at least ~8 queries per wave so the element load is amortised
inline void split_queries(int n_elem_blocks, int K, int* ksplit, int* kchunk) {
  int want = cdiv(4096, n_elem_blocks * (kBlock / 64));
  int s = want < 1 ? 1 : want;
  int maxs = cdiv(K, 8);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  *kchunk = cdiv(K, s);
  *ksplit = cdiv(K, *kchunk);
}

}  // namespace

void launch_instance(hipStream_t st, int N, int r, const double* Qp, const double* ref, const double* mean,
                     const Pose& pose, const double* coeffs, double* x) {
  { ProfScope _ps(st, KID_INSTANCE);
    hipLaunchKernelGGL(k_instance, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, r, Qp, ref, mean, pose, coeffs, x); }
}

void launch_vertex_normals(hipStream_t st, int N, const double* x, const int* tris, const int* adj_off,
                           const int* adj, double* normals) {
  hipLaunchKernelGGL(k_vertex_normals, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, x, tris, adj_off, adj, normals);
}

void launch_tri_spheres(hipStream_t st, int T, const double* verts, const int* tris, double4* spheres) {
  if (T <= 0) return;
  { ProfScope _ps(st, KID_TRI_SPHERES);
    hipLaunchKernelGGL(k_tri_spheres, dim3(cdiv(T, kBlock)), dim3(kBlock), 0, st, T, verts, tris, spheres); }
}

void launch_surface_query(hipStream_t st, int T, const double* verts, const int* tris, const double4* spheres,
                          int K, const double* P, int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri) {
  if (K <= 0) return;
  const int qblocks = cdiv(K, kBlock);
  { ProfScope _ps(st, KID_SURFACE_INIT);
    hipLaunchKernelGGL(k_surface_init, dim3(qblocks), dim3(kBlock), 0, st, K, P, T, verts, tris, hint, qb.thr,
                     qb.best_d2, qb.best_idx); }
  if (T > 0) {
    const int tblocks = cdiv(T, kBlock);
    int ksplit, kchunk;
    split_queries(tblocks, K, &ksplit, &kchunk);
    { ProfScope _ps(st, KID_SURFACE_PASS_A);
      hipLaunchKernelGGL(k_surface_pass<0>, dim3(tblocks, ksplit), dim3(kBlock), 0, st, T, spheres, verts, tris, K,
                       kchunk, P, qb.thr, (const unsigned long long*)nullptr, qb.best_d2, (int*)nullptr); }
    { ProfScope _ps(st, KID_SURFACE_BOUND);
      hipLaunchKernelGGL(k_bound_from_best, dim3(qblocks), dim3(kBlock), 0, st, K, qb.best_d2, qb.thr); }
    { ProfScope _ps(st, KID_SURFACE_PASS_B);
      hipLaunchKernelGGL(k_surface_pass<1>, dim3(tblocks, ksplit), dim3(kBlock), 0, st, T, spheres, verts, tris, K,
                       kchunk, P, qb.thr, qb.best_d2, (unsigned long long*)nullptr, qb.best_idx); }
  }
  { ProfScope _ps(st, KID_SURFACE_FINAL);
    hipLaunchKernelGGL(k_surface_final, dim3(qblocks), dim3(kBlock), 0, st, K, P, verts, tris, qb.best_d2, qb.best_idx,
                     hint, cp, d2, tri); }
}

void launch_vertex_query(hipStream_t st, int V, const double* verts, int K, const double* P, int* hint,
                         const QueryBuffers& qb, double* d2, int* idx) {
  if (K <= 0) return;
  const int qblocks = cdiv(K, kBlock);
  { ProfScope _ps(st, KID_VERTEX_INIT);
    hipLaunchKernelGGL(k_vertex_init, dim3(qblocks), dim3(kBlock), 0, st, K, P, V, verts, hint, qb.thr, qb.best_d2,
                     qb.best_idx); }
  if (V > 0) {
    const int vblocks = cdiv(V, kBlock);
    int ksplit, kchunk;
    split_queries(vblocks, K, &ksplit, &kchunk);
    { ProfScope _ps(st, KID_VERTEX_PASS_A);
      hipLaunchKernelGGL(k_vertex_pass<0>, dim3(vblocks, ksplit), dim3(kBlock), 0, st, V, verts, K, kchunk, P, qb.thr,
                       (const unsigned long long*)nullptr, qb.best_d2, (int*)nullptr); }
    { ProfScope _ps(st, KID_VERTEX_PASS_B);
      hipLaunchKernelGGL(k_vertex_pass<1>, dim3(vblocks, ksplit), dim3(kBlock), 0, st, V, verts, K, kchunk, P, qb.thr,
                       qb.best_d2, (unsigned long long*)nullptr, qb.best_idx); }
  }
  { ProfScope _ps(st, KID_VERTEX_FINAL);
    hipLaunchKernelGGL(k_vertex_final, dim3(qblocks), dim3(kBlock), 0, st, K, qb.best_d2, qb.best_idx, hint, d2, idx); }
}

}  // namespace icp
