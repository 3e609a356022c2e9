// kernels_geometry.hip — instance synthesis and the two brute-force searches (gfx950).
//
// Search design (both K3 point×vertex and K4 point×triangle) — brute force, three launches per query batch:
//   init    per query: exact distance to the previous winner of that query (`hint`) = a valid upper bound (ANY element
//           of the searched set gives one, so a stale hint costs time, never correctness); resets the candidate counter.
//   filter  the SEARCHED set (vertices / triangle bounding spheres) is streamed from HBM exactly once: one element per
//           lane, coalesced, held in VGPRs for the whole kernel.  The query batch is tiny and wave-uniform, so it is
//           read through the scalar unit (s_load, 4 queries per unrolled iteration) and used as SGPR operands — no LDS
//           staging, no barriers.  Every (query, element) pair is visited; an element that the conservative bound cannot
//           rule out is appended to that query's candidate list (one wave-aggregated atomic per wave and query).
//           Triangles: bounding-sphere test in f32 (full-rate VALU) with radii/thresholds inflated so that f32 rounding
//           can only ADD candidates.  Vertices: the exact f64 squared distance itself is the test.
//   resolve one wave per query: exact f64 distance (Ericson point–triangle / squared vertex distance) of every candidate,
//           lexicographic minimum over (d², index) by wave shuffles, closest point of the winner, hint update.
// The result is the lexicographic minimum over ALL elements — identical to a sequential scan with `<` — because the
// true minimiser always survives the filter and the reduction is order independent.
#include "icp_kernels.hpp"

namespace icp {

thread_local Profiler* g_prof = nullptr;
const char* const kKernelNames[KID_COUNT] = {
    "k_instance", "k_surface_init", "k_surface_filter", "k_surface_resolve",
    "k_vertex_init", "k_vertex_filter", "k_vertex_resolve", "k_tri_spheres", "k_correspond",
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
constexpr double kAbsSlack = 3.0 / 8388608.0;  // 3·2^-23 per unit of |coordinate|: covers rounding a point to f32
constexpr int kQU = 4;                         // queries per unrolled iteration of the surface pass

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

__device__ __forceinline__ float round_up_f32(double v) { return nextafterf((float)v, __builtin_inff()); }

// bounding sphere of every triangle for the f32 pruning test: centre = centroid rounded to f32, radius = max corner
// distance (f64) inflated by the rounding of centre and arithmetic, rounded up
__global__ void __launch_bounds__(kBlock) k_tri_spheres(int T, const double* __restrict__ verts, const int* __restrict__ tris,
                                                         float4* __restrict__ spheres) {
  int t = blockIdx.x * kBlock + threadIdx.x;
  if (t >= T) return;
  d3 a = ld3(verts + 3 * tris[3 * t]), b = ld3(verts + 3 * tris[3 * t + 1]), c = ld3(verts + 3 * tris[3 * t + 2]);
  d3 m = {(a.x + b.x + c.x) / 3.0, (a.y + b.y + c.y) / 3.0, (a.z + b.z + c.z) / 3.0};
  d3 da = sub(a, m), db = sub(b, m), dc = sub(c, m);
  double r2 = fmax(dot(da, da), fmax(dot(db, db), dot(dc, dc)));
  double R = sqrt(r2) * (1.0 + 2e-6) + kAbsSlack * (fabs(m.x) + fabs(m.y) + fabs(m.z));
  spheres[t] = make_float4((float)m.x, (float)m.y, (float)m.z, round_up_f32(R));
}

// ---------------------------------------------------------------- K4 closest point on surface

// ---- wave helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// append `value` to list[base_index .. ] for the lanes with `hit`, one atomic per wave; `m` = ballot of hit (non-zero)
__device__ __forceinline__ void wave_append(unsigned long long m, bool hit, int* __restrict__ counter, int* __restrict__ list, int value) {
  const int leader = __ffsll((long long)m) - 1;
  int base = 0;
  if (lane_id() == leader) base = atomicAdd(counter, __popcll(m));
  base = __shfl(base, leader, 64);
  if (hit) {
    const unsigned long long below = m & ((1ull << lane_id()) - 1ull);
    list[base + __popcll(below)] = value;
  }
}

// lexicographic (d², index) minimum across the wave
__device__ __forceinline__ void wave_lexmin(double& d2, int& idx) {
  for (int o = 32; o > 0; o >>= 1) {
    const double od = __shfl_xor(d2, o, 64);
    const int oi = __shfl_xor(idx, o, 64);
    if (od < d2 || (od == d2 && oi < idx)) { d2 = od; idx = oi; }
  }
}

// ---------------------------------------------------------------- K4 closest point on surface

// per query: exact distance to the hinted triangle -> filter bound; f32 copy of the query; zero candidate counter.
// Entries K..Kpad-1 are sentinels (a point at 1e30 with bound 0) so the filter can run unrolled without guards.
__global__ void __launch_bounds__(kBlock) k_surface_init(int K, int Kpad, const double* __restrict__ P, int T,
                                                          const double* __restrict__ verts, const int* __restrict__ tris,
                                                          const int* __restrict__ hint, float4* __restrict__ qrec,
                                                          float* __restrict__ thrA, int* __restrict__ cnt) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= Kpad) return;
  cnt[k] = 0;
  if (k >= K) {
    qrec[k] = make_float4(1e30f, 1e30f, 1e30f, 0.f);
    thrA[k] = 0.f;
    return;
  }
  d3 p = ld3(P + 3 * k);
  int h = hint ? hint[k] : -1;
  double d2 = __builtin_inf();
  if (h >= 0 && h < T) d2 = tri_dist2(p, verts, tris, h, nullptr);
  if (!(d2 == d2)) d2 = __builtin_inf();  // degenerate hint triangle
  const double slack = kAbsSlack * (fabs(p.x) + fabs(p.y) + fabs(p.z));
  qrec[k] = make_float4((float)p.x, (float)p.y, (float)p.z, 0.f);
  thrA[k] = round_up_f32(sqrt(d2) * (1.0 + 2e-6) + slack);
}

__global__ void __launch_bounds__(kBlock) k_surface_filter(int T, const float4* __restrict__ spheres, int Kpad, int kchunk,
                                                            const float4* __restrict__ qrec, const float* __restrict__ thrA,
                                                            int* __restrict__ cnt, int* __restrict__ cand, int stride) {
  const int t = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = t < T;
  float cx = 3e38f, cy = 3e38f, cz = 3e38f, R = 0.f;  // out-of-range lanes: infinitely far away
  if (valid) {
    float4 s = spheres[t];
    cx = s.x; cy = s.y; cz = s.z; R = s.w;
  }
  const int k0 = blockIdx.y * kchunk;
  const int k1 = min(Kpad, k0 + kchunk);
  for (int k = k0; k < k1; k += kQU) {
    bool hit[kQU];
    unsigned long long m[kQU];
#pragma unroll
    for (int u = 0; u < kQU; ++u) {  // wave-uniform query records: scalar loads, kQU queries in flight
      const float4 q = qrec[k + u];
      const float tt = thrA[k + u] + R;
      const float dx = q.x - cx, dy = q.y - cy, dz = q.z - cz;
      const float dc2 = fmaf(dz, dz, fmaf(dy, dy, dx * dx));
      hit[u] = valid && dc2 <= tt * tt;
      m[u] = __ballot(hit[u]);
    }
#pragma unroll
    for (int u = 0; u < kQU; ++u)
      if (m[u] != 0ull) wave_append(m[u], hit[u], cnt + (k + u), cand + (size_t)(k + u) * stride, t);  // uniform branch
  }
}

// one wave per query
__global__ void __launch_bounds__(64) k_surface_resolve(int K, const double* __restrict__ P, const double* __restrict__ verts,
                                                         const int* __restrict__ tris, const int* __restrict__ cnt,
                                                         const int* __restrict__ cand, int stride, int* __restrict__ hint,
                                                         double* __restrict__ cp, double* __restrict__ d2out, int* __restrict__ triout) {
  const int k = blockIdx.x;
  const int n = cnt[k];
  const d3 p = ld3(P + 3 * k);
  const int* list = cand + (size_t)k * stride;
  double best = __builtin_inf();
  int bi = kNoIndex;
  for (int i = lane_id(); i < n; i += 64) {
    const int t = list[i];
    const double d2 = tri_dist2(p, verts, tris, t, nullptr);
    if (d2 < best || (d2 == best && t < bi)) { best = d2; bi = t; }  // NaN (degenerate triangle) never wins
  }
  wave_lexmin(best, bi);
  if (lane_id() == 0) {
    d3 c = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
    if (bi != kNoIndex) tri_dist2(p, verts, tris, bi, &c);
    if (cp) { cp[3 * k] = c.x; cp[3 * k + 1] = c.y; cp[3 * k + 2] = c.z; }
    if (d2out) d2out[k] = best;
    if (triout) triout[k] = bi == kNoIndex ? -1 : bi;
    if (hint) hint[k] = bi == kNoIndex ? -1 : bi;
  }
}

// ---------------------------------------------------------------- K3 nearest vertex

__global__ void __launch_bounds__(kBlock) k_vertex_init(int K, int Kpad, const double* __restrict__ P, int V,
                                                         const double* __restrict__ verts, const int* __restrict__ hint,
                                                         double* __restrict__ thr2, int* __restrict__ cnt) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= Kpad) return;
  cnt[k] = 0;
  if (k >= K) { thr2[k] = -1.0; return; }  // sentinel: nothing passes
  int h = hint ? hint[k] : -1;
  double d2 = __builtin_inf();
  if (h >= 0 && h < V) {
    d3 d = sub(ld3(P + 3 * k), ld3(verts + 3 * h));
    d2 = dot(d, d);
  }
  if (!(d2 == d2)) d2 = __builtin_inf();
  thr2[k] = d2;  // squared bound, same expression as the filter -> the hint vertex itself always passes
}

__global__ void __launch_bounds__(kBlock) k_vertex_filter(int V, const double* __restrict__ verts, int K, int Kpad, int kchunk,
                                                           const double* __restrict__ P, const double* __restrict__ thr2,
                                                           int* __restrict__ cnt, int* __restrict__ cand, int stride) {
  const int v = blockIdx.x * kBlock + threadIdx.x;
  const bool valid = v < V;
  d3 q = {0.0, 0.0, 0.0};
  if (valid) q = ld3(verts + 3 * v);
  const int k0 = blockIdx.y * kchunk;
  const int k1 = min(Kpad, k0 + kchunk);
  for (int k = k0; k < k1; k += kQU) {
    bool hit[kQU];
    unsigned long long m[kQU];
#pragma unroll
    for (int u = 0; u < kQU; ++u) {
      const int kk = min(k + u, K - 1);  // sentinel slots re-read the last real query; their bound is -1
      d3 p = {P[3 * kk], P[3 * kk + 1], P[3 * kk + 2]};  // wave-uniform
      d3 d = sub(p, q);
      const double d2 = dot(d, d);  // (dx·dx + dy·dy) + dz·dz, unfused — the value the argmin is defined on
      hit[u] = valid && d2 <= thr2[k + u];
      m[u] = __ballot(hit[u]);
    }
#pragma unroll
    for (int u = 0; u < kQU; ++u)
      if (m[u] != 0ull) wave_append(m[u], hit[u], cnt + (k + u), cand + (size_t)(k + u) * stride, v);
  }
}

__global__ void __launch_bounds__(64) k_vertex_resolve(int K, const double* __restrict__ P, const double* __restrict__ verts,
                                                        const int* __restrict__ cnt, const int* __restrict__ cand, int stride,
                                                        int* __restrict__ hint, double* __restrict__ d2out, int* __restrict__ idxout) {
  const int k = blockIdx.x;
  const int n = cnt[k];
  const d3 p = ld3(P + 3 * k);
  const int* list = cand + (size_t)k * stride;
  double best = __builtin_inf();
  int bi = kNoIndex;
  for (int i = lane_id(); i < n; i += 64) {
    const int v = list[i];
    d3 d = sub(p, ld3(verts + 3 * v));
    const double d2 = dot(d, d);
    if (d2 < best || (d2 == best && v < bi)) { best = d2; bi = v; }
  }
  wave_lexmin(best, bi);
  if (lane_id() == 0) {
    if (d2out) d2out[k] = best;
    if (idxout) idxout[k] = bi == kNoIndex ? -1 : bi;
    if (hint) hint[k] = bi == kNoIndex ? -1 : bi;
  }
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

void launch_tri_spheres(hipStream_t st, int T, const double* verts, const int* tris, float4* spheres) {
  if (T <= 0) return;
  { ProfScope _ps(st, KID_TRI_SPHERES);
    hipLaunchKernelGGL(k_tri_spheres, dim3(cdiv(T, kBlock)), dim3(kBlock), 0, st, T, verts, tris, spheres); }
}

// queries are processed in batches small enough that every query can list ALL elements as candidates
static int query_batch(int K, int n_elems, size_t cand_capacity) {
  size_t kb = cand_capacity / (size_t)(n_elems > 0 ? n_elems : 1);
  if (kb < 1) kb = 1;
  if (kb > (size_t)K) kb = K;
  return (int)kb;
}

void launch_surface_query(hipStream_t st, int T, const double* verts, const int* tris, const float4* spheres,
                          int K, const double* P, int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri) {
  if (K <= 0) return;
  const int Kb = query_batch(K, T, qb.cand_capacity);
  for (int b0 = 0; b0 < K; b0 += Kb) {
    const int kb = K - b0 < Kb ? K - b0 : Kb;
    const int Kpad = (kb + kQU - 1) / kQU * kQU;
    const double* Pb = P + 3 * (size_t)b0;
    int* hb = hint ? hint + b0 : nullptr;
    { ProfScope _ps(st, KID_SURFACE_INIT);
      hipLaunchKernelGGL(k_surface_init, dim3(cdiv(Kpad, kBlock)), dim3(kBlock), 0, st, kb, Kpad, Pb, T, verts, tris, hb, qb.qrec,
                         qb.thrA, qb.cnt); }
    if (T > 0) {
      const int tblocks = cdiv(T, kBlock);
      int ksplit, kchunk;
      split_queries(tblocks, Kpad, &ksplit, &kchunk);
      kchunk = (kchunk + kQU - 1) / kQU * kQU;
      ksplit = cdiv(Kpad, kchunk);
      { ProfScope _ps(st, KID_SURFACE_FILTER);
        hipLaunchKernelGGL(k_surface_filter, dim3(tblocks, ksplit), dim3(kBlock), 0, st, T, spheres, Kpad, kchunk, qb.qrec, qb.thrA,
                           qb.cnt, qb.cand, T); }
    }
    { ProfScope _ps(st, KID_SURFACE_RESOLVE);
      hipLaunchKernelGGL(k_surface_resolve, dim3(kb), dim3(64), 0, st, kb, Pb, verts, tris, qb.cnt, qb.cand, T, hb,
                         cp ? cp + 3 * (size_t)b0 : nullptr, d2 ? d2 + b0 : nullptr, tri ? tri + b0 : nullptr); }
  }
}

void launch_vertex_query(hipStream_t st, int V, const double* verts, int K, const double* P, int* hint,
                         const QueryBuffers& qb, double* d2, int* idx) {
  if (K <= 0) return;
  const int Kb = query_batch(K, V, qb.cand_capacity);
  for (int b0 = 0; b0 < K; b0 += Kb) {
    const int kb = K - b0 < Kb ? K - b0 : Kb;
    const int Kpad = (kb + kQU - 1) / kQU * kQU;
    const double* Pb = P + 3 * (size_t)b0;
    int* hb = hint ? hint + b0 : nullptr;
    { ProfScope _ps(st, KID_VERTEX_INIT);
      hipLaunchKernelGGL(k_vertex_init, dim3(cdiv(Kpad, kBlock)), dim3(kBlock), 0, st, kb, Kpad, Pb, V, verts, hb, qb.thr2, qb.cnt); }
    if (V > 0) {
      const int vblocks = cdiv(V, kBlock);
      int ksplit, kchunk;
      split_queries(vblocks, Kpad, &ksplit, &kchunk);
      kchunk = (kchunk + kQU - 1) / kQU * kQU;
      ksplit = cdiv(Kpad, kchunk);
      { ProfScope _ps(st, KID_VERTEX_FILTER);
        hipLaunchKernelGGL(k_vertex_filter, dim3(vblocks, ksplit), dim3(kBlock), 0, st, V, verts, kb, Kpad, kchunk, Pb, qb.thr2,
                           qb.cnt, qb.cand, V); }
    }
    { ProfScope _ps(st, KID_VERTEX_RESOLVE);
      hipLaunchKernelGGL(k_vertex_resolve, dim3(kb), dim3(64), 0, st, kb, Pb, verts, qb.cnt, qb.cand, V, hb,
                         d2 ? d2 + b0 : nullptr, idx ? idx + b0 : nullptr); }
  }
}

}  // namespace icp
