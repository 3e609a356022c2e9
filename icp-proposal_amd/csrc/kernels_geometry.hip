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
#include <cstdlib>

#include <algorithm>
#include "icp_kernels.hpp"
#include "icp_search.hpp"

namespace icp {

thread_local Profiler* g_prof = nullptr;
const char* const kKernelNames[KID_COUNT] = {
    "k_instance", "k_surface_init", "k_surface_filter", "k_surface_resolve",
    "k_vertex_init", "k_vertex_filter", "k_vertex_resolve", "k_tri_spheres", "k_correspond",
    "k_regression", "k_posterior_factor", "k_transition_tail", "k_posterior_eigen", "k_propose", "k_reduce",
    "k_step_begin", "k_step_filter", "k_step_resolve", "k_step_regression", "k_step_finish"};

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

constexpr int kBlock = kSearchBlock;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

__global__ void __launch_bounds__(kBlock) k_instance(int N, int r, const double* __restrict__ Qp,
                                                      const double* __restrict__ ref, const double* __restrict__ mean,
                                                      Pose pose, const double* __restrict__ coeffs, double* __restrict__ x) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) instance_vertex(i, N, r, Qp, ref, mean, pose, coeffs, x);
}

__global__ void __launch_bounds__(kBlock) k_instance_keep(int N, int r, const double* __restrict__ Qp, const double* __restrict__ ref,
                                                           const double* __restrict__ mean, Pose pose, const double* __restrict__ coeffs,
                                                           double* __restrict__ x, double* __restrict__ defo) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i < N) instance_vertex_keep(i, N, r, Qp, ref, mean, pose, coeffs, x, defo);
}
// a state with the coefficients of another one: its points from that state's kept deformations (defo_in may equal defo_out)
__global__ void __launch_bounds__(kBlock) k_instance_pose(int N, const double* __restrict__ ref, Pose pose, const double* defo_in,
                                                           double* __restrict__ x, double* defo_out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const double a0 = defo_in[3 * i], a1 = defo_in[3 * i + 1], a2 = defo_in[3 * i + 2];
  defo_out[3 * i] = a0; defo_out[3 * i + 1] = a1; defo_out[3 * i + 2] = a2;
  const d3 p = instance_pose(i, ref, pose, a0, a1, a2);
  x[3 * i] = p.x; x[3 * i + 1] = p.y; x[3 * i + 2] = p.z;
}

__global__ void __launch_bounds__(kBlock) k_vertex_normals(int N, const double* __restrict__ x, const int* __restrict__ tris,
                                                            const int* __restrict__ adj_off, const int* __restrict__ adj,
                                                            double* __restrict__ normals) {
  int v = blockIdx.x * kBlock + threadIdx.x;
  if (v >= N) return;
  d3 n = vertex_normal(x, tris, adj_off, adj, v);
  normals[3 * v] = n.x; normals[3 * v + 1] = n.y; normals[3 * v + 2] = n.z;
}

// position pos of the list holds triangle order[pos] (nullptr: the identity); the triangle ids follow the T spheres in the
// same buffer (sphere_triangles), for the filter to name its candidates
__global__ void __launch_bounds__(kBlock) k_tri_spheres(int T, const double* __restrict__ verts, const int* __restrict__ tris,
                                                         const int* __restrict__ order, float4* __restrict__ spheres) {
  const int pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= T) return;
  const int t = order ? order[pos] : pos;
  spheres[pos] = tri_sphere(verts, tris, t);
  sphere_triangles(spheres, T)[pos] = t;
}

__global__ void __launch_bounds__(kBlock) k_surface_init(SurfaceTask q) { surface_init(q, blockIdx.x * kBlock + threadIdx.x); }
#ifdef ICP_FILTER_STAMPS  // (developer aid, see icp_search.hpp: this translation unit's sums — the per-stage surface queries)
extern "C" __attribute__((visibility("default"))) void icp_debug_filter_stamps_geometry(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  static unsigned long long h[kFltSlots][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_filter_stamps), sizeof(h));
  for (int i = 0; i < 8; ++i) { out[i] = 0; for (int s = 0; s < kFltSlots; ++s) out[i] += h[s][i]; }
  if (reset) { for (auto& row : h) for (auto& v : row) v = 0; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_filter_stamps), h, sizeof(h)); }
}
#endif
// kPrepared: resident spheres, bounds taken by k_surface_init (every query of this file's launcher that hands spheres in): the form of
// the filter without the f64 sphere / bound paths — 38 registers instead of 96, eight workgroups per CU (see surface_filter)
template <bool kPrepared>
__global__ void __launch_bounds__(kBlock, kPrepared ? 8 : 1) k_surface_filter(SurfaceTask q) { surface_filter<kPrepared>(q, blockIdx.x, blockIdx.y); }
__global__ void __launch_bounds__(64) k_surface_resolve(SurfaceTask q) {
  double best; int tri; d3 cp;
  surface_resolve(q, blockIdx.x, &best, &tri, &cp);
}
__global__ void __launch_bounds__(kBlock) k_vertex_init(VertexTask q) { vertex_init(q, blockIdx.x * kBlock + threadIdx.x); }
__global__ void __launch_bounds__(kBlock) k_vertex_filter(VertexTask q) { vertex_filter(q, blockIdx.x, blockIdx.y); }
__global__ void __launch_bounds__(64) k_vertex_resolve(VertexTask q) {
  double best; int idx;
  vertex_resolve(q, blockIdx.x, &best, &idx);
}

}  // namespace

// How many chains' searches the launch being put together carries (search_chains_hint: set by the batched / wide step's host side).  A
// lone chain's filter launch wants many workgroups — it is a chain of latencies — and splits its queries over them; B chains side by
// side launch B times as many, and then the per-wave set-up (the elements' box or ball, the staging of a query tile) is what the launch
// consists of: 25 chains of the face configuration, vertex searches split ten ways — 293 instructions per wave, most of them set-up.
static thread_local int tl_search_chains = 1;
void search_chains_hint(int n_chains) { tl_search_chains = n_chains > 1 ? n_chains : 1; }

// enough waves to fill 256 CUs, but at least ~8 queries per wave so the element load is amortised; kchunk multiple of kQU
void split_queries(int n_elem_blocks, int Kpad, int* ksplit, int* kchunk) {
  static const int target_waves_one = dev_env("ICP_FILTER_WAVES") ? std::atoi(dev_env("ICP_FILTER_WAVES")) : 4096;
  const int target_waves = std::max(target_waves_one / tl_search_chains, 1);
  int want = cdiv(target_waves, n_elem_blocks * (kBlock / 64));
  int s = want < 1 ? 1 : want;
  int maxs = cdiv(Kpad, 8);
  if (s > maxs) s = maxs;
  if (s < 1) s = 1;
  if (s > 65535) s = 65535;
  int kc = cdiv(Kpad, s);
  kc = (kc + kQU - 1) / kQU * kQU;
  *kchunk = kc;
  *ksplit = cdiv(Kpad, kc);
}

// Surface filter: its two-level test makes the arithmetic per query negligible, what is left is a chain of latencies per
// workgroup (spheres + queries in, ball, tests, hits out) whatever the number of queries it takes (tools/ab_chunk.sh: 64
// to 512 queries per workgroup measure the same for one chain) — so few workgroups: a batch of chains launches B times as
// many.  256 queries per workgroup: two workgroups per block of spheres for the 308 queries of the femur step.
void split_surface_queries(int Kpad, int* ksplit, int* kchunk) {
  static const int tile_one = dev_env("ICP_SURFACE_CHUNK") ? std::atoi(dev_env("ICP_SURFACE_CHUNK")) : 256;  // (A/B: <= 512)
  const int tile = tl_search_chains >= 16 ? kSurfaceTile : tile_one;  // (many chains a launch: a whole LDS tile per workgroup — half the waves)
  int kc = Kpad < tile ? Kpad : tile;
  kc = (kc + kQU - 1) / kQU * kQU;
  if (kc < kQU) kc = kQU;
  *kchunk = kc;
  *ksplit = cdiv(Kpad, kc);
}

// queries are processed in batches whose candidate lists (cand_stride entries each) fit the scratch
int query_batch(int K, int n_elems, size_t cand_capacity) {
  size_t kb = cand_capacity / (size_t)cand_stride(n_elems);
  if (kb > 4) kb -= 4;  // room for the sentinel slots
  if (kb < 1) kb = 1;
  if (kb > (size_t)K) kb = K;
  return (int)kb;
}

SurfaceTask make_surface_task(int T, const double* verts, const int* tris, const float4* spheres, int K, const double* P,
                              int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri) {
  SurfaceTask q{};
  q.K = K; q.Kpad = (K + kQU - 1) / kQU * kQU; q.T = T; q.stride = cand_stride_for(T, K, qb.cand_capacity);
  q.P = P; q.verts = verts; q.tris = tris; q.spheres = spheres; q.order = nullptr; q.hint = hint;
  q.qrec = qb.qrec; q.thrA = qb.thrA; q.cnt = qb.cnt; q.cand = qb.cand;
  q.cp = cp; q.d2 = d2; q.tri = tri;
  q.stats = g_prof ? g_prof->counters : nullptr;
  q.tblocks = cdiv(T > 0 ? T : 1, kBlock * kSpheresPerLane);
  split_surface_queries(q.Kpad, &q.ksplit, &q.kchunk);
  return q;
}

VertexTask make_vertex_task(int V, const double* verts, int K, const double* P, int* hint, const QueryBuffers& qb, double* d2, int* idx) {
  VertexTask q{};
  q.K = K; q.Kpad = (K + kQU - 1) / kQU * kQU; q.V = V; q.stride = cand_stride_for(V, K, qb.cand_capacity);
  q.P = P; q.verts = verts; q.hint = hint; q.thr2 = qb.thr2; q.cnt = qb.cnt; q.cand = qb.cand; q.d2 = d2; q.idx = idx;
  q.stats = g_prof ? g_prof->counters : nullptr;
  q.vblocks = cdiv(V > 0 ? V : 1, kBlock);
  split_queries(q.vblocks, q.Kpad, &q.ksplit, &q.kchunk);
  return q;
}

void launch_instance(hipStream_t st, int N, int r, const double* Qp, const double* ref, const double* mean,
                     const Pose& pose, const double* coeffs, double* x) {
  ProfScope _ps(st, KID_INSTANCE);
  hipLaunchKernelGGL(k_instance, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, r, Qp, ref, mean, pose, coeffs, x);
}

void launch_instance_keep(hipStream_t st, int N, int r, const double* Qp, const double* ref, const double* mean,
                          const Pose& pose, const double* coeffs, double* x, double* defo) {
  ProfScope _ps(st, KID_INSTANCE);
  hipLaunchKernelGGL(k_instance_keep, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, r, Qp, ref, mean, pose, coeffs, x, defo);
}
void launch_instance_pose(hipStream_t st, int N, const double* ref, const Pose& pose, const double* defo_in, double* x, double* defo_out) {
  ProfScope _ps(st, KID_INSTANCE);
  hipLaunchKernelGGL(k_instance_pose, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, ref, pose, defo_in, x, defo_out);
}

void launch_vertex_normals(hipStream_t st, int N, const double* x, const int* tris, const int* adj_off,
                           const int* adj, double* normals) {
  hipLaunchKernelGGL(k_vertex_normals, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, x, tris, adj_off, adj, normals);
}

void launch_tri_spheres(hipStream_t st, int T, const double* verts, const int* tris, const int* order, float4* spheres) {
  if (T <= 0) return;
  ProfScope _ps(st, KID_TRI_SPHERES);
  hipLaunchKernelGGL(k_tri_spheres, dim3(cdiv(T, kBlock)), dim3(kBlock), 0, st, T, verts, tris, order, spheres);
}

// Order of the triangles in the sphere list: a k-d split of the centroids (longest box axis, cut at a multiple of 128
// positions) down to runs of 128 — the 128 spheres a wave of the surface filter holds then always form ONE compact patch.
// In file order most runs are patches, too, but the few that are not (a run that ends one strip of the surface and begins
// another) keep every query in the filter's second level and set the duration of the whole launch.
namespace {
void kd_runs(int* idx, int lo, int hi, const float* c) {
  const int n = hi - lo;
  if (n <= 128) return;
  float mn[3] = {c[3 * idx[lo]], c[3 * idx[lo] + 1], c[3 * idx[lo] + 2]}, mx[3] = {mn[0], mn[1], mn[2]};
  for (int i = lo + 1; i < hi; ++i)
    for (int d = 0; d < 3; ++d) {
      const float v = c[3 * idx[i] + d];
      mn[d] = v < mn[d] ? v : mn[d];
      mx[d] = v > mx[d] ? v : mx[d];
    }
  int ax = 0;
  for (int d = 1; d < 3; ++d)
    if (mx[d] - mn[d] > mx[ax] - mn[ax]) ax = d;
  int half = (n / 2 + 64) / 128 * 128;
  if (half < 128) half = 128;
  if (half >= n) half = (n - 1) / 128 * 128;
  std::nth_element(idx + lo, idx + lo + half, idx + hi, [&](int a, int b) {
    const float va = c[3 * a + ax], vb = c[3 * b + ax];
    return va < vb || (va == vb && a < b);
  });
  kd_runs(idx, lo, lo + half, c);
  kd_runs(idx, lo + half, hi, c);
}
}  // namespace
std::vector<int> coherent_triangle_order(int V, int T, const double* verts, const int* tris) {
  std::vector<int> idx(T > 0 ? T : 0);
  std::vector<float> c((size_t)3 * (T > 0 ? T : 0));
  for (int t = 0; t < T; ++t) {
    idx[t] = t;
    for (int d = 0; d < 3; ++d) {
      double s = 0.0;
      for (int k = 0; k < 3; ++k) {
        const int v = tris[3 * t + k];
        s += (v >= 0 && v < V) ? verts[3 * (size_t)v + d] : 0.0;
      }
      const float m = (float)(s / 3.0);
      c[3 * (size_t)t + d] = m == m && m - m == 0.f ? m : 0.f;  // (a non-finite vertex must not break the ordering's comparisons)
    }
  }
  if (T > 128) kd_runs(idx.data(), 0, T, c.data());
  return idx;
}

void launch_surface_query(hipStream_t st, int T, const double* verts, const int* tris, const float4* spheres,
                          int K, const double* P, int* hint, const QueryBuffers& qb, double* cp, double* d2, int* tri) {
  if (K <= 0) return;
  const int Kb = query_batch(K, T, qb.cand_capacity);
  for (int b0 = 0; b0 < K; b0 += Kb) {
    const int kb = K - b0 < Kb ? K - b0 : Kb;
    SurfaceTask q = make_surface_task(T, verts, tris, spheres, kb, P + 3 * (size_t)b0, hint ? hint + b0 : nullptr, qb,
                                      cp ? cp + 3 * (size_t)b0 : nullptr, d2 ? d2 + b0 : nullptr, tri ? tri + b0 : nullptr);
    { ProfScope _ps(st, KID_SURFACE_INIT);
      hipLaunchKernelGGL(k_surface_init, dim3(cdiv(q.Kpad, kBlock)), dim3(kBlock), 0, st, q); }
    if (T > 0) {
      ProfScope _ps(st, KID_SURFACE_FILTER);
      if (q.spheres != nullptr && q.thrA != nullptr) hipLaunchKernelGGL(k_surface_filter<true>, dim3(q.tblocks, q.ksplit), dim3(kBlock), 0, st, q);
      else hipLaunchKernelGGL(k_surface_filter<false>, dim3(q.tblocks, q.ksplit), dim3(kBlock), 0, st, q);
    }
    { ProfScope _ps(st, KID_SURFACE_RESOLVE);
      hipLaunchKernelGGL(k_surface_resolve, dim3(kb), dim3(64), 0, st, q); }
  }
}

void launch_vertex_query(hipStream_t st, int V, const double* verts, int K, const double* P, int* hint,
                         const QueryBuffers& qb, double* d2, int* idx) {
  if (K <= 0) return;
  const int Kb = query_batch(K, V, qb.cand_capacity);
  for (int b0 = 0; b0 < K; b0 += Kb) {
    const int kb = K - b0 < Kb ? K - b0 : Kb;
    VertexTask q = make_vertex_task(V, verts, kb, P + 3 * (size_t)b0, hint ? hint + b0 : nullptr, qb, d2 ? d2 + b0 : nullptr,
                                    idx ? idx + b0 : nullptr);
    { ProfScope _ps(st, KID_VERTEX_INIT);
      hipLaunchKernelGGL(k_vertex_init, dim3(cdiv(q.Kpad, kBlock)), dim3(kBlock), 0, st, q); }
    if (V > 0) {
      ProfScope _ps(st, KID_VERTEX_FILTER);
      hipLaunchKernelGGL(k_vertex_filter, dim3(q.vblocks, q.ksplit), dim3(kBlock), 0, st, q);
    }
    { ProfScope _ps(st, KID_VERTEX_RESOLVE);
      hipLaunchKernelGGL(k_vertex_resolve, dim3(kb), dim3(64), 0, st, q); }
  }
}

}  // namespace icp
