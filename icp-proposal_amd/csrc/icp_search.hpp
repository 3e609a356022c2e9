// icp_search.hpp — device-side bodies of the two brute-force searches (K3 point×vertex, K4 point×triangle).
// Included by kernels_geometry.hip (one kernel per stage) and kernels_step.hip (stages of several query batches merged
// into one launch per Metropolis–Hastings step).  Design notes: kernels_geometry.hip.
#pragma once
#include "icp_device.hpp"

namespace icp {

constexpr int kSearchBlock = 256;
#ifdef ICP_FILTER_STAMPS  // developer aid: time (100 MHz ticks) per part of a surface-filter workgroup, summed over the workgroups
constexpr int kFltSlots = 2048;  // (sums spread over many lines: thousands of workgroups adding to ONE line would time their own atomics)
static __device__ unsigned long long g_filter_stamps[kFltSlots][8];  // (one per translation unit)
#define FLT_T() ((long long)__builtin_amdgcn_s_memrealtime())
#define FLT_DRAIN() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")  // (a part ends when its loads / stores have come back)
#define FLT_ADD(i, v) do { if (threadIdx.x == 0) atomicAdd(&g_filter_stamps[(blockIdx.x + 977u * blockIdx.y) % kFltSlots][i], (unsigned long long)(v)); } while (0)
#else
#define FLT_T() 0ll
#define FLT_DRAIN()
#define FLT_ADD(i, v) (void)(v)
#endif
constexpr double kAbsSlack = 3.0 / 8388608.0;  // 3·2^-23 per unit of |coordinate|: covers rounding a point to f32
constexpr int kQU = 4;                         // queries per unrolled filter iteration

__device__ __forceinline__ float round_up_f32(double v) { return nextafterf((float)v, __builtin_inff()); }
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// f32 bounding sphere of triangle t: centre = centroid rounded to f32, radius = max corner distance (f64) inflated by
// the rounding of centre and arithmetic, rounded up
__device__ __forceinline__ float4 tri_sphere(const double* __restrict__ verts, const int* __restrict__ tris, int t) {
  d3 a = ld3(verts + 3 * tris[3 * t]), b = ld3(verts + 3 * tris[3 * t + 1]), c = ld3(verts + 3 * tris[3 * t + 2]);
  d3 m = {(a.x + b.x + c.x) / 3.0, (a.y + b.y + c.y) / 3.0, (a.z + b.z + c.z) / 3.0};
  d3 da = sub(a, m), db = sub(b, m), dc = sub(c, m);
  double r2 = fmax(dot(da, da), fmax(dot(db, db), dot(dc, dc)));
  double R = sqrt(r2) * (1.0 + 2e-6) + kAbsSlack * (fabs(m.x) + fabs(m.y) + fabs(m.z));
  return make_float4((float)m.x, (float)m.y, (float)m.z, round_up_f32(R));
}

// Candidate lists of the filter loops.  A hit costs an atomic round trip to L2 before its list entries can
// be stored; taken one query after the other, the few waves whose elements lie near many queries spend most of a filter
// launch waiting for those round trips.  Instead the wave parks each (query, hit mask) event in the registers of ONE
// lane — lane e keeps event e — and carries on; at the end (or when 64 events are parked) every lane settles its own
// event: one atomic per event, all of them in flight together, then the lane writes the entries of its mask itself (the
// element behind bit b of a mask is known from the wave's position).  The order of a list's entries is irrelevant — the
// resolve step takes the lexicographic (distance, index) minimum — and its counter ends at the number of entries.
struct ParkedHits {
  int n = 0;                       // wave-uniform
  int k = 0;                       // per lane: the event this lane holds
  unsigned long long m0 = 0, m1 = 0;
};
__device__ __forceinline__ void park_hits(ParkedHits& ph, int k, unsigned long long m0, unsigned long long m1) {
  if (lane_id() == ph.n) { ph.k = k; ph.m0 = m0; ph.m1 = m1; }
  ++ph.n;
}
// element behind bit b of mask w (w = 0, 1): position first + b·step + w of the streamed list
__device__ __forceinline__ void settle_hits(ParkedHits& ph, int* __restrict__ cnt, int* __restrict__ cand, size_t stride, int first, int step) {
  if (lane_id() < ph.n) {
    const int n0 = __popcll(ph.m0), n1 = __popcll(ph.m1);
    int pos = atomicAdd(cnt + ph.k, n0 + n1);  // (the counter keeps counting past the capacity: that is how the resolve stage knows)
    int* list = cand + (size_t)ph.k * stride;
    const int cap = (int)stride;
    unsigned long long m = ph.m0;
    while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; if (pos < cap) list[pos] = first + b * step; ++pos; }
    m = ph.m1;
    while (m) { const int b = __ffsll((long long)m) - 1; m &= m - 1; if (pos < cap) list[pos] = first + b * step + 1; ++pos; }
  }
  ph.n = 0;
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
// `p` = the query point (q.P[k] for k < K; ignored for sentinel slots), passed in so that a producer kernel can
// initialise a query from a point it has just computed
// (the hinted triangle's corners are fetched by the caller — a producer kernel does that ahead of the point itself)
struct HintTriangle { bool have; d3 a, b, c; };
__device__ __forceinline__ HintTriangle load_hint_triangle(const SurfaceTask& q, int k) {
  HintTriangle ht{false, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  if (k >= q.K) return ht;
  const int h = q.hint ? q.hint[k] : -1;
  if (h >= 0 && h < q.T) {
    const int ia = q.tris[3 * h], ib = q.tris[3 * h + 1], ic = q.tris[3 * h + 2];
    ht.have = true;
    ht.a = ld3(q.verts + 3 * ia); ht.b = ld3(q.verts + 3 * ib); ht.c = ld3(q.verts + 3 * ic);
  }
  return ht;
}
// f32 record and inflated bound of one real query (k < K)
__device__ __forceinline__ void surface_bound(d3 p, const HintTriangle& ht, float4* rec, float* thr) {
  double d2 = __builtin_inf();
  if (ht.have) {  // = tri_dist2 of the hinted triangle
    const d3 d = sub(p, closest_point_triangle(p, ht.a, ht.b, ht.c));
    d2 = dot(d, d);
  }
  if (!(d2 == d2)) d2 = __builtin_inf();  // degenerate hint triangle
  const double slack = kAbsSlack * (fabs(p.x) + fabs(p.y) + fabs(p.z));
  *rec = make_float4((float)p.x, (float)p.y, (float)p.z, 0.f);
  *thr = round_up_f32(sqrt(d2) * (1.0 + 2e-6) + slack);
}
__device__ __forceinline__ void surface_init_with(const SurfaceTask& q, int k, d3 p, const HintTriangle& ht) {
  if (k >= q.Kpad) return;
  q.cnt[k] = 0;
  if (k >= q.K) {
    q.qrec[k] = make_float4(1e30f, 1e30f, 1e30f, 0.f);
    q.thrA[k] = 0.f;
    return;
  }
  float4 rec; float thr;
  surface_bound(p, ht, &rec, &thr);
  q.qrec[k] = rec;
  q.thrA[k] = thr;
}
__device__ __forceinline__ void surface_init_at(const SurfaceTask& q, int k, d3 p) {
  surface_init_with(q, k, p, load_hint_triangle(q, k));
}
__device__ __forceinline__ void surface_init(const SurfaceTask& q, int k) {
  d3 p = {0.0, 0.0, 0.0};
  if (k < q.K) p = ld3(q.P + 3 * k);
  surface_init_at(q, k, p);
}

// workgroup (bx, by) of the tblocks × ksplit filter grid; kSearchBlock threads, TWO spheres per lane so that the whole
// test runs on packed-f32 instructions (v_pk_add/mul/fma_f32: two sphere tests per issue slot; the wave-uniform query
// operands come straight from SGPRs with op_sel broadcasts)
constexpr int kSpheresPerLane = 2;
typedef float f2_t __attribute__((ext_vector_type(2)));

constexpr int kFilterTile = 128;   // queries staged in LDS at a time (vertex filter)
constexpr int kSurfaceTile = 512;  // … surface filter: a whole chunk (split_surface_queries), so that a workgroup's query
                                   // records arrive in one go beside its spheres — the kernel is a chain of latencies, not of arithmetic

// Two levels.  A wave holds 128 CONSECUTIVE triangles, i.e. a patch of the surface (mesh files list neighbouring triangles
// together; a synthetic subdivision lists the children of a triangle together): their spheres fit a ball of a few mm
// — centre C = mean of the centres, radius Rw = max(|c_i − C| + R_i), taken once per wave by shuffles — and a query can
// only have a candidate in the wave if |q − C| <= bound + Rw (triangle inequality; Rw carries slack for the f32
// rounding of both tests, so whatever passes the sphere test below passes this one).  That test is made for 64 queries
// at a time, one per lane; the sphere test — every lane its two spheres against ONE query — runs only for the queries
// that survive: 2 % of the (wave, query) pairs on the 58k-vertex target.  Every pair is still decided, the set of
// candidates is the one the plain double loop gives; a wave whose triangles are scattered over the surface just keeps
// (almost) all queries.
// kPrepared: the caller vouches for q.spheres and q.thrA (a search of a mesh whose spheres are resident, its bounds taken by an
// earlier launch) — the f64 paths that make spheres and bounds here are compiled out, and with them two thirds of the kernel's
// registers (102 -> under 64: eight workgroups per CU instead of five; the batched filter launch lives on that)
template <bool kPrepared = false>
__device__ __forceinline__ void surface_filter(const SurfaceTask& q, int bx, int by) {
  __shared__ float4 s_q[kSurfaceTile];
  __shared__ float s_thr[kSurfaceTile];
  const long long ft0 = FLT_T();
  const int t0 = (bx * kSearchBlock + threadIdx.x) * kSpheresPerLane;
  const bool v0 = t0 < q.T, v1 = t0 + 1 < q.T;
  // out-of-range lanes: centre NaN — their squared distance is NaN and passes no threshold, not even an infinite one
  // (a query without a usable hint has bound +inf: everything in range is a candidate)
  f2_t cx = {__builtin_nanf(""), __builtin_nanf("")}, cy = cx, cz = cx, R = {0.f, 0.f};
  auto sphere_at = [&](int t) {
    if constexpr (kPrepared) return q.spheres[t];
    else return q.spheres ? q.spheres[t] : tri_sphere(q.verts, q.tris, q.order[t]);
  };
  if (v0) { const float4 s = sphere_at(t0); cx.x = s.x; cy.x = s.y; cz.x = s.z; R.x = s.w; }
  if (v1) { const float4 s = sphere_at(t0 + 1); cx.y = s.x; cy.y = s.y; cz.y = s.z; R.y = s.w; }
  // ---- the wave's ball (a sphere that is not finite — a triangle with a non-finite corner — passes no test by itself
  // and must not spoil the ball of its neighbours)
  auto finite4 = [](float a, float b, float c, float d) { return fabsf(a) + fabsf(b) + fabsf(c) + fabsf(d) <= 3.0e38f; };
  const bool b0 = v0 && finite4(cx.x, cy.x, cz.x, R.x), b1 = v1 && finite4(cx.y, cy.y, cz.y, R.y);
  float sx = (b0 ? cx.x : 0.f) + (b1 ? cx.y : 0.f), sy = (b0 ? cy.x : 0.f) + (b1 ? cy.y : 0.f), sz = (b0 ? cz.x : 0.f) + (b1 ? cz.y : 0.f);
  float sn = (b0 ? 1.f : 0.f) + (b1 ? 1.f : 0.f);
  for (int o = 32; o > 0; o >>= 1) {
    sx += __shfl_xor(sx, o, 64); sy += __shfl_xor(sy, o, 64); sz += __shfl_xor(sz, o, 64); sn += __shfl_xor(sn, o, 64);
  }
  const bool wave_live = sn > 0.f;  // (uniform; a wave past the end of the list has nothing to offer)
  const float inv_n = wave_live ? 1.f / sn : 0.f;
  const float Cx = sx * inv_n, Cy = sy * inv_n, Cz = sz * inv_n;
  float rad = 0.f;
  if (b0) { const float dx = cx.x - Cx, dy = cy.x - Cy, dz = cz.x - Cz; rad = fmaxf(rad, sqrtf(dx * dx + dy * dy + dz * dz) + R.x); }
  if (b1) { const float dx = cx.y - Cx, dy = cy.y - Cy, dz = cz.y - Cz; rad = fmaxf(rad, sqrtf(dx * dx + dy * dy + dz * dz) + R.y); }
  for (int o = 32; o > 0; o >>= 1) rad = fmaxf(rad, __shfl_xor(rad, o, 64));
  // slack: a few hundred ulps of the largest magnitudes involved (the tests' own rounding errors are a few ulps)
  const float Rw = rad * 1.00002f + 2e-5f * (fabsf(Cx) + fabsf(Cy) + fabsf(Cz) + rad);
  const int k0 = by * q.kchunk;
  const int k1 = min(q.Kpad, k0 + q.kchunk);
  ParkedHits ph;
  FLT_DRAIN();
  const long long ft1 = FLT_T();
  long long ft_stage = 0, ft_test = 0;
  const int first = (bx * kSearchBlock + (threadIdx.x & ~63)) * kSpheresPerLane;  // list position of lane 0's first sphere
                                                                                  // (candidates are named by position: see surface_resolve)
  const int lane = threadIdx.x & 63;
  unsigned n_ball = 0, n_sphere = 0;  // (wave-uniform; reported only when profiling)
  for (int kt = k0; kt < k1; kt += kSurfaceTile) {
    // this workgroup's queries go through LDS: one coalesced read per tile; the loops below never wait for global memory
    const int nq = min(kSurfaceTile, k1 - kt);
    const long long fa = FLT_T();
    if (kt != k0) __syncthreads();
    if (kPrepared || q.thrA) {
      for (int i = threadIdx.x; i < nq; i += kSearchBlock) { s_q[i] = q.qrec[kt + i]; s_thr[i] = q.thrA[kt + i]; }
    } else if constexpr (!kPrepared) {  // the bounds are taken here (the searched mesh was not complete when the step's first launch ran)
      for (int i = threadIdx.x; i < nq; i += kSearchBlock) {
        const int k = kt + i;
        float4 rec = make_float4(1e30f, 1e30f, 1e30f, 0.f);
        float thr = 0.f;
        if (k < q.K) surface_bound(ld3(q.P + 3 * k), load_hint_triangle(q, k), &rec, &thr);
        s_q[i] = rec; s_thr[i] = thr;
      }
    }
    __syncthreads();
    const long long fb = FLT_T();
    ft_stage += fb - fa;
    if (!wave_live) continue;
    for (int g = 0; g < nq; g += 64) {
      bool near = false;
      if (g + lane < nq) {  // lane l: query g + l against the wave's ball
        const float4 qq = s_q[g + lane];
        const float dx = qq.x - Cx, dy = qq.y - Cy, dz = qq.z - Cz;
        const float lim = s_thr[g + lane] + Rw + 2e-5f * (fabsf(qq.x) + fabsf(qq.y) + fabsf(qq.z));
        near = dx * dx + dy * dy + dz * dz <= lim * lim;  // (sentinel slots sit at 1e30 with bound 0: never; bound +inf: always)
      }
      unsigned long long todo = __ballot(near);
      n_ball += (unsigned)min(64, nq - g);
      n_sphere += (unsigned)__popcll(todo);
      while (todo) {  // wave-uniform: the surviving queries, each against the lanes' spheres
        const int j = g + __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const float4 qq = s_q[j];
        const float th = s_thr[j];
        const f2_t tt = f2_t{th, th} + R;
        const f2_t dx = f2_t{qq.x, qq.x} - cx, dy = f2_t{qq.y, qq.y} - cy, dz = f2_t{qq.z, qq.z} - cz;
        const f2_t dc2 = __builtin_elementwise_fma(dz, dz, __builtin_elementwise_fma(dy, dy, dx * dx));
        const f2_t t2 = tt * tt;
        const unsigned long long m0 = __ballot(dc2.x <= t2.x), m1 = __ballot(dc2.y <= t2.y);
        if ((m0 | m1) != 0ull) {
          park_hits(ph, kt + j, m0, m1);
          if (ph.n == 64) settle_hits(ph, q.cnt, q.cand, q.stride, first, kSpheresPerLane);
        }
      }
    }
    ft_test += FLT_T() - fb;
  }
  const long long ft2 = FLT_T();
  settle_hits(ph, q.cnt, q.cand, q.stride, first, kSpheresPerLane);
  FLT_DRAIN();
  const long long ft3 = FLT_T();
  FLT_ADD(0, 1); FLT_ADD(1, ft1 - ft0); FLT_ADD(2, ft_stage); FLT_ADD(3, ft_test); FLT_ADD(4, ft3 - ft2); FLT_ADD(5, ft3 - ft0);
  FLT_ADD(6, n_ball); FLT_ADD(7, n_sphere);
  if (q.stats && lane == 0) {
    atomicAdd(q.stats + 0, (unsigned long long)n_ball);
    atomicAdd(q.stats + 1, (unsigned long long)n_sphere * 64ull * kSpheresPerLane);
  }
}

// one wave per query.  Returns (all lanes) the winner, its squared distance and its closest point; lane 0 writes the
// task's outputs.
__device__ __forceinline__ void surface_resolve(const SurfaceTask& q, int k, double* best_out, int* tri_out, d3* cp_out) {
  const int n = q.cnt[k];
  const d3 p = ld3(q.P + 3 * k);
  const int* list = q.cand + (size_t)k * q.stride;
  double best = __builtin_inf();
  int bi = kNoIndex;
  d3 bc = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};  // closest point of this lane's best candidate
  const int* tri_of = q.spheres ? sphere_triangles(q.spheres, q.T) : q.order;  // the filter names its candidates by their position in the sphere list
  const bool listed = n <= q.stride;  // (uniform) otherwise the list overflowed (a query far from the surface, or without a hint)
  if (listed) {
    for (int i = lane_id(); i < n; i += 64) {
      const int t = tri_of[list[i]];
      d3 c;
      const double d2 = tri_dist2(p, q.verts, q.tris, t, &c);
      if (d2 < best || (d2 == best && t < bi)) { best = d2; bi = t; bc = c; }  // NaN (degenerate triangle) never wins
    }
  } else {
    // every triangle is considered, through its bounding sphere first: a triangle can only win (or tie) if
    // |p − centre| − R <= the smaller of the query's bound and the best distance this lane has seen — the spheres contain
    // their triangles with room to spare (tri_sphere), the comparison is relaxed by 1e-6 relative on top, so what is skipped
    // is strictly farther than the winner: the same lexicographic minimum as a scan of all exact distances
    const double thr = q.thrA ? (double)q.thrA[k] : __builtin_inf();  // (inflated distance to the hinted triangle; +inf without a hint)
    double lim2 = thr * thr;
    for (int pos = lane_id(); pos < q.T; pos += 64) {
      const float4 s = q.spheres ? q.spheres[pos] : tri_sphere(q.verts, q.tris, q.order[pos]);
      const double dx = p.x - (double)s.x, dy = p.y - (double)s.y, dz = p.z - (double)s.z;
      const double dc = sqrt(dx * dx + dy * dy + dz * dz) - (double)s.w;
      const double lb = dc > 0.0 ? dc * dc * (1.0 - 1e-6) : 0.0;
      if (lb > lim2) continue;  // (NaN spheres fall through to the exact evaluation)
      const int t = tri_of[pos];
      d3 c;
      const double d2 = tri_dist2(p, q.verts, q.tris, t, &c);
      if (d2 < best || (d2 == best && t < bi)) { best = d2; bi = t; bc = c; if (d2 < lim2) lim2 = d2; }
    }
  }
  const double my_best = best;
  const int my_bi = bi;
  wave_lexmin(best, bi);
  // the winner's closest point sits in the lane that evaluated it (same triangle, same point: the same value a
  // re-evaluation would give) — fetch it from there instead of walking triangle -> vertices -> Ericson again
  d3 c = {__builtin_nan(""), __builtin_nan(""), __builtin_nan("")};
  if (bi != kNoIndex) {
    const unsigned long long owners = __ballot(my_bi == bi && my_best == best);
    const int src = __ffsll((long long)owners) - 1;
    c.x = __shfl(bc.x, src, 64); c.y = __shfl(bc.y, src, 64); c.z = __shfl(bc.z, src, 64);
  }
  if (lane_id() == 0) {
    if (q.cp) { q.cp[3 * k] = c.x; q.cp[3 * k + 1] = c.y; q.cp[3 * k + 2] = c.z; }
    if (q.d2) q.d2[k] = best;
    if (q.tri) q.tri[k] = bi == kNoIndex ? -1 : bi;
    if (q.hint) q.hint[k] = bi == kNoIndex ? -1 : bi;
    if (q.stats) atomicAdd(q.stats + 2, (unsigned long long)(listed ? n : q.T));  // (an overflowed list: at most T, through the spheres)
  }
  *best_out = best;
  *tri_out = bi;
  *cp_out = c;
}

// ---------------------------------------------------------------- K3 nearest vertex

// `e` = position of the hinted vertex h (q.verts[h]), `have` = the hint is valid
__device__ __forceinline__ void vertex_init_at(const VertexTask& q, int k, bool have, d3 e) {
  if (k >= q.Kpad) return;
  q.cnt[k] = 0;
  if (k >= q.K) { q.thr2[k] = -1.0; return; }  // sentinel: nothing passes
  double d2 = __builtin_inf();
  if (have) {
    d3 d = sub(ld3(q.P + 3 * k), e);
    d2 = dot(d, d);
  }
  if (!(d2 == d2)) d2 = __builtin_inf();
  q.thr2[k] = d2;  // squared bound, same expression as the filter -> the hint vertex itself always passes
}
__device__ __forceinline__ void vertex_init(const VertexTask& q, int k) {
  if (k >= q.Kpad) return;
  int h = (q.hint && k < q.K) ? q.hint[k] : -1;
  const bool have = h >= 0 && h < q.V;
  d3 e = {0.0, 0.0, 0.0};
  if (have) e = ld3(q.verts + 3 * h);
  vertex_init_at(q, k, have, e);
}

// bound of query k, as vertex_init_at stores it: squared distance to the previous winner (−1 for a sentinel slot)
__device__ __forceinline__ double vertex_bound(const VertexTask& q, int k) {
  if (k >= q.K) return -1.0;
  const int h = q.hint ? q.hint[k] : -1;
  double d2 = __builtin_inf();
  if (h >= 0 && h < q.V) {
    d3 d = sub(ld3(q.P + 3 * k), ld3(q.verts + 3 * h));
    d2 = dot(d, d);
  }
  if (!(d2 == d2)) d2 = __builtin_inf();
  return d2;
}

// The workgroup's query chunk goes through LDS (point + bound per query, every lane then reads the same record).  With
// q.thr2 == nullptr the bounds are computed here from the hints instead of being read — the merged step does that: the
// searched vertices (the new instance) are complete when this launch starts, whereas its first launch would have to
// compute every hinted vertex a second time.
__device__ __forceinline__ void vertex_filter(const VertexTask& q, int bx, int by) {
  __shared__ double s_vq[kFilterTile][4];
  __shared__ float4 s_vf[kFilterTile];  // the same queries in f32 for the ball test: point, and the bound's root rounded up
  const int v = bx * kSearchBlock + threadIdx.x;
  const bool valid = v < q.V;
  d3 e = {0.0, 0.0, 0.0};
  if (valid) e = ld3(q.verts + 3 * v);
  // Two levels, as the surface filter (round 6; until then every (vertex, query) pair was tested in f64 — 33 M pairs per chain and
  // step of the face configuration).  A wave holds 64 CONSECUTIVE vertices — a stretch of the mesh: files list neighbouring vertices
  // together, a grid lists a row — inside their axis-aligned BOX [lo, hi] (f32, widened by the slack of both tests' roundings: what
  // passes the exact test below passes this one); a query can only have a candidate in the wave if its distance to the box is within
  // sqrt(bound).  (A box, not a ball: a row of a grid is 64 spacings long and one wide — its ball holds 3,600 grid points, its box
  // widened by a bound of one or two spacings a few hundred.)  That test runs for 64 queries at a time, one per lane; the exact f64
  // test — every lane its vertex against ONE query — only for the survivors.  Every pair is still decided: the candidate set is the one
  // the plain double loop gives.
  const int lane = threadIdx.x & 63;
  const bool fin = valid && fabs(e.x) + fabs(e.y) + fabs(e.z) <= 3.0e38;  // (a non-finite vertex passes no exact test against a finite bound, and must not spoil the box)
  const float big = 3.0e38f;
  float lox = fin ? (float)e.x : big, loy = fin ? (float)e.y : big, loz = fin ? (float)e.z : big;
  float hix = fin ? (float)e.x : -big, hiy = fin ? (float)e.y : -big, hiz = fin ? (float)e.z : -big;
  for (int o = 32; o > 0; o >>= 1) {
    lox = fminf(lox, __shfl_xor(lox, o, 64)); loy = fminf(loy, __shfl_xor(loy, o, 64)); loz = fminf(loz, __shfl_xor(loz, o, 64));
    hix = fmaxf(hix, __shfl_xor(hix, o, 64)); hiy = fmaxf(hiy, __shfl_xor(hiy, o, 64)); hiz = fmaxf(hiz, __shfl_xor(hiz, o, 64));
  }
  const bool wave_live = lox <= hix;  // (uniform: at least one finite vertex)
  {  // slack: a few hundred ulps of the largest magnitudes involved (the vertices' rounding to f32, the test's own errors)
    const float sl = 2e-5f * (fmaxf(fabsf(lox), fabsf(hix)) + fmaxf(fabsf(loy), fabsf(hiy)) + fmaxf(fabsf(loz), fabsf(hiz)));
    lox -= sl; loy -= sl; loz -= sl; hix += sl; hiy += sl; hiz += sl;
  }
  // (a wave that holds a non-finite vertex beside finite ones: that vertex can only pass against an INFINITE bound, whose query passes
  // the box test whatever the box — handled by the exact test as before)
  const int k0 = by * q.kchunk;
  const int k1 = min(q.Kpad, k0 + q.kchunk);
  ParkedHits ph;
  const int first = bx * kSearchBlock + (threadIdx.x & ~63);
  unsigned n_exact = 0;  // (wave-uniform; reported only when profiling)
  for (int kt = k0; kt < k1; kt += kFilterTile) {
    const int nq = min(kFilterTile, k1 - kt);  // multiple of kQU
    if (kt != k0) __syncthreads();
    if ((int)threadIdx.x < nq) {
      const int k = kt + threadIdx.x, kk = min(k, q.K - 1);  // sentinel slots re-read the last real query; their bound is -1
      const double px = q.P[3 * kk], py = q.P[3 * kk + 1], pz = q.P[3 * kk + 2];
      const double b2 = q.thr2 ? q.thr2[k] : vertex_bound(q, k);
      s_vq[threadIdx.x][0] = px; s_vq[threadIdx.x][1] = py; s_vq[threadIdx.x][2] = pz;
      s_vq[threadIdx.x][3] = b2;
      // (a sentinel's bound −1 has no root: NaN, which passes no comparison; +inf — no usable hint — passes every one)
      s_vf[threadIdx.x] = make_float4((float)px, (float)py, (float)pz, round_up_f32(sqrt(b2) * (1.0 + 2e-6)));
    }
    __syncthreads();
    if (!wave_live) continue;
    for (int g = 0; g < nq; g += 64) {
      bool near = false;
      if (g + lane < nq) {  // lane l: query g + l against the wave's box
        const float4 qq = s_vf[g + lane];
        const float dx = fmaxf(fmaxf(lox - qq.x, qq.x - hix), 0.f), dy = fmaxf(fmaxf(loy - qq.y, qq.y - hiy), 0.f),
                    dz = fmaxf(fmaxf(loz - qq.z, qq.z - hiz), 0.f);
        const float lim = qq.w + 2e-5f * (fabsf(qq.x) + fabsf(qq.y) + fabsf(qq.z));
        near = dx * dx + dy * dy + dz * dz <= lim * lim;  // (a NaN query or a sentinel's NaN bound: false; an infinite bound: true)
      }
      unsigned long long todo = __ballot(near);
      n_exact += (unsigned)__popcll(todo);
      while (todo) {  // wave-uniform: the surviving queries, two at a time (their records' LDS reads in flight together)
        const int j0 = g + __ffsll((long long)todo) - 1;
        todo &= todo - 1;
        const bool two = todo != 0ull;
        const int j1 = two ? g + __ffsll((long long)todo) - 1 : j0;
        todo &= todo - 1;  // (0 & anything stays 0)
        const d3 p0 = {s_vq[j0][0], s_vq[j0][1], s_vq[j0][2]}, p1 = {s_vq[j1][0], s_vq[j1][1], s_vq[j1][2]};
        const double t0 = s_vq[j0][3], t1 = s_vq[j1][3];
        const d3 d0 = sub(p0, e), d1 = sub(p1, e);
        const double a0 = dot(d0, d0), a1 = dot(d1, d1);  // (dx·dx + dy·dy) + dz·dz, unfused — the value the argmin is defined on
        const unsigned long long m0 = __ballot(valid && a0 <= t0);
        const unsigned long long m1 = two ? __ballot(valid && a1 <= t1) : 0ull;
        if (m0 != 0ull) {
          park_hits(ph, kt + j0, m0, 0ull);
          if (ph.n == 64) settle_hits(ph, q.cnt, q.cand, q.stride, first, 1);
        }
        if (m1 != 0ull) {
          park_hits(ph, kt + j1, m1, 0ull);
          if (ph.n == 64) settle_hits(ph, q.cnt, q.cand, q.stride, first, 1);
        }
      }
    }
  }
  settle_hits(ph, q.cnt, q.cand, q.stride, first, 1);
  if (q.stats && lane == 0) atomicAdd(q.stats + 3, 64ull * (unsigned long long)n_exact);
}

__device__ __forceinline__ void vertex_resolve(const VertexTask& q, int k, double* best_out, int* idx_out) {
  const int n = q.cnt[k];
  const d3 p = ld3(q.P + 3 * k);
  const int* list = q.cand + (size_t)k * q.stride;
  double best = __builtin_inf();
  int bi = kNoIndex;
  const bool listed = n <= q.stride;  // (uniform) otherwise the list overflowed: every vertex is looked at
  for (int i = lane_id(); i < (listed ? n : q.V); i += 64) {
    const int v = listed ? list[i] : i;
    d3 d = sub(p, ld3(q.verts + 3 * v));
    const double d2 = dot(d, d);
    if (d2 < best || (d2 == best && v < bi)) { best = d2; bi = v; }
  }
  wave_lexmin(best, bi);
  if (lane_id() == 0) {
    if (q.d2) q.d2[k] = best;
    if (q.idx) q.idx[k] = bi == kNoIndex ? -1 : bi;
    if (q.hint) q.hint[k] = bi == kNoIndex ? -1 : bi;
    if (q.stats) atomicAdd(q.stats + 4, (unsigned long long)(listed ? n : q.V));
  }
  *best_out = best;
  *idx_out = bi;
}

// K1: one vertex of x = s(R(x̄ + μ + Q c − ctr) + ctr + t); Qp = scaled basis in planes [(j*3+d)*N + i].
// Summed in basis order with separately rounded multiply and add (the value every search index is defined on).
template <int kU>
__device__ __forceinline__ void instance_batch(const double* __restrict__ q, int N, int r, const double* coeffs, int& j, double& a0,
                                               double& a1, double& a2) {
  for (; j + kU <= r; j += kU) {
    double v[3 * kU];
#pragma unroll
    for (int u = 0; u < 3 * kU; ++u) v[u] = q[(size_t)(3 * j + u) * N];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const double c = coeffs[j + u];
      a0 = a0 + v[3 * u] * c;
      a1 = a1 + v[3 * u + 1] * c;
      a2 = a2 + v[3 * u + 2] * c;
    }
  }
}
// the rigid part of instance_point: x̄ + deformation -> posed point
__device__ __forceinline__ d3 instance_pose(int i, const double* __restrict__ ref, const Pose& pose, double a0, double a1, double a2) {
  double u0 = ref[3 * i] + a0, u1 = ref[3 * i + 1] + a1, u2 = ref[3 * i + 2] + a2;
  double v0 = u0 - pose.ctr[0], v1 = u1 - pose.ctr[1], v2 = u2 - pose.ctr[2];
  double w0 = (pose.R[0] * v0 + pose.R[1] * v1) + pose.R[2] * v2;
  double w1 = (pose.R[3] * v0 + pose.R[4] * v1) + pose.R[5] * v2;
  double w2 = (pose.R[6] * v0 + pose.R[7] * v1) + pose.R[8] * v2;
  return {pose.s * ((w0 + pose.ctr[0]) + pose.t[0]), pose.s * ((w1 + pose.ctr[1]) + pose.t[1]), pose.s * ((w2 + pose.ctr[2]) + pose.t[2])};
}
__device__ __forceinline__ d3 instance_point(int i, int N, int r, const double* __restrict__ Qp, const double* __restrict__ ref,
                                             const double* __restrict__ mean, const Pose& pose, const double* coeffs) {
  double a0 = mean[3 * i], a1 = mean[3 * i + 1], a2 = mean[3 * i + 2];
  const double* q = Qp + i;
  int j = 0;
  // the loop is bound by the latency of its loads (rows of Qp, 8·N bytes apart): 75, then 30 of them in flight per batch;
  // the sums keep their order
  instance_batch<25>(q, N, r, coeffs, j, a0, a1, a2);
  instance_batch<10>(q, N, r, coeffs, j, a0, a1, a2);
  for (; j < r; ++j) {
    double c = coeffs[j];
    a0 = a0 + q[(size_t)(3 * j) * N] * c;
    a1 = a1 + q[(size_t)(3 * j + 1) * N] * c;
    a2 = a2 + q[(size_t)(3 * j + 2) * N] * c;
  }
  return instance_pose(i, ref, pose, a0, a1, a2);
}
__device__ __forceinline__ void instance_vertex(int i, int N, int r, const double* __restrict__ Qp, const double* __restrict__ ref,
                                                const double* __restrict__ mean, const Pose& pose, const double* coeffs,
                                                double* __restrict__ x) {
  const d3 p = instance_point(i, N, r, Qp, ref, mean, pose, coeffs);
  x[3 * i] = p.x; x[3 * i + 1] = p.y; x[3 * i + 2] = p.z;
}
// the same, and the point's deformation mean + Q·c kept (`defo`): a state that differs from this one in its pose only is
// instance_pose of the kept deformation — the same operations on the same values, without the 24·r bytes of basis per point
__device__ __forceinline__ void instance_vertex_keep(int i, int N, int r, const double* __restrict__ Qp, const double* __restrict__ ref,
                                                     const double* __restrict__ mean, const Pose& pose, const double* coeffs,
                                                     double* __restrict__ x, double* __restrict__ defo) {
  double a0 = mean[3 * i], a1 = mean[3 * i + 1], a2 = mean[3 * i + 2];
  const double* q = Qp + i;
  int j = 0;
  instance_batch<25>(q, N, r, coeffs, j, a0, a1, a2);
  instance_batch<10>(q, N, r, coeffs, j, a0, a1, a2);
  for (; j < r; ++j) {
    double c = coeffs[j];
    a0 = a0 + q[(size_t)(3 * j) * N] * c;
    a1 = a1 + q[(size_t)(3 * j + 1) * N] * c;
    a2 = a2 + q[(size_t)(3 * j + 2) * N] * c;
  }
  defo[3 * i] = a0; defo[3 * i + 1] = a1; defo[3 * i + 2] = a2;
  const d3 p = instance_pose(i, ref, pose, a0, a1, a2);
  x[3 * i] = p.x; x[3 * i + 1] = p.y; x[3 * i + 2] = p.z;
}

}  // namespace icp
