// kernels_step.hip — the device work of ONE Metropolis–Hastings step as five dependent launches (gfx950).
//
// A step needs, for a NEW state θ' (SURVEY.md §3.1): the instance, one search per ICP direction + the likelihood
// queries, the correspondences, one GP regression per direction, and the transition-density tails.  Issued stage by
// stage that is ~25 small launches plus ~20 staging copies, each a 2.5–5 µs dependent boundary on an otherwise idle
// chip; the step is bound by those boundaries, not by bandwidth or arithmetic.  Here the stages that do not depend on
// each other share a launch, inputs arrive as kernel arguments and results are written straight into pinned host
// memory, so a step is five boundaries and one synchronisation:
//
//   1 k_step_begin       [a8 propose: c' from the cached posterior of the current state] -> x(θ') -> per-query bounds
//   2 k_step_filter      every search of the step: candidate lists (searched sets streamed once)
//   3 k_step_resolve     exact resolve per query (one wave) -> correspondence record (normal, inverse pose, boundary)
//   4 k_step_regression  f64-MFMA normal-equation partial sums of every posterior + the likelihood reduction
//   5 k_step_finish      per posterior: Cholesky + α, then the backward tail; the forward tails run beside them
//
// The device bodies are the ones the per-stage kernels use (icp_search.hpp, icp_dense.hpp): results are bit-identical
// to the per-method entry points.
//
// Launches 1-4 of a step do not depend on launch 5 of the step before it: the host side (icp_abi.hip, enqueue_front)
// alternates steps between two streams and issues 1-4 of the next step ahead, under the assumption that the step in
// flight is rejected.  The cross-stream order is taken on the device: launch 1 waits for a word that launch 5 of the
// previous step raises when it starts (StepBeginArgs::wait_flag), and for the completion word of the eigen-decomposition
// it draws from.
#include <algorithm>
#include <cstddef>
#include <cstdlib>
#include <cstring>

#include "icp_kernels.hpp"
#include "icp_search.hpp"
#include "icp_dense.hpp"
#include "../../include/icp_sincos.h"

namespace icp {

namespace {

constexpr int kStepBlock = 256;
inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- 1: coefficients -> instance -> query bounds

__device__ __forceinline__ void step_begin_body(const StepBeginArgs& a, const int bx) {
  __shared__ double s_c[512];
  const int tid = threadIdx.x, r = a.r;
  const double* zsrc = r <= kStepInlineZ ? a.zin : a.z_ptr;
  if (a.wait_flag || a.wait2_flag) {  // see StepBeginArgs
    if (tid == 0) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {
        const bool ok1 = !a.wait_flag || __hip_atomic_load(a.wait_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - a.wait_seq >= 0;
        const bool ok2 = !a.wait2_flag || __hip_atomic_load(a.wait2_flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - a.wait2_seq >= 0;
        if (ok1 && ok2) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000) {
          __hip_atomic_store(a.wait_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
      if (a.wait_ticks && bx == 0) atomicAdd((unsigned long long*)a.wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0));
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (invalidate only: nothing of ours to write back)
  }
  const bool inst = bx < a.inst_blocks;
  const int i = bx * kStepBeginPoints + tid;
  const bool my_point = inst && tid < kStepBeginPoints && i < a.N;
  // what does not depend on the new coefficients is fetched first, so that its latency runs beside the proposal
  // arithmetic: the corners of the triangle that bounds this thread's query
  HintTriangle ht{false, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  const bool my_query = my_point && a.has_surf && i < a.surf.K;
  if (my_query) ht = load_hint_triangle(a.surf, i);
  if (a.propose) {
    ProposeIn in = a.prop;
    in.z = zsrc;
    propose_body(r, in, s_c, a.tpr_log2);  // NonRigidIcpProposal.scala:53-62 (every workgroup computes its own copy)
  } else {
    for (int j = tid; j < r; j += kStepBlock) s_c[j] = zsrc[j];
    __syncthreads();
  }
  if (bx == 0)
    for (int j = tid; j < r; j += kStepBlock) {
      const double c = s_c[j];
      for (int o = 0; o < a.n_out; ++o) a.out[o][j] = c;
    }
  if (inst) {
    if (my_point) {  // ModelFittingParameters.scala:108-110
      const d3 p = instance_point(i, a.N, r, a.Qp, a.ref, a.mean, a.pose, s_c);
      a.x[3 * i] = p.x; a.x[3 * i + 1] = p.y; a.x[3 * i + 2] = p.z;
      if (my_query) surface_init_with(a.surf, i, p, ht);  // query i = model point i (:96)
    }
    if (a.has_surf && bx == 0 && tid < kQU) {
      const int k = a.surf.K + tid;  // sentinel slots
      if (k < a.surf.Kpad) surface_init_at(a.surf, k, d3{0.0, 0.0, 0.0});
    }
  } else {
    // TargetSampling queries, the evaluator's target -> model queries: only the candidate counters are reset here; their bounds
    // (distance to the previous winner at its NEW position) are taken by the filter launch, when the new instance is complete
    const int k = (bx - a.inst_blocks) * kStepBlock + tid;
    if (a.has_vert && k < a.vert.Kpad) a.vert.cnt[k] = 0;
    if (k < a.n_zero2) a.zero2[k] = 0;
  }
}
// The same for ranks 32..RMAX with everything the step does NOT depend on held in registers before the wait: the point's
// 3·r basis values (the instance is bound by the latency of those loads, 24·r bytes per point through one CU's L2 port),
// the thread's share of P = (G + σ²I)⁻¹ and the posterior mean.  What is left behind the wait — for the word of the
// decomposition the proposal draws from, on the accepted path — is the product with the fresh basis V, two reductions and
// arithmetic on registers: ≈ 3 µs instead of ≈ 10.  Same operations in the same order as propose_body / instance_point
// (for these ranks block_matvec runs with four threads per row and one pass): bit-identical results.
template <int RMAX>
__device__ __forceinline__ void step_begin_body_reg(const StepBeginArgs& a, const int bx) {
  __shared__ double s_c[64], s_px[64], s_py[64], s_pw[64];
  const int tid = threadIdx.x, r = a.r;
  const double* zsrc = a.zin;  // (r <= kStepInlineZ)
  const bool inst = bx < a.inst_blocks;
  const int i = bx * kStepBeginPoints + tid;
  const bool my_point = inst && tid < kStepBeginPoints && i < a.N;
  const int row = tid >> 2, sub = tid & 3;
  const ProposeIn& in = a.prop;
  // ---- before the wait: immutable model data
  double qv[3 * RMAX];
  double m0 = 0.0, m1 = 0.0, m2 = 0.0;
  if (my_point) {
    const double* q = a.Qp + i;
#pragma unroll
    for (int u = 0; u < 3 * RMAX; ++u) qv[u] = u < 3 * r ? q[(size_t)u * a.N] : 0.0;
    m0 = a.mean[3 * i]; m1 = a.mean[3 * i + 1]; m2 = a.mean[3 * i + 2];
  }
  double pv[16], al = 0.0, isl = 0.0;
  if (a.propose) {
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int j = sub + 4 * t; pv[t] = (row < r && j < r) ? in.P[(size_t)row * r + j] : 0.0; }
    if (tid < r) { al = in.alpha[tid]; isl = in.inv_sqrt_lambda[tid]; }
  }
  if (a.wait_flag || a.wait2_flag) {  // see StepBeginArgs
    if (tid == 0) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {
        const bool ok1 = !a.wait_flag || __hip_atomic_load(a.wait_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.wait_seq >= 0;
        const bool ok2 = !a.wait2_flag || __hip_atomic_load(a.wait2_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - a.wait2_seq >= 0;
        if (ok1 && ok2) break;
        if (__builtin_amdgcn_s_memrealtime() - t0 > 5000000) {
          __hip_atomic_store(a.wait_error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
          break;
        }
        __builtin_amdgcn_s_sleep(2);
      }
      if (a.wait_ticks && bx == 0) atomicAdd((unsigned long long*)a.wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0));
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (invalidate only: nothing of ours to write back)
  }
  HintTriangle ht{false, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}, {0.0, 0.0, 0.0}};
  const bool my_query = my_point && a.has_surf && i < a.surf.K;
  if (my_query) ht = load_hint_triangle(a.surf, i);
  if (a.propose) {  // NonRigidIcpProposal.scala:53-62 (every workgroup computes its own copy)
    double vv[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) { const int j = sub + 4 * t; vv[t] = (row < r && j < r) ? in.V[(size_t)row * r + j] : 0.0; }
    if (tid < r) s_px[tid] = sqrt(in.S[tid]) * zsrc[tid];
    __syncthreads();
    {
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) { const int j = sub + 4 * t; if (j < r) acc = fma(vv[t], s_px[j], acc); }
      acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
      if (row < r && sub == 0) s_py[row] = acc;
    }
    __syncthreads();
    if (tid < r) s_pw[tid] = fma(s_py[tid], isl, al);
    __syncthreads();
    {
      double acc = 0.0;
#pragma unroll
      for (int t = 0; t < 16; ++t) { const int j = sub + 4 * t; if (j < r) acc = fma(pv[t], s_pw[j], acc); }
      acc += __shfl_xor(acc, 2, 64); acc += __shfl_xor(acc, 1, 64);
      if (row < r && sub == 0) s_py[row] = acc;
    }
    __syncthreads();
    if (tid < r) {
      const double cnew = fma(-in.sigma2, s_py[tid], s_pw[tid]);  // model.coefficients(...) with σ² = 1e-5 (:59)
      const double c0 = in.c[tid];
      s_c[tid] = c0 + (cnew - c0) * in.step;                       // :61-62
    }
    __syncthreads();
  } else {
    if (tid < r) s_c[tid] = zsrc[tid];
    __syncthreads();
  }
  if (bx == 0 && tid < r) {
    const double c = s_c[tid];
    for (int o = 0; o < a.n_out; ++o) a.out[o][tid] = c;
  }
  if (inst) {
    if (my_point) {  // ModelFittingParameters.scala:108-110, summed in basis order with separately rounded multiply and add
      double a0 = m0, a1 = m1, a2 = m2;
#pragma unroll
      for (int j = 0; j < RMAX; ++j)
        if (j < r) {
          const double c = s_c[j];
          a0 = a0 + qv[3 * j] * c; a1 = a1 + qv[3 * j + 1] * c; a2 = a2 + qv[3 * j + 2] * c;
        }
      const d3 p = instance_pose(i, a.ref, a.pose, a0, a1, a2);
      a.x[3 * i] = p.x; a.x[3 * i + 1] = p.y; a.x[3 * i + 2] = p.z;
      if (my_query) surface_init_with(a.surf, i, p, ht);  // query i = model point i (:96)
    }
    if (a.has_surf && bx == 0 && tid < kQU) {
      const int k = a.surf.K + tid;  // sentinel slots
      if (k < a.surf.Kpad) surface_init_at(a.surf, k, d3{0.0, 0.0, 0.0});
    }
  } else {
    const int k = (bx - a.inst_blocks) * kStepBlock + tid;
    if (a.has_vert && k < a.vert.Kpad) a.vert.cnt[k] = 0;
    if (k < a.n_zero2) a.zero2[k] = 0;
  }
}
__device__ __forceinline__ int step_begin_grid(const StepBeginArgs& a) {
  const int nz = max(a.has_vert ? a.vert.Kpad : 0, a.n_zero2);
  return a.inst_blocks + (nz + kStepBlock - 1) / kStepBlock;
}

__global__ void __launch_bounds__(kStepBlock) k_step_begin(StepBeginArgs a) { step_begin_body(a, blockIdx.x); }
template <int RMAX>
__global__ void __launch_bounds__(kStepBlock) k_step_begin_reg(StepBeginArgs a) { step_begin_body_reg<RMAX>(a, blockIdx.x); }

// ---------------------------------------------------------------- 2: filters of every search
#ifdef ICP_FILTER_STAMPS
extern "C" __attribute__((visibility("default"))) void icp_debug_filter_stamps(unsigned long long* out, int reset) {
  (void)hipDeviceSynchronize();
  static unsigned long long h[kFltSlots][8];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_filter_stamps), sizeof(h));
  for (int i = 0; i < 8; ++i) { out[i] = 0; for (int s = 0; s < kFltSlots; ++s) out[i] += h[s][i]; }
  if (reset) { std::memset(h, 0, sizeof(h)); (void)hipMemcpyToSymbol(HIP_SYMBOL(g_filter_stamps), h, sizeof(h)); }
}
#endif

template <bool kPrepared = false>
__device__ __forceinline__ void step_filter_body(const StepSearchArgs& a, const int b) {
  int task = 0;
  while (task + 1 < a.n_surf + a.n_vert && b >= a.fstart[task + 1]) ++task;
  const int l = b - a.fstart[task];
  const int ksplit = task < a.n_surf ? a.s[task].ksplit : a.v[task - a.n_surf].ksplit;
  const int bx = l / (8 * ksplit) * 8 + (l & 7), by = (l % (8 * ksplit)) >> 3;  // see filter_grid_blocks
  if (task < a.n_surf) {
    if (bx < a.s[task].tblocks) surface_filter<kPrepared>(a.s[task], bx, by);
  } else {
    if (bx < a.v[task - a.n_surf].vblocks) vertex_filter(a.v[task - a.n_surf], bx, by);
  }
}

__global__ void __launch_bounds__(kSearchBlock) k_step_filter(StepSearchArgs a) { step_filter_body(a, blockIdx.x); }

// ---------------------------------------------------------------- 3: resolve + correspondence record (one wave per query)

__device__ __forceinline__ void resolve_surface_query(const SurfaceTask& q, int k, int ci, const StepSearchArgs& a) {
  double best; int tri; d3 cp;
  surface_resolve(q, k, &best, &tri, &cp);
  if (ci == 0 && k < a.corr[0].K && tri != kNoIndex) correspond_model_wave(a.corr[0], k, cp);
  if (ci == 1 && k < a.corr[1].K && tri != kNoIndex) correspond_model_wave(a.corr[1], k, cp);
}
__device__ __forceinline__ void resolve_vertex_query(const VertexTask& q, int k, int ci, const StepSearchArgs& a) {
  double best; int idx;
  vertex_resolve(q, k, &best, &idx);
  if (ci == 0 && k < a.corr[0].K && idx != kNoIndex) correspond_target_wave(a.corr[0], k, idx);
  if (ci == 1 && k < a.corr[1].K && idx != kNoIndex) correspond_target_wave(a.corr[1], k, idx);
}

__device__ __forceinline__ void step_resolve_body(const StepSearchArgs& a, const int b) {
  if (a.n_surf > 0 && b < a.rstart[1]) resolve_surface_query(a.s[0], b - a.rstart[0], a.s_corr[0], a);
  else if (a.n_surf > 1 && b < a.rstart[2]) resolve_surface_query(a.s[1], b - a.rstart[1], a.s_corr[1], a);
  else if (a.n_vert > 0 && b < a.rstart[a.n_surf + 1]) resolve_vertex_query(a.v[0], b - a.rstart[a.n_surf], a.v_corr[0], a);
  else if (a.n_vert > 1) resolve_vertex_query(a.v[1], b - a.rstart[a.n_surf + 1], a.v_corr[1], a);
}

__global__ void __launch_bounds__(64) k_step_resolve(StepSearchArgs a) { step_resolve_body(a, blockIdx.x); }

// ---------------------------------------------------------------- 4: regression partial sums + likelihood reduction

__device__ __forceinline__ void dist_stats_body(int K, const double* __restrict__ d2, double* __restrict__ out) {
  __shared__ double s_red_d[16];
  double sum = 0.0, mx = -__builtin_inf(), cnt = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    const double d = sqrt(d2[k]);
    sum += d;
    mx = fmax(mx, d);
    cnt += 1.0;
  }
  sum = block_sum(sum, s_red_d);
  cnt = block_sum(cnt, s_red_d);
  mx = block_max(mx, s_red_d);
  if (threadIdx.x == 0) { out[0] = sum; out[1] = mx; out[2] = cnt; }
}

// Units and XCDs.  Workgroups are dealt to the eight XCDs round robin by their index, and every XCD has an L2 of its own: with the
// units in their natural order — the tiles of a split side by side — a split's gathered basis rows were fetched into seven or eight
// L2s (configs[2], K = 1,622, rank 100: 23 MB fetched per launch for 3.9 MB of rows, profiles/r04_pmc_traffic.json).  The
// workgroups of one XCD (index ≡ x mod 8) therefore take a CONTIGUOUS range of units — whole splits: `step_regression_blocks`
// workgroups (a multiple of 8), workgroup bx works as logical block (bx mod 8)·per + bx div 8.
__host__ __device__ inline int step_regression_blocks(int n_units) { return 8 * ((((n_units + 3) >> 2) + 7) >> 3); }
template <bool kFolded = false>
__device__ __forceinline__ void step_regression_body(const StepRegressionArgs& a, const int bx) {
  const int n_units = a.ustart[a.n];
  const int n_blocks = (n_units + 3) >> 2;  // one wave per (tile, split) unit, four per workgroup
  const int n_grid = step_regression_blocks(n_units);
  if (bx < n_grid) {
    const int lb = (bx & 7) * (n_grid >> 3) + (bx >> 3);
    const int u = lb * 4 + (threadIdx.x >> 6);
    if (lb < n_blocks && u < n_units) {
      const int which = u < a.ustart[1] ? 0 : 1;
      const int l = u - (which ? a.ustart[1] : 0), tile = l % a.ntiles, split = l / a.ntiles;
      if (l == 0 && (threadIdx.x & 63) == 0) { a.status[which][1] = 0; a.status[which][2] = 0; }
      if constexpr (kFolded) {
        // (a kernel of its own: inside the plain one these bodies' registers — two accumulator sets per tile, four tiles — took its occupancy
        // from 152 to 328 registers per lane, and every lone chain's regression with it: configs[2] −7 %, 64 femur chains −8 %)
        if (a.macro[which] > 1) regression_macro_fold<2>(l, a.fold[which], a.K[which], a.kchunk[which], a.r, a.Q, a.cb[which], a.wt[which], a.kappa[which], a.Mpart[which]);
        else regression_tile_fold(tile, split, a.K[which], a.kchunk[which], a.r, a.Q, a.cb[which], a.wt[which], a.kappa[which], a.Mpart[which], a.fold[which]);
      } else if (which == 0) regression_tile(tile, split, a.K[0], a.kchunk[0], a.r, a.Q, a.cb[0], a.wt[0], a.kappa[0], a.Mpart[0]);
      else regression_tile(tile, split, a.K[1], a.kchunk[1], a.r, a.Q, a.cb[1], a.wt[1], a.kappa[1], a.Mpart[1]);
    }
  } else {
    // the last workgroup: likelihood reduction over the surface distances of the evaluator's model ids (results 0-3) and, for a
    // TargetToModel / Symmetric evaluator, of its target points against the new model surface (results 4-7)
    if (a.reduce_kind == 1) {
      if (a.Kred > 0) sum_gauss_logpdf_body(a.Kred, a.d2, a.mean, a.sigma, a.red_out);  // IndependentPointDistanceEvaluator.scala:40-46
      if (a.Kred2 > 0) { __syncthreads(); sum_gauss_logpdf_body(a.Kred2, a.d2b, a.mean, a.sigma, a.red_out + 4); }  // :49-54
    } else if (a.reduce_kind == 2) {
      if (a.Kred > 0) dist_stats_body(a.Kred, a.d2, a.red_out);                                    // Collective…Evaluator.scala:43-52 (no boundary)
      if (a.Kred2 > 0) { __syncthreads(); dist_stats_body(a.Kred2, a.d2b, a.red_out + 4); }        // :55-64
    }
  }
}
__device__ __forceinline__ int step_regression_grid(const StepRegressionArgs& a) {
  return step_regression_blocks(a.ustart[a.n]) + (a.reduce_kind ? 1 : 0);
}

__global__ void __launch_bounds__(kStepBlock) k_step_regression(StepRegressionArgs a) { step_regression_body(a, blockIdx.x); }

__global__ void __launch_bounds__(256) k_step_reduce(StepReduceArgs a) {
  const int which = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (which >= a.n || e >= a.nn) return;
  double* P = a.Mpart[which];
  const int S = a.splits[which];
  double acc = 0.0;
  for (int s0 = 0; s0 < S; s0 += 8) {  // eight splits in flight, summed in split order
    double q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = P[(size_t)min(s0 + u, S - 1) * a.nn + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (s0 + u < S) acc += q[u];
  }
  P[e] = acc;
}

// ---------------------------------------------------------------- 5: factorisations + transition tails

template <int E, int NT>
__device__ __forceinline__ void step_finish_body(const StepFinishArgs& a, const int b) {
  if (b == 0 && threadIdx.x == 0 && a.ready_flag) __hip_atomic_store(a.ready_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  if (b < a.n) {
    // (the matrices of the backward tail are put into LDS on the way, behind the factor's own region: a.tail_base)
    const bool ok = factor_reg_body<E, NT>(a.r, a.Mpart[b], a.splits[b], a.M[b], a.alpha[b], a.status[b], a.tail_base, a.Ginv);
    if (threadIdx.x == 0) a.host_status[b][0] = ok ? 0 : 1;
    if (ok) {  // uniform
      __syncthreads();  // M and alpha of this posterior are complete (written by this workgroup)
      const TransitionTailIO& t = a.bwd[b];
      tail_body(a.r, t.alpha, t.M, t.c_from, t.c_to, t.step, t.out, t.status, a.Ginv, a.sigma2, a.n_lds, a.tpr_log2,
                a.tail_base >= 0 ? a.tail_base : 0, a.tail_base >= 0);
    } else if (threadIdx.x == 0) {
      a.bwd[b].out[0] = __builtin_nan("");
      a.bwd[b].status[0] = 0;
    }
  } else {
    const TransitionTailIO& t = a.fwd[b - a.n];
    tail_body(a.r, t.alpha, t.M, t.c_from, t.c_to, t.step, t.out, t.status, a.Ginv, a.sigma2, a.n_lds, a.tpr_log2);
  }
  // every workgroup's results are in pinned host memory: count in, the last one raises the flag
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_system();  // this workgroup's results, before it is counted in
    if (atomicAdd(a.done_counter, 1) == 2 * a.n - 1) {
      // (everybody else's results became visible before their increments, which this one has observed: the flag needs no
      // fence of its own — a second system-scope fence here cost ≈ 1.5 µs at the very end of every step)
      *a.done_counter = 0;
      __hip_atomic_store(a.host_flag, a.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
  }
}

template <int E, int NT>
__global__ void __launch_bounds__(NT) k_step_finish(StepFinishArgs a) { step_finish_body<E, NT>(a, blockIdx.x); }

// ---------------------------------------------------------------- the same five launches for B chains at once
//
// icp_chain_step_batched: blockIdx.y = chain, arguments of chain y = batch[y] in device memory (brought there by
// k_step_batch_args at the head of the sequence); a chain whose grid is smaller than the launch's leaves its surplus
// workgroups idle.  (The record is read in place: a private copy of it lands in scratch, its arrays being indexed at run
// time.)

__global__ void __launch_bounds__(256) k_step_batch_args(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16, StepBatchGate gate) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];  // src: pinned host memory, read in place
  // the gate (see StepBatchGate): ONE thread of the sequence's first kernel waits — a single wave on a single CU keeps nothing
  // from becoming resident — for every workgroup of the batch's decompositions to have started
  if (gate.counter && blockIdx.x == 0 && threadIdx.x == 0) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
    while (__hip_atomic_load(gate.counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - gate.expected < 0) {
      if (__builtin_amdgcn_s_memrealtime() - t0 > 200000000ll) {
        __hip_atomic_store(gate.error, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
  }
}

__global__ void __launch_bounds__(kStepBlock) k_step_begin_batch(const StepBeginArgs* __restrict__ batch) {
  const StepBeginArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= step_begin_grid(a)) return;
  step_begin_body(a, blockIdx.x);
}
template <int RMAX>
__global__ void __launch_bounds__(kStepBlock) k_step_begin_batch_reg(const StepBeginArgs* __restrict__ batch) {
  const StepBeginArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= step_begin_grid(a)) return;
  step_begin_body_reg<RMAX>(a, blockIdx.x);
}
// kPrepared: every surface task of every chain of the batch comes with its spheres and bounds (step_filter_prepared, checked by the
// launcher on the host's copy of the arguments) — see surface_filter
template <bool kPrepared>
__global__ void __launch_bounds__(kSearchBlock, kPrepared ? 8 : 1) k_step_filter_batch(const StepSearchArgs* __restrict__ batch) {
  const StepSearchArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= a.fstart[a.n_surf + a.n_vert]) return;
  step_filter_body<kPrepared>(a, blockIdx.x);
}
__global__ void __launch_bounds__(64) k_step_resolve_batch(const StepSearchArgs* __restrict__ batch) {
  const StepSearchArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= a.rstart[a.n_surf + a.n_vert]) return;
  step_resolve_body(a, blockIdx.x);
}
__global__ void __launch_bounds__(kStepBlock) k_step_regression_batch(const StepRegressionArgs* __restrict__ batch) {
  const StepRegressionArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= step_regression_grid(a)) return;
  step_regression_body(a, blockIdx.x);
}
// … with the posteriors' split-K leaves folded into one partial each (StepRegressionArgs::fold > 1 for the whole launch)
__global__ void __launch_bounds__(kStepBlock) k_step_regression_batch_fold(const StepRegressionArgs* __restrict__ batch) {
  const StepRegressionArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= step_regression_grid(a)) return;
  step_regression_body<true>(a, blockIdx.x);
}
template <int E, int NT>
__global__ void __launch_bounds__(NT) k_step_finish_batch(const StepFinishArgs* __restrict__ batch) {
  const StepFinishArgs& a = batch[blockIdx.y];
  if ((int)blockIdx.x >= 2 * a.n) return;
  step_finish_body<E, NT>(a, blockIdx.x);
}

thread_local StepCapture* t_capture = nullptr;

static void set_dyn_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

struct FinishPlan { int E /* tiles per thread */, NT, n_lds, tail_base; size_t shmem; bool ok; };

FinishPlan finish_plan(int r) {
  FinishPlan p{0, 0, 0, -1, 0, false};
  const int ld = r | 1;
  const size_t tiles = (size_t)factor_tile_count(r);
  const size_t w = (size_t)(r + 1) * ld, one = (size_t)r * ld;
  const size_t budget = (size_t)kLdsDoubles - 4700;  // static LDS of factor_reg_body + tail_body
  if (w > budget) return p;
  if (tiles <= 256) { p.E = 1; p.NT = 256; }
  // (up to 1024 tiles: two per thread on 512 threads — 1024 threads leave 128 VGPRs per lane and the tiles' bookkeeping spilled 156 B
  // of scratch, ISA of round 2; 512 threads have 256)
  else if (tiles <= 1024) { p.E = 2; p.NT = 512; }
  else if (tiles <= 2048) { p.E = 2; p.NT = 1024; }
  else return p;
  p.n_lds = 2 * one <= budget ? 2 : (one <= budget ? 1 : 0);
  const size_t tails = one * p.n_lds;
  // the backward tail's matrices behind the factor's region when both fit (then the factor stages them on its way)
  if (p.n_lds == 2 && w + tails <= budget) { p.tail_base = (int)w; p.shmem = sizeof(double) * (w + tails); }
  else p.shmem = sizeof(double) * (w > tails ? w : tails);
  p.ok = true;
  return p;
}

template <int E, int NT>
void launch_finish(hipStream_t st, const StepFinishArgs& a, size_t shmem) {
  static size_t lds_granted = 0;  // (one runtime call per process and size class instead of one per launch)
  if (shmem > lds_granted) { set_dyn_lds((const void*)k_step_finish<E, NT>, shmem); lds_granted = shmem; }
  hipLaunchKernelGGL((k_step_finish<E, NT>), dim3(2 * a.n), dim3(NT), shmem, st, a);
}

}  // namespace

bool step_finish_supported(int r) { return finish_plan(r).ok; }

void launch_step_begin(hipStream_t st, const StepBeginArgs& a_in) {
  StepBeginArgs a = a_in;
  a.inst_blocks = cdiv(a.N, kStepBeginPoints);
  const int vblocks = cdiv(std::max(a.has_vert ? a.vert.Kpad : 0, a.n_zero2), kStepBlock);
  if (t_capture) { t_capture->begin = a; t_capture->grid[0] = a.inst_blocks + vblocks; return; }
  ProfScope _ps(st, KID_STEP_BEGIN);
  static const bool no_reg = dev_env("ICP_BEGIN_STREAMED") != nullptr;  // (A/B switch)
  if (a.hold_regs && a.r >= 32 && a.r <= 52 && !no_reg) hipLaunchKernelGGL(k_step_begin_reg<52>, dim3(a.inst_blocks + vblocks), dim3(kStepBlock), 0, st, a);
  else if (a.hold_regs && a.r >= 32 && a.r <= 64 && !no_reg) hipLaunchKernelGGL(k_step_begin_reg<64>, dim3(a.inst_blocks + vblocks), dim3(kStepBlock), 0, st, a);
  else hipLaunchKernelGGL(k_step_begin, dim3(a.inst_blocks + vblocks), dim3(kStepBlock), 0, st, a);
}

void launch_step_filter(hipStream_t st, const StepSearchArgs& a) {
  const int grid = a.fstart[a.n_surf + a.n_vert];
  if (t_capture) { t_capture->search = a; t_capture->grid[1] = grid; return; }
  if (grid <= 0) return;
  ProfScope _ps(st, KID_STEP_FILTER);
  hipLaunchKernelGGL(k_step_filter, dim3(grid), dim3(kSearchBlock), 0, st, a);
}

void launch_step_resolve(hipStream_t st, const StepSearchArgs& a) {
  const int grid = a.rstart[a.n_surf + a.n_vert];
  if (t_capture) { t_capture->grid[2] = grid; return; }  // (same arguments as the filter launch)
  if (grid <= 0) return;
  ProfScope _ps(st, KID_STEP_RESOLVE);
  hipLaunchKernelGGL(k_step_resolve, dim3(grid), dim3(64), 0, st, a);
}

void launch_step_regression(hipStream_t st, const StepRegressionArgs& a) {
  const int blocks = step_regression_blocks(a.ustart[a.n]) + (a.reduce_kind ? 1 : 0);
  if (t_capture) { t_capture->regression = a; t_capture->grid[3] = blocks; return; }
  if (blocks <= 0) return;
  ProfScope _ps(st, KID_STEP_REGRESSION);
  hipLaunchKernelGGL(k_step_regression, dim3(blocks), dim3(kStepBlock), 0, st, a);
}

void launch_step_reduce(hipStream_t st, const StepReduceArgs& a) {
  if (t_capture) return;  // (B chains per launch: the finish launch sums the partials itself)
  ProfScope _ps(st, KID_STEP_REGRESSION);
  hipLaunchKernelGGL(k_step_reduce, dim3(cdiv(a.nn, 256), a.n), dim3(256), 0, st, a);
}

void launch_step_finish(hipStream_t st, const StepFinishArgs& a_in) {
  const FinishPlan p = finish_plan(a_in.r);
  StepFinishArgs a = a_in;
  a.n_lds = p.n_lds;
  a.tail_base = p.tail_base;
  a.tpr_log2 = matvec_tpr_log2(a.r, p.NT);
  if (t_capture) { t_capture->finish = a; t_capture->grid[4] = 2 * a.n; return; }
  ProfScope _ps(st, KID_STEP_FINISH);
  if (p.E == 1 && p.NT == 256) launch_finish<1, 256>(st, a, p.shmem);
  else if (p.E == 2 && p.NT == 512) launch_finish<2, 512>(st, a, p.shmem);
  else launch_finish<2, 1024>(st, a, p.shmem);
}


void step_capture(StepCapture* c) { t_capture = c; }

namespace {
template <int E, int NT>
void launch_finish_batch(hipStream_t st, const StepFinishArgs* batch, int gx, int B, size_t shmem) {
  static size_t lds_granted = 0;
  if (shmem > lds_granted) { set_dyn_lds((const void*)k_step_finish_batch<E, NT>, shmem); lds_granted = shmem; }
  hipLaunchKernelGGL((k_step_finish_batch<E, NT>), dim3(gx, B), dim3(NT), shmem, st, batch);
}
inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }
}  // namespace

void launch_step_batch_resident(hipStream_t st, int B, const int gx[5], int r, const StepBeginArgs* begin, const StepSearchArgs* search,
                                const StepRegressionArgs* regression, const StepFinishArgs* finish, bool filter_prepared, bool reg_folded) {
  if (B <= 0) return;
  if (gx[0] > 0) { ProfScope _ps(st, KID_STEP_BEGIN); hipLaunchKernelGGL(k_step_begin_batch, dim3(gx[0], B), dim3(kStepBlock), 0, st, begin); }
  if (gx[1] > 0) {
    ProfScope _ps(st, KID_STEP_FILTER);
    if (filter_prepared) hipLaunchKernelGGL(k_step_filter_batch<true>, dim3(gx[1], B), dim3(kSearchBlock), 0, st, search);
    else hipLaunchKernelGGL(k_step_filter_batch<false>, dim3(gx[1], B), dim3(kSearchBlock), 0, st, search);
  }
  if (gx[2] > 0) { ProfScope _ps(st, KID_STEP_RESOLVE); hipLaunchKernelGGL(k_step_resolve_batch, dim3(gx[2], B), dim3(64), 0, st, search); }
  if (gx[3] > 0) {
    ProfScope _ps(st, KID_STEP_REGRESSION);
    if (reg_folded) hipLaunchKernelGGL(k_step_regression_batch_fold, dim3(gx[3], B), dim3(kStepBlock), 0, st, regression);
    else hipLaunchKernelGGL(k_step_regression_batch, dim3(gx[3], B), dim3(kStepBlock), 0, st, regression);
  }
  if (gx[4] > 0) {
    ProfScope _ps(st, KID_STEP_FINISH);
    const FinishPlan p = finish_plan(r);
    if (p.E == 1 && p.NT == 256) launch_finish_batch<1, 256>(st, finish, gx[4], B, p.shmem);
    else if (p.E == 2 && p.NT == 512) launch_finish_batch<2, 512>(st, finish, gx[4], B, p.shmem);
    else launch_finish_batch<2, 1024>(st, finish, gx[4], B, p.shmem);
  }
}

// ---------------------------------------------------------------- the on-device Metropolis–Hastings loop (see MhChain)

namespace {
__device__ __forceinline__ unsigned long long mh_splitmix64(unsigned long long x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}
// StepRandom::uniform (host/icp_host.hpp) = orc_rng_uniform (the oracle): integer hash -> (0, 1), exact in double
__device__ __forceinline__ double mh_uniform(unsigned long long seed, unsigned long long step, unsigned long long lane) {
  const unsigned long long h = mh_splitmix64(mh_splitmix64(mh_splitmix64(seed) ^ (step * 0xD1342543DE82EF95ull)) ^ (lane * 0x2545F4914F6CDD1Dull));
  return ((double)(h >> 11) + 0.5) * (1.0 / 9007199254740992.0);
}
// first component whose cumulative normalised weight reaches u (Scalismo MixtureProposal; icp_host.hpp: pick_component)
__device__ __forceinline__ int mh_pick(int n, const double* w, double u) {
  double acc = 0.0;
  for (int i = 0; i < n; ++i) {
    acc += w[i];
    if (acc >= u) return i;
  }
  return n - 1;
}
// Scalismo MixtureProposal.logTransitionProbability: log-sum-exp over the components (icp_host.hpp)
__device__ __forceinline__ double mh_lse(int n, const double* w, const double* t) {
  double mx = -__builtin_inf();
  for (int i = 0; i < n; ++i)
    if (t[i] > mx) mx = t[i];
  if (mx == -__builtin_inf()) return mx;
  double s = 0.0;
  for (int i = 0; i < n; ++i) s += w[i] * exp(t[i] - mx);
  return log(s) + mx;
}
template <class T>
__device__ __forceinline__ void mh_copy(T* dst, const T* src, int tid, int nt) {
  static_assert(sizeof(T) % 8 == 0, "argument records are copied in 8-byte words");
  const unsigned long long* s = (const unsigned long long*)src;
  unsigned long long* d = (unsigned long long*)dst;
  for (int i = tid; i < (int)(sizeof(T) / 8); i += nt) d[i] = s[i];
}
}  // namespace

// the step's mixture draw (MixtureProposal.propose: outer draw on lane 0, inner on lane 1), by one thread
__device__ __forceinline__ void mh_draw(MhChain& c, int* gen_out, int* pose_leaf_out) {
  const unsigned long long step = (unsigned long long)c.step;
  const int o = mh_pick(c.n_outer, c.outer_w, mh_uniform(c.seed, step, 0));
  int gen = -1, leaf = 2, pose_leaf = -1;  // the shape random walk (RandomShapeUpdateProposal) = leaf 2
  if (c.outer_kind[o] == 1) {
    gen = mh_pick(c.n_icp, c.icp_w, mh_uniform(c.seed, step, 1));
    leaf = gen;
  } else if (c.outer_kind[o] == 0) {  // mixedRandomPoseProposal: the inner draw picks one of the six walks
    pose_leaf = mh_pick(c.n_pose, c.pose_w, mh_uniform(c.seed, step, 1));
    leaf = 3 + pose_leaf;
  }
  c.gen = gen; c.leaf = leaf; c.pose_move = pose_leaf >= 0 ? 1 : 0;
  *gen_out = gen; *pose_leaf_out = pose_leaf;
}
// the pose the step's kernels work with: the chain's current one — or, for a pose walk, the current one with ONE parameter moved by
// σ·z₀ (PoseProposals.scala:39-41, :72-74).  The rotation matrix from include/icp_sincos.h: the bits the host's pose_from_theta and the
// oracle's orc_rotation_matrix give for these angles.
__device__ __forceinline__ Pose mh_step_pose(MhChain& c, int pose_leaf, double z0) {
  double th[10];
  for (int k = 0; k < 10; ++k) th[k] = c.theta[k];
  if (pose_leaf >= 0) th[c.pose_index[pose_leaf]] = th[c.pose_index[pose_leaf]] + c.pose_sigma[pose_leaf] * z0;
  for (int k = 0; k < 10; ++k) c.prop_pose[k] = th[k];
  Pose P;
  icp_rotation_matrix(th[4], th[5], th[6], P.R);
  for (int d = 0; d < 3; ++d) { P.t[d] = th[1 + d]; P.ctr[d] = th[7 + d]; }
  P.s = th[0];
  return P;
}
// head of a step: mixture draw, the step's arguments.
// One wave per chain; called by k_mh_front (first step of a block of normals) and by k_mh_decide for the step behind its own.
__device__ __forceinline__ void mh_front_body(MhChain& c, const int tid) {
  const int r = c.r;
  __shared__ int s_gen, s_pose_leaf;
  if (tid == 0) mh_draw(c, &s_gen, &s_pose_leaf);
  const int sel = c.cur_sel;
  mh_copy(c.begin_live, c.begin_alt[sel], tid, 64);
  mh_copy(c.search_live, c.search_alt[sel], tid, 64);
  mh_copy(c.regression_live, c.regression_alt[sel], tid, 64);
  mh_copy(c.finish_live, c.finish_alt[sel], tid, 64);
  __threadfence_block();
  __syncthreads();
  const int gen = s_gen;
  const double* z = c.normals + (size_t)(c.step - c.normals_first) * r;
  StepBeginArgs& b = *c.begin_live;
  if (tid == 0) {
    b.propose = gen >= 0 ? 1 : 0;
    if (gen >= 0) b.prop = c.prop_alt[sel][gen];
  }
  const int pose_leaf = s_pose_leaf;
  if (tid < r) {
    // ICP: posterior.sample()'s standard normals (NonRigidIcpProposal.scala:55); shape walk: the sample itself, c + σ·z
    // (RandomShapeUpdateProposal.scala:31-35); pose walk: the coefficients stay
    b.zin[tid] = gen >= 0 ? z[tid] : (pose_leaf >= 0 ? c.theta[10 + tid] : c.theta[10 + tid] + c.rw_sigma * z[tid]);
  }
  if (c.n_pose > 0 && tid == 0) {
    // the step's pose goes where its kernels read it: the instance (launch 1) and the correspondences' inverse rigid transform (launch 3)
    const Pose P = mh_step_pose(c, pose_leaf, z[0]);
    b.pose = P;
    c.search_live->corr[0].pose = P;
    c.search_live->corr[1].pose = P;
  }
}
__global__ void __launch_bounds__(64) k_mh_front(MhChain* __restrict__ chains) {
  MhChain& c = chains[blockIdx.x];
  if (c.error) return;
  mh_front_body(c, threadIdx.x);
}

// … of a chain that takes the wide step (MhWide): its launch records are the same every step but for the proposal's inputs (W1) and the
// proposed state's pose (W2's instance, the correspondences of W5 / W7).  A pose walk's instance is made from the coefficients like any
// other (the operations of the kept deformations: bit-identical points).  Ranks up to 256.
// (restate != 0: no proposal — the "proposed" state is the chain's current one, whose posterior the launches behind this one then
// compute into the proposed state's entries: how a run fills the current states' entries that were not on record, for all chains at once)
__global__ void __launch_bounds__(64) k_mhw_front(MhChain* __restrict__ chains, int restate) {
  MhChain& c = chains[blockIdx.x];
  if (c.error) return;
  const int tid = threadIdx.x, r = c.r;
  MhWide& w = *c.wide;
  __shared__ int s_gen, s_pose_leaf;
  if (tid == 0) {
    if (restate) { s_gen = -1; s_pose_leaf = -2; }
    else mh_draw(c, &s_gen, &s_pose_leaf);
  }
  __syncthreads();
  const int gen = s_gen, pose_leaf = s_pose_leaf == -2 ? 0 : s_pose_leaf;  // (restate: the coefficients stay, as for a pose walk …
  const double* z = restate ? c.theta : c.normals + (size_t)(c.step - c.normals_first) * r;  // (… never read as normals)
  if (tid == 0) {
    WideProposeItem& it = *w.item;
    it.kind = gen >= 0 ? 1 : 0;
    if (gen >= 0) {  // ICP: posterior.sample()'s standard normals (NonRigidIcpProposal.scala:55)
      ProposeIn in = w.prop_in[gen];
      in.z = z;
      it.in = in;
    }
    it.src = w.given;
  }
  // shape walk: the sample itself, c + σ·z (RandomShapeUpdateProposal.scala:31-35); pose walk: the coefficients stay
  for (int j = tid; j < r; j += 64) w.given[j] = pose_leaf >= 0 ? c.theta[10 + j] : c.theta[10 + j] + c.rw_sigma * z[j];
  if (c.n_pose > 0 && tid == 0) {
    const Pose P = mh_step_pose(c, restate ? -1 : pose_leaf, z[0]);  // (… and so does the pose)
    w.inst->pose = P;
    for (int k = 0; k < 2; ++k) { w.search[k]->corr[0].pose = P; w.search[k]->corr[1].pose = P; }
  }
}

// tail of a step: MetropolisHastings.next with the device results.  One wave per chain: lane j holds coefficient j of the proposed and
// of the current state; the O(r) sums run in the host's order (ascending j, one rounding per operation) over values broadcast from the
// lanes — every lane computes the same numbers, the decision is uniform — and the state, the record and the decomposition records
// are written by all lanes together (a single lane walking through global memory took 30-45 µs per step).
__device__ __forceinline__ double mh_bcast(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// The kernel is one wave walking through memory it alone uses, so its time is the number of DEPENDENT trips there: everything the step's
// outcome does not decide — the results, the state, the next step's mixture draw and normals, and the OTHER set's launch arguments and
// decomposition records, which an accepted step switches to — is requested up front, behind the chain record's own words (two trips);
// the outcome then only selects what is stored.  The arguments of the next step are written once, with the step's own entries
// (propose / prop / zin) already in them; a rejected step, whose live arguments are those of its set already, writes only those.
// kWide: a chain of the wide step (MhWide) — NS coefficients per lane (ranks up to 64·NS), the full-mesh Hausdorff evaluator as well, the
// next step's head left to k_mhw_front; its records do not change with the outcome (k_mhw_adopt moves the accepted posterior instead).
template <int NS, bool kWide>
__device__ __forceinline__ void mh_decide_body(MhChain& c) {
  typedef unsigned long long u64;
  const int lane = threadIdx.x;
  const int r = c.r, P = 10 + r, n_icp = c.n_icp, cur_sel = c.cur_sel, other = cur_sel ^ 1;
  const long long step = c.step;
  const u64 seed = c.seed;
  const double cur_p = c.cur_p, rw_sigma = c.rw_sigma;
  const double ninf = -__builtin_inf();
  // (uniform) the next step's normals are on the device — and its head is not k_mh_front's business (mixtures with pose walks)
  const bool has_next = !kWide && !c.front_every_step && step + 1 - c.normals_first < (long long)c.normals_rows;
  const int pose_move = c.pose_move, leaf_now = c.leaf;
  // the NEXT step's mixture draw (MixtureProposal.propose, as mh_front_body): an integer hash of the step number
  int gen_n = -1, leaf_n = 2;
  {
    const int o = mh_pick(c.n_outer, c.outer_w, mh_uniform(seed, (u64)(step + 1), 0));
    if (c.outer_kind[o] == 1) { gen_n = mh_pick(n_icp, c.icp_w, mh_uniform(seed, (u64)(step + 1), 1)); leaf_n = gen_n; }
  }
  // ---- the step's results and the chain's state
  double cpj[NS], thj[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    cpj[s] = lane + 64 * s < r ? c.coeff_prop[lane + 64 * s] : 0.0;   // proposed coefficients (launch 1's copy in the state slot)
    thj[s] = lane + 64 * s < r ? c.theta[10 + lane + 64 * s] : 0.0;   // current ones
  }
  const double thp = lane < 10 ? c.theta[lane] : 0.0;       // the current pose …
  const double thq = lane < 10 ? (c.n_pose > 0 ? c.prop_pose[lane] : thp) : 0.0;  // … and the proposed state's (a pose walk moves one of its parameters)
  int st_chol = 0;
  if (lane < n_icp) {
    if constexpr (kWide) st_chol = c.wide->chol[lane][0]; else st_chol = c.chol_status[lane];
  }
  const int st_tail = lane < 2 * n_icp ? c.tail_status[lane] : 0;
  const double resv = lane < 8 ? c.red[lane] : 0.0;
  const double tlv = lane < 2 * n_icp ? c.tails[lane] : 0.0;
  const int seqv = lane < n_icp ? c.eig_seq[lane] : 0;
  // the status word of the decomposition the CURRENT state's posteriors went through (written by the launch behind an earlier step's
  // decision, ahead of this kernel in stream order; 2 = the iteration did not converge): proposals drawn from such a basis are not
  // the reference's — the chain stops, as the host-stepped path does (chain_step_record -> check_status)
  const int st_eig = lane < n_icp ? c.eig_alt[cur_sel][lane].status[0] : 0;
  const double zn = (has_next && lane < r) ? c.normals[(size_t)(step + 1 - c.normals_first) * r + lane] : 0.0;
  // ---- the other set's records, by 8-byte words
  constexpr int kWB = sizeof(StepBeginArgs) / 8, kWS = sizeof(StepSearchArgs) / 8, kWR = sizeof(StepRegressionArgs) / 8, kWF = sizeof(StepFinishArgs) / 8;
  constexpr int kPB = (kWB + 63) / 64, kPS = (kWS + 63) / 64, kPR = (kWR + 63) / 64, kPF = (kWF + 63) / 64;
  constexpr int kWE = sizeof(EigenProblem) / 8, kWP = sizeof(ProposeIn) / 8;
  static_assert(sizeof(StepBeginArgs) % 8 == 0 && sizeof(StepSearchArgs) % 8 == 0 && sizeof(StepRegressionArgs) % 8 == 0 && sizeof(StepFinishArgs) % 8 == 0 &&
                sizeof(EigenProblem) % 8 == 0 && sizeof(ProposeIn) % 8 == 0, "records are moved in 8-byte words");
  static_assert(2 * kWE <= 64 && 4 * kWP <= 64, "one lane per word of the decomposition records / proposal inputs");
  constexpr int kOffPropose = offsetof(StepBeginArgs, propose) / 8, kOffProp = offsetof(StepBeginArgs, prop) / 8, kOffZin = offsetof(StepBeginArgs, zin) / 8;
  static_assert(offsetof(StepBeginArgs, propose) % 8 == 0 && offsetof(StepBeginArgs, prop) == offsetof(StepBeginArgs, propose) + 8 &&
                offsetof(StepBeginArgs, zin) % 8 == 0 && offsetof(StepBeginArgs, prop) % 8 == 0, "layout of the per-step entries of StepBeginArgs");
  constexpr int kOffVwarm = offsetof(EigenProblem, Vwarm) / 8, kOffLaunch = offsetof(EigenProblem, launch_id) / 8, kOffDone = offsetof(EigenProblem, done_value) / 8;
  static_assert(offsetof(EigenProblem, launch_id) % 8 == 0 && offsetof(EigenProblem, done_value) % 8 == 0, "the two ints lead their words");
  u64 wb[kPB], ws[kPS], wr[kPR], wf[kPF];
  if (has_next) {
    const u64* sb = (const u64*)c.begin_alt[other];
    const u64* ss = (const u64*)c.search_alt[other];
    const u64* sr = (const u64*)c.regression_alt[other];
    const u64* sf = (const u64*)c.finish_alt[other];
#pragma unroll
    for (int p = 0; p < kPB; ++p) wb[p] = lane + 64 * p < kWB ? sb[lane + 64 * p] : 0ull;
#pragma unroll
    for (int p = 0; p < kPS; ++p) ws[p] = lane + 64 * p < kWS ? ss[lane + 64 * p] : 0ull;
#pragma unroll
    for (int p = 0; p < kPR; ++p) wr[p] = lane + 64 * p < kWR ? sr[lane + 64 * p] : 0ull;
#pragma unroll
    for (int p = 0; p < kPF; ++p) wf[p] = lane + 64 * p < kWF ? sf[lane + 64 * p] : 0ull;
  } else {
#pragma unroll
    for (int p = 0; p < kPB; ++p) wb[p] = 0ull;
#pragma unroll
    for (int p = 0; p < kPS; ++p) ws[p] = 0ull;
#pragma unroll
    for (int p = 0; p < kPR; ++p) wr[p] = 0ull;
#pragma unroll
    for (int p = 0; p < kPF; ++p) wf[p] = 0ull;
  }
  const int ei = lane / kWE, ew = lane - ei * kWE;  // lane -> (decomposition record, word)
  const u64 eigw = ei < n_icp ? ((const u64*)&c.eig_alt[other][ei])[ew] : 0ull;
  // proposal inputs of the next step: blocks 0, 1 = prop_alt[set][its proposal]; 2, 3 = the entry the set's own arguments carry (a
  // step that proposes by the random walk leaves that one in place)
  const int pb = lane / kWP, pw = lane - pb * kWP;
  u64 propw = 0ull;
  if (has_next && pb < 4) propw = pb < 2 ? ((const u64*)&c.prop_alt[pb][gen_n < 0 ? 0 : gen_n])[pw] : ((const u64*)&c.begin_alt[pb - 2]->prop)[pw];

  // (a pose move's transition tails are discarded — fw_i = bw_i = −∞, NonRigidIcpProposal.scala:69-71: the reference and the
  // host-stepped wide step never look at them, so a tail that fell back on such a step must not stop the chain)
  const unsigned long long bad_chol = __ballot(st_chol != 0), bad_tail = pose_move ? 0ull : __ballot(st_tail != 0);
  int err = bad_chol ? 3 : (bad_tail ? 2 : 0);
  if (__ballot(st_eig != 0) && !err) err = 6;
  // ---- evaluators: ModelPriorEvaluator (:24-31), the likelihood from launch 4's reductions (finish_eval), ProductEvaluator
  double nn = 0.0, dd = 0.0, dd_b = 0.0;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    for (int j = 0; j < 64 && 64 * s + j < r; ++j) {
      const double cj = mh_bcast(cpj[s], j), tj = mh_bcast(thj[s], j);
      nn += cj * cj;
      const double d = cj - tj; dd += d * d;        // RandomShapeUpdateProposal.scala:37-45, to − from
      const double e = tj - cj; dd_b += e * e;      // … and the other way
    }
  const double prior = -0.5 * nn - c.prior_c;
  double res[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) res[k] = mh_bcast(resv, k);
  double lik;
  if (c.eval_kind == 0) {  // IndependentPointDistanceEvaluator.scala:60-64
    const double m2t = res[0], t2m = res[4];
    lik = c.eval_mode == 0 ? m2t : c.eval_mode == 1 ? t2m : 0.5 * m2t + 0.5 * t2m;
  } else if (kWide && c.eval_kind == 1) {  // HausdorffDistanceEvaluator.scala:33-34 (finish_eval)
    const double hd = res[1] > res[5] ? res[1] : res[5];
    lik = -c.exp_rate * hd + c.exp_lograte;
  } else {                 // CollectiveAverageHausdorffDistanceBoundaryAwareEvaluator.scala:66-78 (on an open target: the reductions drop the flagged ids)
    const double a0 = res[0] / res[2], h0 = res[1], a1 = res[4] / res[6], h1 = res[5];
    double a, h;
    bool empty;
    if (c.eval_mode == 0) { a = a0; h = h0; empty = res[2] == 0.0; }
    else if (c.eval_mode == 1) { a = a1; h = h1; empty = res[6] == 0.0; }
    else { a = 0.5 * a0 + 0.5 * a1; h = h0 > h1 ? h0 : h1; empty = res[2] == 0.0 || res[6] == 0.0; }
    if (empty && !err) err = 5;
    const double d = (a - c.gauss_mean) / c.gauss_sigma;
    lik = (-d * d / 2.0 - c.gauss_logn) + (-c.exp_rate * h + c.exp_lograte);
  }
  if (!(lik == lik) && !err) err = 4;
  double prop_p = 0.0;
  prop_p += prior;
  prop_p += lik;
  // ---- transition ratio: every leaf's density both ways, log-sum-exp through the mixture tree
  double fw_i[2] = {ninf, ninf}, bw_i[2] = {ninf, ninf};
  for (int i = 0; i < n_icp; ++i) {
    // (across a pose change the ICP proposals' densities are −∞: NonRigidIcpProposal.scala:72-74; what the tails computed is not looked at)
    fw_i[i] = pose_move ? ninf : mh_bcast(tlv, 2 * i); bw_i[i] = pose_move ? ninf : mh_bcast(tlv, 2 * i + 1);
    if ((!(fw_i[i] == fw_i[i]) || !(bw_i[i] == bw_i[i])) && !err) err = 4;
  }
  // RandomShapeUpdateProposal.scala:38-40: −∞ unless only the shape differs
  const double rw_t = pose_move ? ninf : -0.5 * dd / (rw_sigma * rw_sigma) - c.rw_logc;
  const double rw_tb = pose_move ? ninf : -0.5 * dd_b / (rw_sigma * rw_sigma) - c.rw_logc;
  // the six pose walks (PoseProposals.scala:46-60, :77-88): −∞ when anything OUTSIDE the walk's own group (rotation triple / translation)
  // differs — the shape, for an ICP or shape-walk proposal; the other group, for a pose walk — else the Gaussian density of the
  // walk's own parameter's residual (zero for the other walks of the moved group)
  double pose_f[6] = {ninf, ninf, ninf, ninf, ninf, ninf}, pose_b[6] = {ninf, ninf, ninf, ninf, ninf, ninf};
  if (pose_move) {
    const int moved = c.pose_index[leaf_now - 3], grp = moved >= 4 ? 4 : 1;
    const double from = mh_bcast(thp, moved), to = mh_bcast(thq, moved);
    for (int a = 0; a < c.n_pose; ++a) {
      const int idx = c.pose_index[a];
      if ((idx >= 4 ? 4 : 1) != grp) continue;
      const double df = (idx == moved ? to - from : 0.0) / c.pose_sigma[a], db = (idx == moved ? from - to : 0.0) / c.pose_sigma[a];
      pose_f[a] = -df * df / 2.0 - c.pose_logc[a];
      pose_b[a] = -db * db / 2.0 - c.pose_logc[a];
    }
  }
  double of[3] = {ninf, ninf, ninf}, ob[3] = {ninf, ninf, ninf};
  for (int o = 0; o < c.n_outer; ++o) {
    if (c.outer_kind[o] == 1) { of[o] = mh_lse(n_icp, c.icp_w, fw_i); ob[o] = mh_lse(n_icp, c.icp_w, bw_i); }
    else if (c.outer_kind[o] == 0) { of[o] = mh_lse(c.n_pose, c.pose_w, pose_f); ob[o] = mh_lse(c.n_pose, c.pose_w, pose_b); }
    else {  // mixedRandomShapeProposal: a one-component mixture (weight 0.5 / 0.5 = 1)
      const double one = 1.0;
      of[o] = mh_lse(1, &one, &rw_t); ob[o] = mh_lse(1, &one, &rw_tb);
    }
  }
  const double fw = mh_lse(c.n_outer, c.outer_w, of), bw = mh_lse(c.n_outer, c.outer_w, ob);
  if ((!(fw == fw) || !(bw == bw)) && !err) err = 4;
  if (err) {  // (uniform) the chain needs the host: it stands still from here on
    if (lane == 0) c.error = err;
    return;
  }
  const double t = (fw == ninf && bw == ninf) ? 0.0 : fw - bw;
  const double a = prop_p - cur_p - t;
  const bool acc = a > 0.0 || mh_uniform(seed, (u64)step, 2) < exp(a);
  // ---- state, record (host/icp_host.h: [index, status, leaf, log value of the state after the step, theta])
  const double new_p = acc ? prop_p : cur_p;
  const int new_sel = acc ? other : cur_sel;
#pragma unroll
  for (int s = 0; s < NS; ++s)
    if (acc && lane + 64 * s < r) c.theta[10 + lane + 64 * s] = cpj[s];
  if (acc && pose_move && lane < 10) c.theta[lane] = thq;
  if (c.records) {
    double* rec = c.records + (size_t)(step - c.rec_first) * (4 + P);
    if (lane == 0) { rec[0] = (double)step; rec[1] = acc ? 1.0 : 0.0; rec[2] = (double)leaf_now; rec[3] = new_p; }
    if (lane < 10) rec[4 + lane] = acc ? thq : thp;
#pragma unroll
    for (int s = 0; s < NS; ++s)
      if (lane + 64 * s < r) rec[14 + lane + 64 * s] = acc ? cpj[s] : thj[s];
  }
  // ---- the KL bases of an accepted state's posteriors (both directions), as icp_chain_step_batched starts them
  {
    const int q = __shfl(seqv, ei < n_icp ? ei : 0, 64) + 1;  // this lane's record: its decomposition number
    if (acc && ei < n_icp) {
      u64 v = eigw;
      if (ew == kOffVwarm && ((q + 1) & 127) == 0) v = 0ull;  // every 128th cold (icp_proposal::prepare_eigen)
      if (ew == kOffLaunch) v = (v & 0xffffffff00000000ull) | (u64)(unsigned)(1 + (int)((unsigned)q % (unsigned)c.pw_id_mask));
      if (ew == kOffDone) v = (v & 0xffffffff00000000ull) | (u64)(unsigned)q;
      ((u64*)&c.eig_live[ei])[ew] = v;
    }
    if (lane < n_icp) {
      c.eig_skip[lane] = acc ? 0 : 1;
      if (acc) c.eig_seq[lane] = seqv + 1;
    }
  }
  if (lane == 0) {
    if (acc) { c.cur_p = prop_p; c.cur_sel = new_sel; ++c.accepted; }
    c.step = step + 1;
  }
  // ---- the head of the NEXT step, while its normals are on the device (one launch and its boundary less per step): mh_front_body with
  // everything in registers already
  if (has_next) {
    if (lane == 0) { c.gen = gen_n; c.leaf = leaf_n; }
    // ICP: posterior.sample()'s standard normals (NonRigidIcpProposal.scala:55); shape walk: the sample itself, c + σ·z
    // (RandomShapeUpdateProposal.scala:31-35)
    const double zval = gen_n >= 0 ? zn : (acc ? cpj[0] : thj[0]) + rw_sigma * zn;
    const int pbase = (gen_n >= 0 ? new_sel : 2 + new_sel) * kWP;  // where this step's proposal inputs sit among the lanes
    u64* db = (u64*)c.begin_live;
#pragma unroll
    for (int p = 0; p < kPB; ++p) {
      const int w = lane + 64 * p;
      const bool is_prop = w >= kOffProp && w < kOffProp + kWP, is_z = w >= kOffZin && w < kOffZin + r, is_flag = w == kOffPropose;
      const u64 pv = __shfl(propw, is_prop ? pbase + (w - kOffProp) : 0, 64);
      const double zv = __shfl(zval, is_z ? w - kOffZin : 0, 64);
      u64 v = wb[p];
      if (is_prop) v = pv;
      if (is_z) v = (u64)__double_as_longlong(zv);
      if (is_flag) v = (u64)(unsigned)(gen_n >= 0 ? 1 : 0);
      if (w < kWB && (acc || is_prop || is_z || is_flag)) db[w] = v;
    }
    if (acc) {  // (uniform) the other set's searches, regression and finish
      u64* ds = (u64*)c.search_live;
      u64* dr = (u64*)c.regression_live;
      u64* df = (u64*)c.finish_live;
#pragma unroll
      for (int p = 0; p < kPS; ++p) if (lane + 64 * p < kWS) ds[lane + 64 * p] = ws[p];
#pragma unroll
      for (int p = 0; p < kPR; ++p) if (lane + 64 * p < kWR) dr[lane + 64 * p] = wr[p];
#pragma unroll
      for (int p = 0; p < kPF; ++p) if (lane + 64 * p < kWF) df[lane + 64 * p] = wf[p];
    }
  }
}

__global__ void __launch_bounds__(64) k_mh_decide(MhChain* __restrict__ chains) {
  MhChain& c = chains[blockIdx.x];
  if (c.error) return;
  mh_decide_body<1, false>(c);
}
template <int NS>
__global__ void __launch_bounds__(64) k_mhw_decide(MhChain* __restrict__ chains) {
  MhChain& c = chains[blockIdx.x];
  if (c.error) return;
  mh_decide_body<NS, true>(c);
}
// an accepted state's posterior becomes the current state's: M, alpha and the coefficients of the proposed state's entry are copied
// into the current state's (record q of the launch; skip[q] != 0: the chain did not move).  r² + 2r doubles — 0.3 MB at rank 200.
__global__ void __launch_bounds__(256) k_mhw_adopt(int r, const MhAdopt* __restrict__ records, const int* __restrict__ skip) {
  const int q = blockIdx.y;
  if (skip && skip[q] != 0) return;
  const MhAdopt a = records[q];
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e < r * r) a.M_to[e] = a.M_from[e];
  if (e < r) { a.alpha_to[e] = a.alpha_from[e]; a.c_to[e] = a.c_from[e]; }
  if (a.V_from) {
    if (e < r * r) { a.V_to[e] = a.V_from[e]; a.Vt_to[e] = a.Vt_from[e]; }
    if (e < r) a.S_to[e] = a.S_from[e];
    if (e == 0) a.st_to[0] = a.st_from[0];
  }
  // (grid-stride: the grid is sized by r², a posterior's correspondence records by K — 3K + 1 words may be more than r² threads)
  const int stride = gridDim.x * 256;
#pragma unroll
  for (int k = 0; k < 6; ++k) {
    const int nb = a.corr_bytes[k], nw = nb >> 3;
    for (int w = e; w <= nw; w += stride) {
      if (w < nw) ((unsigned long long*)a.corr_to[k])[w] = ((const unsigned long long*)a.corr_from[k])[w];
      else
        for (int t = 8 * nw; t < nb; ++t) a.corr_to[k][t] = a.corr_from[k][t];
    }
  }
}

// a block of standard normals has arrived: chain b's rows start at base + b·stride, row 0 = the run's step `offset`
__global__ void k_mh_set_normals(MhChain* __restrict__ chains, int B, const double* base, int stride, int offset, int rows) {
  const int b = blockIdx.x * 64 + threadIdx.x;
  if (b >= B) return;
  chains[b].normals = base + (size_t)b * stride;
  chains[b].normals_first = chains[b].rec_first + offset;
  chains[b].normals_rows = rows;
}
void launch_mh_set_normals(hipStream_t st, int B, MhChain* chains, const double* base, int stride, int offset, int rows) {
  if (B > 0) hipLaunchKernelGGL(k_mh_set_normals, dim3((B + 63) / 64), dim3(64), 0, st, chains, B, base, stride, offset, rows);
}
void launch_mh_front(hipStream_t st, int B, MhChain* chains) {
  if (B > 0) hipLaunchKernelGGL(k_mh_front, dim3(B), dim3(64), 0, st, chains);
}
void launch_mh_decide(hipStream_t st, int B, MhChain* chains) {
  if (B > 0) hipLaunchKernelGGL(k_mh_decide, dim3(B), dim3(64), 0, st, chains);
}
void launch_mhw_front(hipStream_t st, int B, MhChain* chains, int restate) {
  if (B > 0) hipLaunchKernelGGL(k_mhw_front, dim3(B), dim3(64), 0, st, chains, restate);
}
void launch_mhw_decide(hipStream_t st, int B, int r, MhChain* chains) {
  if (B <= 0) return;
  if (r <= 64) hipLaunchKernelGGL(k_mhw_decide<1>, dim3(B), dim3(64), 0, st, chains);
  else if (r <= 128) hipLaunchKernelGGL(k_mhw_decide<2>, dim3(B), dim3(64), 0, st, chains);
  else hipLaunchKernelGGL(k_mhw_decide<4>, dim3(B), dim3(64), 0, st, chains);
}
void launch_mhw_adopt(hipStream_t st, int r, int n, const MhAdopt* records, const int* skip) {
  if (n > 0) hipLaunchKernelGGL(k_mhw_adopt, dim3((r * r + 255) / 256, n), dim3(256), 0, st, r, records, skip);
}

size_t step_batch_bytes(int B) {
  return up16(sizeof(StepBeginArgs) * B) + up16(sizeof(StepSearchArgs) * B) + up16(sizeof(StepRegressionArgs) * B) +
         up16(sizeof(StepFinishArgs) * B);
}

void launch_step_batch(hipStream_t st, int B, const StepCapture* caps, void* pinned, void* device, hipStream_t st_finish, hipEvent_t ev,
                       StepBatchGate gate) {
  if (B <= 0) return;
  const size_t o1 = up16(sizeof(StepBeginArgs) * B), o2 = o1 + up16(sizeof(StepSearchArgs) * B),
               o3 = o2 + up16(sizeof(StepRegressionArgs) * B), total = step_batch_bytes(B);
  char* h = (char*)pinned;
  char* d = (char*)device;
  int gx[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < B; ++b) {
    ((StepBeginArgs*)h)[b] = caps[b].begin;
    ((StepSearchArgs*)(h + o1))[b] = caps[b].search;
    ((StepRegressionArgs*)(h + o2))[b] = caps[b].regression;
    ((StepFinishArgs*)(h + o3))[b] = caps[b].finish;
    for (int k = 0; k < 5; ++k) gx[k] = caps[b].grid[k] > gx[k] ? caps[b].grid[k] : gx[k];
  }
  const int n16 = (int)(total / 16);
  hipLaunchKernelGGL(k_step_batch_args, dim3(cdiv(n16, 256)), dim3(256), 0, st, (const uint4*)h, (uint4*)d, n16, gate);
  if (gx[0] > 0) {
    ProfScope _ps(st, KID_STEP_BEGIN);
    static const bool no_reg = dev_env("ICP_BEGIN_STREAMED") != nullptr;  // (A/B switch)
    const int r = caps[0].begin.r;  // (one rank per batch: checked by the caller)
    bool hold = false;  // any chain of the batch still waiting for its decomposition
    for (int b = 0; b < B; ++b) hold = hold || caps[b].begin.hold_regs != 0;
    if (hold && r >= 32 && r <= 52 && !no_reg) hipLaunchKernelGGL(k_step_begin_batch_reg<52>, dim3(gx[0], B), dim3(kStepBlock), 0, st, (const StepBeginArgs*)d);
    else if (hold && r >= 32 && r <= 64 && !no_reg) hipLaunchKernelGGL(k_step_begin_batch_reg<64>, dim3(gx[0], B), dim3(kStepBlock), 0, st, (const StepBeginArgs*)d);
    else hipLaunchKernelGGL(k_step_begin_batch, dim3(gx[0], B), dim3(kStepBlock), 0, st, (const StepBeginArgs*)d);
  }
  if (gx[1] > 0) {
    ProfScope _ps(st, KID_STEP_FILTER);
    bool prepared = true;
    for (int b = 0; b < B; ++b) prepared = prepared && step_filter_prepared(caps[b].search);
    if (prepared) hipLaunchKernelGGL(k_step_filter_batch<true>, dim3(gx[1], B), dim3(kSearchBlock), 0, st, (const StepSearchArgs*)(d + o1));
    else hipLaunchKernelGGL(k_step_filter_batch<false>, dim3(gx[1], B), dim3(kSearchBlock), 0, st, (const StepSearchArgs*)(d + o1));
  }
  if (gx[2] > 0) { ProfScope _ps(st, KID_STEP_RESOLVE); hipLaunchKernelGGL(k_step_resolve_batch, dim3(gx[2], B), dim3(64), 0, st, (const StepSearchArgs*)(d + o1)); }
  if (gx[3] > 0) {
    ProfScope _ps(st, KID_STEP_REGRESSION);
    // regression_fold is decided per POSTERIOR (it depends on K: a proposal of at most eight correspondences beside a larger one keeps
    // fold = 1), the kernel per LAUNCH: the folded kernel takes records of either kind (regression_tile_fold with fold = 1 is the plain
    // tile), the plain one would read a folded record's units as (tile, split) pairs — so any folded record selects the folded kernel
    bool reg_folded = false;
    for (int b = 0; b < B; ++b)
      for (int i = 0; i < caps[b].regression.n && i < 2; ++i) reg_folded = reg_folded || caps[b].regression.fold[i] > 1;
    if (reg_folded) hipLaunchKernelGGL(k_step_regression_batch_fold, dim3(gx[3], B), dim3(kStepBlock), 0, st, (const StepRegressionArgs*)(d + o2));
    else hipLaunchKernelGGL(k_step_regression_batch, dim3(gx[3], B), dim3(kStepBlock), 0, st, (const StepRegressionArgs*)(d + o2));
  }
  if (gx[4] > 0) {
    const FinishPlan p = finish_plan(caps[0].finish.r);  // (one rank per batch: checked by the caller)
    // launch 5 keeps four CUs per chain busy for 30 µs whatever the batch: on a stream of its own it runs beside the first
    // launches of the NEXT batch on `st` (other chains: nothing of theirs depends on it)
    if (st_finish && st_finish != st && ev) {
      (void)hipEventRecord(ev, st);
      (void)hipStreamWaitEvent(st_finish, ev, 0);
      st = st_finish;
    }
    ProfScope _ps(st, KID_STEP_FINISH);
    const StepFinishArgs* fb = (const StepFinishArgs*)(d + o3);
    if (p.E == 1 && p.NT == 256) launch_finish_batch<1, 256>(st, fb, gx[4], B, p.shmem);
    else if (p.E == 2 && p.NT == 512) launch_finish_batch<2, 512>(st, fb, gx[4], B, p.shmem);
    else launch_finish_batch<2, 1024>(st, fb, gx[4], B, p.shmem);
  }
}

}  // namespace icp
