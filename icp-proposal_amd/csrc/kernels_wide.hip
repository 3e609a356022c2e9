// kernels_wide.hip — the wide step (gfx950): one Metropolis–Hastings step of B chains for what the five merged launches of
// kernels_step.hip do not cover (see icp_kernels.hpp, "the wide step"): targets with a boundary, the full-mesh Hausdorff evaluator,
// ranks up to 200, pose moves.  The stages are the per-stage kernels' own device bodies (icp_search.hpp, icp_dense.hpp); what is
// new is that nothing between them goes through the host — inputs arrive as kernel arguments or per-chain records in device
// memory, results are written into pinned host memory by the last launch — and that B chains share every launch (chain =
// blockIdx.y, or one entry of a by-value argument array for the one-workgroup kernels).
#include <algorithm>
#include <cstddef>
#include <cstring>

#include "icp_kernels.hpp"
#include "icp_search.hpp"
#include "icp_dense.hpp"

namespace icp {

namespace {

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

// ---------------------------------------------------------------- W1: proposal (NonRigidIcpProposal.scala:53-62), or given coefficients
template <int NT>
__global__ void __launch_bounds__(NT) k_wide_propose(int r, WideProposeArgs a, int tpr_log2) {
  __shared__ double s_c[512];
  const WideProposeItem& it = a.it[blockIdx.x];
  if (it.kind == 1) {
    propose_body<true>(r, it.in, s_c, tpr_log2);  // (the arithmetic of k_propose<NT>: same block size, same matvec layout)
  } else {
    for (int j = threadIdx.x; j < r; j += NT) s_c[j] = it.src[j];
    __syncthreads();
  }
  for (int j = threadIdx.x; j < r; j += NT) {
    const double c = s_c[j];
    for (int o = 0; o < it.n_out; ++o) it.out[o][j] = c;
  }
}

// … from records in device memory (the on-device loop: k_mhw_front sets kind / in / src per step)
template <int NT>
__global__ void __launch_bounds__(NT) k_wide_propose_resident(int r, const WideProposeItem* __restrict__ items, int tpr_log2) {
  __shared__ double s_c[512];
  const WideProposeItem& it = items[blockIdx.x];
  if (it.kind == 1) {
    propose_body<true>(r, it.in, s_c, tpr_log2);
  } else {
    for (int j = threadIdx.x; j < r; j += NT) s_c[j] = it.src[j];
    __syncthreads();
  }
  for (int j = threadIdx.x; j < r; j += NT) {
    const double c = s_c[j];
    for (int o = 0; o < it.n_out; ++o) it.out[o][j] = c;
  }
}

// ---------------------------------------------------------------- W2: instances of up to G chains from ONE pass over the basis
// x_b = pose_b(x̄ + μ + Q c_b) for the chains b of a group: thread = model point, the 3·r basis values of the point are fetched once
// and multiplied with every chain's coefficients (LDS), each chain's sums in basis order with separately rounded multiply and add —
// the operations of instance_vertex_keep, so every chain's points are bit-identical to its own k_instance_keep launch.  A chain
// that only changes its pose (kind 1) takes its kept deformations instead (k_instance_pose).  137 MB of basis at N = 28,561 /
// rank 200: read once per group of chains instead of once per chain.
// One WAVE per workgroup and many loads in flight per lane: at N = 28,561 that is 447 waves for 1,024 SIMDs — every wave has a SIMD
// to itself and the kernel is the latency of its ⌈r/16⌉ batches of 48 loads (128-thread workgroups with 24 loads in flight left
// most CUs idle: 157 µs for a group of five chains where one chain's own launch took 23)
constexpr int kWideInstBlock = 64;
constexpr int kWideInstU = 16;  // basis columns (× 3 rows) in flight per batch of loads
constexpr int kWideMaxRankLds = 256;

template <int G>
__global__ void __launch_bounds__(kWideInstBlock) k_wide_instance(int B, int N, int r, const double* __restrict__ Qp, const double* __restrict__ ref,
                                                                  const double* __restrict__ mean, const WideInstArgs* __restrict__ batch, int block0, int nblocks) {
  __shared__ double s_c[G][kWideMaxRankLds];
  if ((int)blockIdx.x >= nblocks) return;  // (the row is padded to a multiple of eight)
  const int g0 = blockIdx.y * G;
  const int ng = min(G, B - g0);
  const int tid = threadIdx.x;
  bool any_full = false;
#pragma unroll
  for (int g = 0; g < G; ++g)
    if (g < ng && batch[g0 + g].kind == 0) {
      any_full = true;
      for (int j = tid; j < r; j += kWideInstBlock) s_c[g][j] = batch[g0 + g].coeffs[j];
    }
  __syncthreads();
  const int i = (block0 + (int)blockIdx.x) * kWideInstBlock + tid;  // (block0: a launch that takes the blocks from there on — the instance's head and rest)
  if (i >= N) return;
  double a0[G], a1[G], a2[G];
  const double m0 = mean[3 * i], m1 = mean[3 * i + 1], m2 = mean[3 * i + 2];
#pragma unroll
  for (int g = 0; g < G; ++g) { a0[g] = m0; a1[g] = m1; a2[g] = m2; }
  if (any_full) {  // (uniform)
    const double* q = Qp + i;
    int j = 0;
    for (; j + kWideInstU <= r; j += kWideInstU) {
      double v[3 * kWideInstU];
#pragma unroll
      for (int u = 0; u < 3 * kWideInstU; ++u) v[u] = q[(size_t)(3 * j + u) * N];
      // (all of them requested before the first multiply: left alone, the scheduler sinks each load to its use to save registers —
      // one load in flight per lane, and a group of five chains took 150 µs where one chain's launch takes 23)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int u = 0; u < kWideInstU; ++u)
#pragma unroll
        for (int g = 0; g < G; ++g) {
          const double c = s_c[g][j + u];
          a0[g] = a0[g] + v[3 * u] * c;
          a1[g] = a1[g] + v[3 * u + 1] * c;
          a2[g] = a2[g] + v[3 * u + 2] * c;
        }
    }
    for (; j < r; ++j) {
      const double v0 = q[(size_t)(3 * j) * N], v1 = q[(size_t)(3 * j + 1) * N], v2 = q[(size_t)(3 * j + 2) * N];
#pragma unroll
      for (int g = 0; g < G; ++g) {
        const double c = s_c[g][j];
        a0[g] = a0[g] + v0 * c;
        a1[g] = a1[g] + v1 * c;
        a2[g] = a2[g] + v2 * c;
      }
    }
  }
#pragma unroll
  for (int g = 0; g < G; ++g) {
    if (g >= ng) continue;
    const WideInstArgs& a = batch[g0 + g];
    double d0 = a0[g], d1 = a1[g], d2 = a2[g];
    if (a.kind == 1) { d0 = a.defo_src[3 * i]; d1 = a.defo_src[3 * i + 1]; d2 = a.defo_src[3 * i + 2]; }
    a.defo[3 * i] = d0; a.defo[3 * i + 1] = d1; a.defo[3 * i + 2] = d2;
    const d3 p = instance_pose(i, ref, a.pose, d0, d1, d2);  // ModelFittingParameters.scala:108-110
    a.x[3 * i] = p.x; a.x[3 * i + 1] = p.y; a.x[3 * i + 2] = p.z;
    if (a.has_surf) {
      if (i < a.surf.K) surface_init_with(a.surf, i, p, load_hint_triangle(a.surf, i));  // query i = model point i (NonRigidIcpProposal.scala:96)
      if (i < kQU && a.surf.K + i < a.surf.Kpad) surface_init_at(a.surf, a.surf.K + i, d3{0.0, 0.0, 0.0});  // sentinel slots
    }
    if (a.has_surf2) {
      const int k = i - (a.has_surf ? a.surf.K : 0);  // the ids behind the first task's
      if (k >= 0 && k < a.surf2.K) surface_init_with(a.surf2, k, p, load_hint_triangle(a.surf2, k));
      if (i < kQU && a.surf2.K + i < a.surf2.Kpad) surface_init_at(a.surf2, a.surf2.K + i, d3{0.0, 0.0, 0.0});
    }
  }
}

// ---------------------------------------------------------------- W3: what needs the COMPLETE new instance before the searches
// bounding spheres of its triangles (the evaluator's target -> model queries search it), the bounds of those queries (distance to
// each query's previous winner at its new position), the counters of the vertex searches, the accumulating reduction outputs
// mode 0: everything; 1: only what does NOT read the new instance — the counters and the accumulating outputs (the main sequence's
// nearest-vertex stage waits for these, not for the instance's 28,561 points); 2: only what does (spheres, the t2m queries' bounds)
__global__ void __launch_bounds__(kSearchBlock) k_wide_prepare(const WidePrepArgs* __restrict__ batch, int mode) {
  const WidePrepArgs& a = batch[blockIdx.y];
  int b = blockIdx.x;
  const int nb_s = (a.T + kSearchBlock - 1) / kSearchBlock;
  const int nb_q0 = a.has_t2m ? (a.t2m.Kpad + kSearchBlock - 1) / kSearchBlock : 0;
  if (mode == 1 && b < nb_s + nb_q0) return;
  if (mode == 2 && b >= nb_s + nb_q0) return;
  if (b < nb_s) {
    const int pos = b * kSearchBlock + threadIdx.x;
    if (pos < a.T) {
      const int t = a.order ? a.order[pos] : pos;
      a.spheres[pos] = tri_sphere(a.x, a.tris, t);
      sphere_triangles(a.spheres, a.T)[pos] = t;
    }
    return;
  }
  b -= nb_s;
  const int nb_q = a.has_t2m ? (a.t2m.Kpad + kSearchBlock - 1) / kSearchBlock : 0;
  if (b < nb_q) {
    surface_init(a.t2m, b * kSearchBlock + threadIdx.x);
    return;
  }
  b -= nb_q;
  for (int c = 0; c < a.n_cnt; ++c) {
    const int nb_c = (a.cnt_n[c] + kSearchBlock - 1) / kSearchBlock;
    if (b < nb_c) {
      const int k = b * kSearchBlock + threadIdx.x;
      if (k < a.cnt_n[c]) a.cnt[c][k] = 0;
      if (c == 0 && b == 0 && (int)threadIdx.x < a.n_zero_d) a.zero_d[threadIdx.x] = 0.0;
      return;
    }
    b -= nb_c;
  }
  if (a.n_cnt == 0 && b == 0 && (int)threadIdx.x < a.n_zero_d) a.zero_d[threadIdx.x] = 0.0;
}
int wide_prep_blocks(const WidePrepArgs& a) {
  int n = cdiv(a.T, kSearchBlock) + (a.has_t2m ? cdiv(a.t2m.Kpad, kSearchBlock) : 0);
  for (int c = 0; c < a.n_cnt; ++c) n += cdiv(a.cnt_n[c], kSearchBlock);
  if (a.n_cnt == 0) n += 1;
  return n;
}

// ---------------------------------------------------------------- W4-W7: the step's searches (the batched filter / resolve bodies)
template <bool kPrepared>
__global__ void __launch_bounds__(kSearchBlock, kPrepared ? 8 : 1) k_wide_filter(const StepSearchArgs* __restrict__ batch) {
  const StepSearchArgs& a = batch[blockIdx.y];
  const int b = blockIdx.x;
  const int nt = a.n_surf + a.n_vert;
  if (b >= a.fstart[nt]) return;
  int task = 0;
  while (task + 1 < nt && b >= a.fstart[task + 1]) ++task;
  const int l = b - a.fstart[task];
  const int ksplit = task < a.n_surf ? a.s[task].ksplit : a.v[task - a.n_surf].ksplit;
  const int bx = l / (8 * ksplit) * 8 + (l & 7), by = (l % (8 * ksplit)) >> 3;  // see filter_grid_blocks
  if (task < a.n_surf) {
    if (bx < a.s[task].tblocks) surface_filter<kPrepared>(a.s[task], bx, by);
  } else {
    if (bx < a.v[task - a.n_surf].vblocks) vertex_filter(a.v[task - a.n_surf], bx, by);
  }
}

// ModelSampling correspondence of model id k behind the nearest-vertex search of its surface point (NonRigidIcpProposal.scala:97-109)
__device__ __forceinline__ void correspond_model_nnv_wave(const CorrTask& c, int k, int nnv) {
  const d3 cp = ld3(c.cp + 3 * k);
  const bool on_boundary = nnv >= 0 && nnv != kNoIndex ? c.boundary[nnv] != 0 : false;  // :98-99
  const d3 n = vertex_normal_wave(c.x, c.tris, c.adj_off, c.adj, k);                      // :100
  if ((threadIdx.x & 63) == 0)
    write_corr(c.cb, k, k, nnv == kNoIndex ? -1 : nnv, cp, c.boundary_aware ? !on_boundary : true, n, c.pose, c.ref, c.mean);
}

__device__ __forceinline__ void wide_resolve_surface(const SurfaceTask& q, int k, int ci, const StepSearchArgs& a) {
  double best; int tri; d3 cp;
  surface_resolve(q, k, &best, &tri, &cp);
  if (ci >= 0 && k < a.corr[ci].K && tri != kNoIndex) correspond_model_wave(a.corr[ci], k, cp);
}
__device__ __forceinline__ void wide_resolve_vertex(const VertexTask& q, int k, int ci, const StepSearchArgs& a) {
  double best; int idx;
  vertex_resolve(q, k, &best, &idx);
  if (ci >= 0 && k < a.corr[ci].K) {
    if (a.corr[ci].cp) correspond_model_nnv_wave(a.corr[ci], k, idx);
    else if (idx != kNoIndex) correspond_target_wave(a.corr[ci], k, idx);
  }
}
__global__ void __launch_bounds__(64) k_wide_resolve(const StepSearchArgs* __restrict__ batch) {
  const StepSearchArgs& a = batch[blockIdx.y];
  const int b = blockIdx.x;
  if (b >= a.rstart[a.n_surf + a.n_vert]) return;
  if (a.n_surf > 0 && b < a.rstart[1]) wide_resolve_surface(a.s[0], b - a.rstart[0], a.s_corr[0], a);
  else if (a.n_surf > 1 && b < a.rstart[2]) wide_resolve_surface(a.s[1], b - a.rstart[1], a.s_corr[1], a);
  else if (a.n_vert > 0 && b < a.rstart[a.n_surf + 1]) wide_resolve_vertex(a.v[0], b - a.rstart[a.n_surf], a.v_corr[0], a);
  else if (a.n_vert > 1) wide_resolve_vertex(a.v[1], b - a.rstart[a.n_surf + 1], a.v_corr[1], a);
}

// ---------------------------------------------------------------- W8: regression partial sums + the likelihood's reductions
// Σ kept distances, their maximum and count with an optional boundary-flag test (k_dist_stats: Collective…Evaluator.scala:44-63)
__device__ __forceinline__ void dist_stats_flags_body(int K, const double* __restrict__ d2, const unsigned char* __restrict__ flags,
                                                      const int* __restrict__ idx, int n_flags, double* __restrict__ out) {
  __shared__ double s_red[16];
  double sum = 0.0, mx = -__builtin_inf(), cnt = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    bool drop = false;
    if (flags) {
      const int i = idx ? idx[k] : k;
      drop = (i >= 0 && i < n_flags) ? flags[i] != 0 : false;
    }
    if (!drop) {
      const double d = sqrt(d2[k]);
      sum += d;
      mx = fmax(mx, d);
      cnt += 1.0;
    }
  }
  sum = block_sum(sum, s_red);
  cnt = block_sum(cnt, s_red);
  mx = block_max(mx, s_red);
  if (threadIdx.x == 0) { out[0] = sum; out[1] = mx; out[2] = cnt; }
}
// exact maximum of the distances by 64-bit atomic maxima of their bit patterns (k_dist_max); out_max zeroed by W3
__device__ __forceinline__ void dist_max_body(int K, const double* __restrict__ d2, int block, double* __restrict__ out_max) {
  __shared__ double s_red[16];
  const int k = block * blockDim.x + threadIdx.x;
  double mx = k < K ? d2[k] : 0.0;
  mx = block_max(mx, s_red);
  if (threadIdx.x == 0) atomicMax((unsigned long long*)out_max, d2bits(sqrt(mx)));
}

constexpr int kWideRegBlock = 256;
// (a multiple of 8 workgroups, the workgroups of one XCD on a contiguous range of units: see step_regression_body, kernels_step.hip)
__host__ __device__ inline int wide_reg_units_blocks(const StepRegressionArgs& a) { return 8 * ((((a.ustart[a.n] + 3) >> 2) + 7) >> 3); }
__host__ __device__ inline int wide_red_blocks(const WideRegArgs& a, int dir /* 0: model -> target, 1: target -> model */) {
  const bool on = dir == 0 ? a.eval_m2t != 0 : a.eval_t2m != 0;
  if (!on) return 0;
  const int K = dir == 0 ? a.Km : a.Kt;
  return a.eval_kind == 1 ? (K + kWideRegBlock - 1) / kWideRegBlock : 1;
}
template <bool kFolded>
__device__ __forceinline__ void wide_regression_body(const WideRegArgs* __restrict__ batch) {
  int chain = blockIdx.y, b = blockIdx.x;
  if constexpr (kFolded) {
    // XCD-aware (round 6).  Workgroups go to the eight XCDs in turn by their linear index (x fastest), and each XCD has an L2 of its
    // own: with blockIdx.y = chain, the eight or so workgroups of a posterior sat on eight XCDs and each fetched the posterior's operand
    // rows (2.6 MB at the face model's size) from HBM again — 317 MB per launch of 25 chains for 65 MB of rows (profiles/r06_pmc_traffic.json
    // before this change).  Here the launch's (chain, block) pairs are dealt out in order, XCD by XCD: XCD j takes the pairs
    // [start_j, start_{j+1}) — three or four WHOLE chains, whose rows then come from its L2 after the first touch.
    const int G = gridDim.x, total = G * (int)gridDim.y, L = (int)blockIdx.y * G + (int)blockIdx.x;
    const int q = total >> 3, rem = total & 7, j = L & 7;
    const int wk = j * q + min(j, rem) + (L >> 3);  // (the L-th workgroup is the (L >> 3)-th of XCD j)
    chain = wk / G;
    b = wk - chain * G;
  }
  const WideRegArgs& w = batch[chain];
  const StepRegressionArgs& a = w.reg;
  const int nb = a.n > 0 ? wide_reg_units_blocks(a) : 0;
  if (b < nb) {
    const int n_units = a.ustart[a.n];
    const int lb = kFolded ? b : (b & 7) * (nb >> 3) + (b >> 3);
    const int u = lb * 4 + (threadIdx.x >> 6);
    if (u < n_units) {
      const int which = u < a.ustart[1] ? 0 : 1;
      const int l = u - (which ? a.ustart[1] : 0), tile = l % a.ntiles, split = l / a.ntiles;
      if (l == 0 && (threadIdx.x & 63) == 0) { a.status[which][1] = 0; a.status[which][2] = 0; }
      if constexpr (kFolded) {  // (a kernel of its own: see step_regression_body, kernels_step.hip)
        if (a.macro[which] > 1 && a.X[which]) regression_macro_fold_x<2>(l, a.fold[which], a.K[which], a.kchunk[which], a.r, a.X[which], a.xrs, a.cb[which], a.wt[which], a.kappa[which], a.Mpart[which]);
        else if (a.macro[which] > 1) regression_macro_fold<2>(l, a.fold[which], a.K[which], a.kchunk[which], a.r, a.Q, a.cb[which], a.wt[which], a.kappa[which], a.Mpart[which]);
        else regression_tile_fold(tile, split, a.K[which], a.kchunk[which], a.r, a.Q, a.cb[which], a.wt[which], a.kappa[which], a.Mpart[which], a.fold[which]);
      } else if (which == 0) regression_tile(tile, split, a.K[0], a.kchunk[0], a.r, a.Q, a.cb[0], a.wt[0], a.kappa[0], a.Mpart[0]);
      else regression_tile(tile, split, a.K[1], a.kchunk[1], a.r, a.Q, a.cb[1], a.wt[1], a.kappa[1], a.Mpart[1]);
    }
    return;
  }
  b -= nb;
  for (int dir = 0; dir < 2; ++dir) {
    const int nr = wide_red_blocks(w, dir);
    if (b < nr) {
      const int K = dir == 0 ? w.Km : w.Kt;
      const double* d2 = dir == 0 ? w.d2m : w.d2t;
      double* out = w.red_out + 4 * dir;
      if (w.eval_kind == 0) sum_gauss_logpdf_body(K, d2, w.mean, w.sigma, out);             // IndependentPointDistanceEvaluator.scala:40-54
      else if (w.eval_kind == 1) dist_max_body(K, d2, b, out + 1);                            // HausdorffDistanceEvaluator.scala:33
      else dist_stats_flags_body(K, d2, dir == 0 ? w.flags_m : w.flags_t, dir == 0 ? w.idx_m : w.idx_t, w.n_flags, out);  // Collective…:44-63
      return;
    }
    b -= nr;
  }
}
__global__ void __launch_bounds__(kWideRegBlock) k_wide_regression(const WideRegArgs* __restrict__ batch) { wide_regression_body<false>(batch); }
// … with the posteriors' split-K leaves folded into one partial each, in 2 x 2 macro tiles at ranks >= 80 (WideLaunchPlan::reg_folded)
__global__ void __launch_bounds__(kWideRegBlock) k_wide_regression_fold(const WideRegArgs* __restrict__ batch) { wide_regression_body<true>(batch); }

// the folded posteriors' operand rows (StepRegressionArgs::X), ahead of W8: blockIdx.y = chain, blockIdx.z = posterior of the chain
__global__ void __launch_bounds__(256) k_wide_xrows(const WideRegArgs* __restrict__ batch) {
  const StepRegressionArgs& a = batch[blockIdx.y].reg;
  const int which = blockIdx.z;
  if (which >= a.n || !a.X[which] || a.fold[which] <= 1) return;
  regression_xrows(blockIdx.x * 256 + threadIdx.x, a.K[which], a.r, a.xrs, a.Q, a.cb[which], a.X[which]);
}

// ---------------------------------------------------------------- W12: results and completion flags into pinned host memory
__global__ void __launch_bounds__(64) k_wide_done(WideDoneArgs a) {
  const WideDoneItem& it = a.it[blockIdx.x];
  const int l = threadIdx.x;
  if (l < 8 && it.red_src) it.red_dst[l] = it.red_src[l];
  if (l >= 8 && l < 12 && it.st_src[l - 8]) it.st_dst[l - 8][0] = it.st_src[l - 8][0];
  __threadfence_system();
  __syncthreads();
  if (l == 0) __hip_atomic_store(it.host_flag, it.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ void __launch_bounds__(256) k_wide_args(const uint4* __restrict__ src, uint4* __restrict__ dst, int n16) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n16) dst[i] = src[i];  // src: pinned host memory, read in place
}

}  // namespace

int wide_reg_blocks(const WideRegArgs& a) {
  return (a.reg.n > 0 ? wide_reg_units_blocks(a.reg) : 0) + wide_red_blocks(a, 0) + wide_red_blocks(a, 1);
}

void launch_wide_propose(hipStream_t st, int r, const WideProposeArgs& a) {
  if (a.n <= 0) return;
  ProfScope _ps(st, KID_PROPOSE);
  // (block size and matvec layout as launch_propose chooses them: the same arithmetic, rank by rank)
  bool root_big = false;
  for (int i = 0; i < a.n; ++i) root_big = root_big || (a.it[i].kind == 1 && a.it[i].in.root && r > 64);
  if (root_big) hipLaunchKernelGGL(k_wide_propose<1024>, dim3(a.n), dim3(1024), 0, st, r, a, matvec_tpr_log2(r, 1024));
  else if (r > 134) hipLaunchKernelGGL(k_wide_propose<1024>, dim3(a.n), dim3(1024), 0, st, r, a, 4);
  else hipLaunchKernelGGL(k_wide_propose<256>, dim3(a.n), dim3(256), 0, st, r, a, matvec_tpr_log2(r, 256));
}

void launch_wide_propose_resident(hipStream_t st, int r, int n, const WideProposeItem* items) {
  if (n <= 0) return;
  ProfScope _ps(st, KID_PROPOSE);
  // (launch_wide_propose's choice for the KL-basis sampler — and for the Cholesky-root sampler up to rank 64)
  if (r > 134) hipLaunchKernelGGL(k_wide_propose_resident<1024>, dim3(n), dim3(1024), 0, st, r, items, 4);
  else hipLaunchKernelGGL(k_wide_propose_resident<256>, dim3(n), dim3(256), 0, st, r, items, matvec_tpr_log2(r, 256));
}

size_t wide_batch_bytes(int B) {
  return up16(sizeof(WideInstArgs) * B) + up16(sizeof(WidePrepArgs) * B) + 4 * up16(sizeof(StepSearchArgs) * B) + 2 * up16(sizeof(WideRegArgs) * B);
}
namespace {
struct WideOffsets { size_t prep, s1, s2, reg, s1b, s2b, regb, total; };
WideOffsets wide_offsets(int B) {
  WideOffsets o;
  o.prep = up16(sizeof(WideInstArgs) * B);
  o.s1 = o.prep + up16(sizeof(WidePrepArgs) * B);
  o.s2 = o.s1 + up16(sizeof(StepSearchArgs) * B);
  o.reg = o.s2 + up16(sizeof(StepSearchArgs) * B);
  o.s1b = o.reg + up16(sizeof(WideRegArgs) * B);
  o.s2b = o.s1b + up16(sizeof(StepSearchArgs) * B);
  o.regb = o.s2b + up16(sizeof(StepSearchArgs) * B);
  o.total = o.regb + up16(sizeof(WideRegArgs) * B);
  return o;
}
}  // namespace

WideInstArgs* wide_inst_record(void* block, int B, int b) { (void)B; return (WideInstArgs*)block + b; }
StepSearchArgs* wide_search_record(void* block, int B, int stage, int b) {
  const WideOffsets o = wide_offsets(B);
  return (StepSearchArgs*)((char*)block + (stage == 0 ? o.s1 : o.s2)) + b;
}

void wide_pack_args(int B, const WideChainArgs* chains, void* dst) {
  const WideOffsets o = wide_offsets(B);
  char* h = (char*)dst;
  for (int b = 0; b < B; ++b) {
    ((WideInstArgs*)h)[b] = chains[b].inst;
    ((WidePrepArgs*)(h + o.prep))[b] = chains[b].prep;
    ((StepSearchArgs*)(h + o.s1))[b] = chains[b].s1;
    ((StepSearchArgs*)(h + o.s2))[b] = chains[b].s2;
    ((WideRegArgs*)(h + o.reg))[b] = chains[b].reg;
    ((StepSearchArgs*)(h + o.s1b))[b] = chains[b].s1b;
    ((StepSearchArgs*)(h + o.s2b))[b] = chains[b].s2b;
    ((WideRegArgs*)(h + o.regb))[b] = chains[b].regb;
  }
}

void launch_wide_head(hipStream_t st, const WideLaunchPlan& plan, const WideChainArgs* chains, void* pinned, void* device) {
  const int B = plan.B;
  if (B <= 0) return;
  wide_pack_args(B, chains, pinned);
  const int n16 = (int)(wide_offsets(B).total / 16);
  hipLaunchKernelGGL(k_wide_args, dim3(cdiv(n16, 256)), dim3(256), 0, st, (const uint4*)pinned, (uint4*)device, n16);
  launch_wide_head_resident(st, plan, device);
}

void launch_wide_head_resident(hipStream_t st, const WideLaunchPlan& plan, void* device, int part) {
  const int B = plan.B;
  if (B <= 0) return;
  const WideOffsets o = wide_offsets(B);
  char* d = (char*)device;
  const int gx_all = cdiv(plan.N, kWideInstBlock);
  const int head = (part != 0 && plan.inst_head_blocks > 0 && plan.inst_head_blocks < gx_all) ? plan.inst_head_blocks : 0;
  if (part == 1 && head == 0) part = 0;            // (no split in this plan: everything now …
  if (part == 2 && head == 0) return;              //  … and nothing later)
  const int block0 = part == 2 ? head : 0;
  const int gx = part == 1 ? head : gx_all - block0;
  {
    ProfScope _ps(st, KID_INSTANCE);
    const dim3 block(kWideInstBlock);
    const WideInstArgs* ia = (const WideInstArgs*)d;
    // (the head: a handful of point blocks the main sequence waits for — a wave per chain and block: a wave's time is its own chain of 13 batches
    // of 48 loads and G x 96 f64 operations each, 81 µs in groups of seven whatever else is resident)
    if (B == 1 || part == 1) hipLaunchKernelGGL(k_wide_instance<1>, dim3(gx, B), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
    else if (B == 2) hipLaunchKernelGGL(k_wide_instance<2>, dim3(gx, 1), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
    else if (B <= 4) hipLaunchKernelGGL(k_wide_instance<4>, dim3(gx, 1), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
    else if (B <= 6) hipLaunchKernelGGL(k_wide_instance<6>, dim3(gx, 1), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
    // (round 6, tried: groups of thirteen — two passes over the 137 MB basis instead of four for 25 chains — 151 µs against 128: beyond
    // eight chains a group the wave is bound by its own unfused f64 multiply-adds, 6 per chain and basis column, not by HBM)
    else {
      // more than eight chains: groups of equal size, at most eight chains each (25 chains: 7 + 7 + 7 + 4, not 8 + 8 + 8 + 1), the row of point
      // blocks padded to a multiple of eight so that the groups of ONE point block — linear workgroup ids a whole row apart — go to the
      // same XCD: all groups are resident at once, and the basis rows the first fetches are L2 hits for the others (round 6: 129 -> 99 µs for
      // 25 chains; HBM bytes per launch from four passes over the 137 MB basis towards one)
      const int ng = cdiv(B, 8), G = cdiv(B, ng), gx8 = (gx + 7) / 8 * 8;
      if (G <= 5) hipLaunchKernelGGL(k_wide_instance<5>, dim3(gx8, cdiv(B, 5)), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
      else if (G == 6) hipLaunchKernelGGL(k_wide_instance<6>, dim3(gx8, cdiv(B, 6)), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
      else if (G == 7) hipLaunchKernelGGL(k_wide_instance<7>, dim3(gx8, cdiv(B, 7)), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
      else hipLaunchKernelGGL(k_wide_instance<8>, dim3(gx8, cdiv(B, 8)), block, 0, st, B, plan.N, plan.r, plan.Qp, plan.ref, plan.mean, ia, block0, gx);
    }
  }
  if (plan.grid_prep > 0) {
    ProfScope _ps(st, KID_TRI_SPHERES);
    hipLaunchKernelGGL(k_wide_prepare, dim3(plan.grid_prep, B), dim3(kSearchBlock), 0, st, (const WidePrepArgs*)(d + o.prep), part);
  }
}

namespace {
void launch_wide_searches(hipStream_t st, int B, int gf, int gr, bool prepared, const StepSearchArgs* a, int kid_f, int kid_r) {
  if (gf > 0) {
    ProfScope _ps(st, kid_f);
    if (prepared) hipLaunchKernelGGL(k_wide_filter<true>, dim3(gf, B), dim3(kSearchBlock), 0, st, a);
    else hipLaunchKernelGGL(k_wide_filter<false>, dim3(gf, B), dim3(kSearchBlock), 0, st, a);
  }
  if (gr > 0) {
    ProfScope _ps(st, kid_r);
    hipLaunchKernelGGL(k_wide_resolve, dim3(gr, B), dim3(64), 0, st, a);
  }
}
}  // namespace

void launch_wide_main(hipStream_t st, const WideLaunchPlan& plan, void* device) {
  const int B = plan.B;
  if (B <= 0) return;
  const WideOffsets o = wide_offsets(B);
  char* d = (char*)device;
  launch_wide_searches(st, B, plan.grid_f1, plan.grid_r1, plan.f1_prepared, (const StepSearchArgs*)(d + o.s1), KID_STEP_FILTER, KID_STEP_RESOLVE);
  launch_wide_searches(st, B, plan.grid_f2, plan.grid_r2, true, (const StepSearchArgs*)(d + o.s2), KID_VERTEX_FILTER, KID_VERTEX_RESOLVE);
  if (plan.grid_reg > 0 && plan.reg_folded && plan.grid_xrows > 0)
    hipLaunchKernelGGL(k_wide_xrows, dim3(plan.grid_xrows, B, 2), dim3(256), 0, st, (const WideRegArgs*)(d + o.reg));
  if (plan.grid_reg > 0) {
    ProfScope _ps(st, KID_STEP_REGRESSION);
    if (plan.reg_folded) hipLaunchKernelGGL(k_wide_regression_fold, dim3(plan.grid_reg, B), dim3(kWideRegBlock), 0, st, (const WideRegArgs*)(d + o.reg));
    else hipLaunchKernelGGL(k_wide_regression, dim3(plan.grid_reg, B), dim3(kWideRegBlock), 0, st, (const WideRegArgs*)(d + o.reg));
  }
}

void launch_wide_eval(hipStream_t st, const WideLaunchPlan& plan, void* device) {
  const int B = plan.B;
  if (B <= 0) return;
  const WideOffsets o = wide_offsets(B);
  char* d = (char*)device;
  launch_wide_searches(st, B, plan.grid_f1b, plan.grid_r1b, plan.f1_prepared, (const StepSearchArgs*)(d + o.s1b), KID_SURFACE_FILTER, KID_SURFACE_RESOLVE);
  launch_wide_searches(st, B, plan.grid_f2b, plan.grid_r2b, true, (const StepSearchArgs*)(d + o.s2b), KID_VERTEX_FILTER, KID_VERTEX_RESOLVE);
  if (plan.grid_regb > 0) {
    ProfScope _ps(st, KID_REDUCE);
    hipLaunchKernelGGL(k_wide_regression, dim3(plan.grid_regb, B), dim3(kWideRegBlock), 0, st, (const WideRegArgs*)(d + o.regb));
  }
}

int wide_prep_grid(const WidePrepArgs& a) { return wide_prep_blocks(a); }

void launch_wide_done(hipStream_t st, const WideDoneArgs& a) {
  if (a.n > 0) hipLaunchKernelGGL(k_wide_done, dim3(a.n), dim3(64), 0, st, a);
}

}  // namespace icp
