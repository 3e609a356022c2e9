// icp_dense.hpp — device-side bodies of the correspondence / regression / r-space kernels (math: kernels_posterior.hip).
// Included by kernels_posterior.hip (one kernel per stage) and kernels_step.hip (merged per-step launches).
#pragma once
#include "icp_kernels.hpp"

namespace icp {

constexpr int kLdsDoubles = 18432;  // 144 KiB of the 160 KiB LDS for matrix-sized buffers

// ---------------------------------------------------------------- correspondences

__device__ __forceinline__ void write_corr(const CorrBuffers& cb, int k, int id, int aux, d3 pt, bool keep, d3 n,
                                           const Pose& pose, const double* __restrict__ ref, const double* __restrict__ mean) {
  // inverse RIGID pose (NonRigidIcpProposal.scala:142): Rᵀ((pt − t) − ctr) + ctr, then minus x̄_id (:108) and μ_id
  double v0 = (pt.x - pose.t[0]) - pose.ctr[0], v1 = (pt.y - pose.t[1]) - pose.ctr[1], v2 = (pt.z - pose.t[2]) - pose.ctr[2];
  double b0 = ((pose.R[0] * v0 + pose.R[3] * v1) + pose.R[6] * v2) + pose.ctr[0];
  double b1 = ((pose.R[1] * v0 + pose.R[4] * v1) + pose.R[7] * v2) + pose.ctr[1];
  double b2 = ((pose.R[2] * v0 + pose.R[5] * v1) + pose.R[8] * v2) + pose.ctr[2];
  cb.id[k] = id;
  cb.aux[k] = aux;
  cb.pt[3 * k] = pt.x; cb.pt[3 * k + 1] = pt.y; cb.pt[3 * k + 2] = pt.z;
  cb.keep[k] = keep ? 1 : 0;
  cb.nhat[3 * k] = n.x; cb.nhat[3 * k + 1] = n.y; cb.nhat[3 * k + 2] = n.z;
  cb.e[3 * k] = (b0 - ref[3 * id]) - mean[3 * id];
  cb.e[3 * k + 1] = (b1 - ref[3 * id + 1]) - mean[3 * id + 1];
  cb.e[3 * k + 2] = (b2 - ref[3 * id + 2]) - mean[3 * id + 2];
}


// NonRigidIcpProposal.scala:94-109 for model id k with surface point cp
__device__ __forceinline__ void correspond_model_one(const CorrTask& c, int k, d3 cp) {
  int aux = c.nnv ? c.nnv[k] : -1;
  bool on_boundary = (c.nnv && aux >= 0) ? c.boundary[aux] != 0 : false;  // :98-99
  d3 n = vertex_normal(c.x, c.tris, c.adj_off, c.adj, k);                 // :100
  write_corr(c.cb, k, k, aux, cp, c.boundary_aware ? !on_boundary : true, n, c.pose, c.ref, c.mean);
}

// NonRigidIcpProposal.scala:117-130 for decimated-target point k with nearest model vertex id
__device__ __forceinline__ void correspond_target_one(const CorrTask& c, int k, int id) {
  bool on_boundary = c.boundary[id] != 0;                 // :119
  d3 n = vertex_normal(c.x, c.tris, c.adj_off, c.adj, id); // :120
  write_corr(c.cb, k, id, -1, ld3(c.tpts + 3 * k), c.boundary_aware ? !on_boundary : true, n, c.pose, c.ref, c.mean);
}

// wave-cooperative versions (all 64 lanes call; the adjacent cell normals are computed one per lane, then summed by
// every lane in ascending triangle id exactly like vertex_normal) — used where one wave owns one correspondence
__device__ __forceinline__ d3 vertex_normal_wave(const double* __restrict__ x, const int* __restrict__ tris,
                                                 const int* __restrict__ adj_off, const int* __restrict__ adj, int v) {
  const int l = threadIdx.x & 63;
  const int k0 = adj_off[v], k1 = adj_off[v + 1];
  d3 n = {0.0, 0.0, 0.0};
  for (int base = k0; base < k1; base += 64) {
    const int cnt = min(64, k1 - base);
    d3 cn = {0.0, 0.0, 0.0};
    if (l < cnt) cn = cell_normal(x, tris, adj[base + l]);
    for (int j = 0; j < cnt; ++j) {
      n.x += __shfl(cn.x, j, 64); n.y += __shfl(cn.y, j, 64); n.z += __shfl(cn.z, j, 64);
    }
  }
  return normalized(n);
}
__device__ __forceinline__ void correspond_model_wave(const CorrTask& c, int k, d3 cp) {
  int aux = c.nnv ? c.nnv[k] : -1;
  bool on_boundary = (c.nnv && aux >= 0) ? c.boundary[aux] != 0 : false;
  d3 n = vertex_normal_wave(c.x, c.tris, c.adj_off, c.adj, k);
  if ((threadIdx.x & 63) == 0) write_corr(c.cb, k, k, aux, cp, c.boundary_aware ? !on_boundary : true, n, c.pose, c.ref, c.mean);
}
__device__ __forceinline__ void correspond_target_wave(const CorrTask& c, int k, int id) {
  bool on_boundary = c.boundary[id] != 0;
  d3 n = vertex_normal_wave(c.x, c.tris, c.adj_off, c.adj, id);
  if ((threadIdx.x & 63) == 0) write_corr(c.cb, k, id, -1, ld3(c.tpts + 3 * k), c.boundary_aware ? !on_boundary : true, n, c.pose, c.ref, c.mean);
}

// ---------------------------------------------------------------- K5a regression assembly on the f64 matrix cores
// Maug = Σ_i X_iᵀ W_i X_i with X_i = the 4×(r+1) block of correspondence i: rows 0-2 = [Q_i | e_i] (the three
// coordinate rows, weight w_t), row 3 = n̂_iᵀ[Q_i | e_i] (weight κ).  The contraction length per correspondence is
// exactly the K = 4 of v_mfma_f64_16x16x4_f64: one MFMA per correspondence per 16×16 output tile.
// Operand maps (cdna_hip_programming.md §3): lane l supplies A[i = l&15][k = l>>4] and B[k = l>>4][j = l&15];
// result register g of lane l is D[row = (l>>4) + 4g][col = l&15].
// grid = (tiles, splits); block = one wave; partial sums per split are reduced by the factor kernel (deterministic).
// The correspondence loop is unrolled ×4 so that the 24 gathered basis values of four correspondences are in flight
// together (the loop is L2-latency bound, not bandwidth bound).

typedef double d4_t __attribute__((ext_vector_type(4)));

struct RegOperands { double A, B; };

__device__ __forceinline__ void regression_load(int k, int r, const double* __restrict__ Q, const CorrBuffers& cb, int ca, int cbi,
                                                double ma, double mb, double ea, double eb, double* a, double* b) {
  const double* q = Q + (size_t)3 * cb.id[k] * r;
  const double e0 = cb.e[3 * k], e1 = cb.e[3 * k + 1], e2 = cb.e[3 * k + 2];
  a[0] = fma(ma, q[ca], ea * e0); a[1] = fma(ma, q[r + ca], ea * e1); a[2] = fma(ma, q[2 * r + ca], ea * e2);
  b[0] = fma(mb, q[cbi], eb * e0); b[1] = fma(mb, q[r + cbi], eb * e1); b[2] = fma(mb, q[2 * r + cbi], eb * e2);
}

__device__ __forceinline__ d4_t regression_mac(int k, const CorrBuffers& cb, const double* a, const double* b, int kk, double wt,
                                               double kappa, d4_t acc) {
  const double on = cb.keep[k] ? 1.0 : 0.0;  // boundary-filtered correspondences contribute weight 0
  const double n0 = cb.nhat[3 * k], n1 = cb.nhat[3 * k + 1], n2 = cb.nhat[3 * k + 2];
  const double va = fma(a[2], n2, fma(a[1], n1, a[0] * n0));
  const double vb = fma(b[2], n2, fma(b[1], n1, b[0] * n0));
  const double A_op = kk == 0 ? a[0] : kk == 1 ? a[1] : kk == 2 ? a[2] : va;
  const double B_op = (kk == 0 ? b[0] : kk == 1 ? b[1] : kk == 2 ? b[2] : vb) * (kk == 3 ? kappa : wt) * on;
  return __builtin_amdgcn_mfma_f64_16x16x4f64(A_op, B_op, acc, 0, 0, 0);
}

// tile = index into the LOWER triangle (tile row ti >= tile column tj, row-major: regression_tiles(r) of them) of the grid of
// 16×16 output tiles — the matrix is symmetric and every reader of the partial sums takes entries (i, k <= i) —
// split = which slice of the correspondence list; one wave
__device__ __forceinline__ void regression_tile(int tile, int split, int K, int kchunk, int r, const double* __restrict__ Q,
                                                const CorrBuffers& cb, double wt, double kappa, double* __restrict__ Mpart) {
  const int n = r + 1;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int l = threadIdx.x & 63, i16 = l & 15, kk = l >> 4;
  const int a = 16 * ti + i16, b = 16 * tj + i16;
  const int ca = a < r ? a : 0, cbi = b < r ? b : 0;
  const double ma = a < r ? 1.0 : 0.0, mb = b < r ? 1.0 : 0.0;      // basis column?
  const double ea = a == r ? 1.0 : 0.0, eb = b == r ? 1.0 : 0.0;    // the appended observation column?
  const int k0 = split * kchunk, k1 = min(K, k0 + kchunk);
  d4_t acc = {0.0, 0.0, 0.0, 0.0};
  int k = k0;
  for (; k + 4 <= k1; k += 4) {
    double xa[4][3], xb[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) regression_load(k + u, r, Q, cb, ca, cbi, ma, mb, ea, eb, xa[u], xb[u]);
#pragma unroll
    for (int u = 0; u < 4; ++u) acc = regression_mac(k + u, cb, xa[u], xb[u], kk, wt, kappa, acc);
  }
  for (; k < k1; ++k) {
    double xa[3], xb[3];
    regression_load(k, r, Q, cb, ca, cbi, ma, mb, ea, eb, xa, xb);
    acc = regression_mac(k, cb, xa, xb, kk, wt, kappa, acc);
  }
  double* out = Mpart + (size_t)split * n * n;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = 16 * ti + kk + 4 * g, col = 16 * tj + i16;
    if (row < n && col < n) out[(size_t)row * n + col] = acc[g];
  }
}

// The folded form of regression_tile (kept apart: the single-leaf function above is the one every lone chain's step runs, and a loop
// and a second accumulator around its body cost configs[2] 7 % — measured against the round-4 build on one box).
// fold > 1 (round 5): the wave takes `fold` consecutive slices — leaves of kchunk correspondences, each accumulated from zero on the
// matrix cores exactly as a lone slice is — and adds the leaves' sums IN ORDER, from 0.0, as every reader of the partials does: with
// fold = the number of slices the one partial it writes (slot `split`) holds, bit for bit, what the readers' sum over all slices would.
// Chains side by side bring parallelism of their own (30 chains x 91 tiles of the face model: 2,730 waves without any split-K): the
// 50 partials per tile — 9 MB per posterior written, read again and summed by a launch of its own, 280 MB and a third of a millisecond
// per 30-chain step — shrink to one.
__device__ __forceinline__ void regression_tile_fold(int tile, int split, int K, int kchunk, int r, const double* __restrict__ Q,
                                                     const CorrBuffers& cb, double wt, double kappa, double* __restrict__ Mpart, int fold) {
  const int n = r + 1;
  int ti = 0;
  while ((ti + 1) * (ti + 2) / 2 <= tile) ++ti;
  const int tj = tile - ti * (ti + 1) / 2;
  const int l = threadIdx.x & 63, i16 = l & 15, kk = l >> 4;
  const int a = 16 * ti + i16, b = 16 * tj + i16;
  const int ca = a < r ? a : 0, cbi = b < r ? b : 0;
  const double ma = a < r ? 1.0 : 0.0, mb = b < r ? 1.0 : 0.0;      // basis column?
  const double ea = a == r ? 1.0 : 0.0, eb = b == r ? 1.0 : 0.0;    // the appended observation column?
  d4_t acc = {0.0, 0.0, 0.0, 0.0}, run = {0.0, 0.0, 0.0, 0.0};
  const int nf = fold > 1 ? fold : 1;  // (an argument record that never heard of folding holds 0)
  for (int f = 0; f < nf; ++f) {
    const int k0 = (split * nf + f) * kchunk, k1 = min(K, k0 + kchunk);
    acc = d4_t{0.0, 0.0, 0.0, 0.0};
    int k = k0;
    for (; k + 4 <= k1; k += 4) {
      double xa[4][3], xb[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) regression_load(k + u, r, Q, cb, ca, cbi, ma, mb, ea, eb, xa[u], xb[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u) acc = regression_mac(k + u, cb, xa[u], xb[u], kk, wt, kappa, acc);
    }
    for (; k < k1; ++k) {
      double xa[3], xb[3];
      regression_load(k, r, Q, cb, ca, cbi, ma, mb, ea, eb, xa, xb);
      acc = regression_mac(k, cb, xa, xb, kk, wt, kappa, acc);
    }
    if (fold > 1) {
#pragma unroll
      for (int g = 0; g < 4; ++g) run[g] += acc[g];  // (0.0 + leaf 0) + leaf 1 + …: the readers' own order
    }
  }
  if (fold > 1) acc = run;
  double* out = Mpart + (size_t)split * n * n;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = 16 * ti + kk + 4 * g, col = 16 * tj + i16;
    if (row < n && col < n) out[(size_t)row * n + col] = acc[g];
  }
}

// MACRO TILES (round 5; folded launches only).  A wave that owns MT x MT neighbouring output tiles gathers, per correspondence, the basis
// rows of MT row blocks and MT column blocks ONCE — 6·MT gathered values for MT² matrix instructions instead of 6 per instruction:
// the launch is bound by its gathers (30 face-model chains: 8.2 M wave loads, 0.4 ms), not by the matrix cores.  Every tile is
// accumulated exactly as regression_tile accumulates it (its leaves from zero in correspondence order, the leaves' sums added in order
// from 0.0): the same bits, whatever the tiling.  mtile = index into the lower triangle of the grid of macro tiles; tiles of a
// diagonal macro tile that lie above the diagonal, and tiles beyond the matrix, are skipped.
__host__ __device__ inline int regression_macro_tiles(int r, int MT) {
  const int nt = (r + 1 + 15) >> 4, nm = (nt + MT - 1) / MT;
  return nm * (nm + 1) / 2;
}
#ifndef ICP_REG_MACRO_G
#define ICP_REG_MACRO_G 3
#endif
template <int MT>
__device__ __forceinline__ void regression_macro_fold(int mtile, int leaves, int K, int kchunk, int r, const double* __restrict__ Q,
                                                      const CorrBuffers& cb, double wt, double kappa, double* __restrict__ Mpart) {
  const int n = r + 1, nt = (n + 15) >> 4;
  int mi = 0;
  while ((mi + 1) * (mi + 2) / 2 <= mtile) ++mi;
  const int mj = mtile - mi * (mi + 1) / 2;
  const int l = threadIdx.x & 63, i16 = l & 15, kk = l >> 4;
  int ca[MT], cbi[MT];
  double ma[MT], mb[MT], ea[MT], eb[MT];
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    const int a = 16 * (MT * mi + p) + i16, b = 16 * (MT * mj + p) + i16;
    ca[p] = a < r ? a : 0; cbi[p] = b < r ? b : 0;
    ma[p] = a < r ? 1.0 : 0.0; mb[p] = b < r ? 1.0 : 0.0;
    ea[p] = a == r ? 1.0 : 0.0; eb[p] = b == r ? 1.0 : 0.0;
  }
  bool on[MT][MT];  // (uniform) tile (MT·mi + p, MT·mj + q) exists and lies in the lower triangle
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) on[p][q] = MT * mi + p < nt && MT * mj + q <= MT * mi + p;
  d4_t acc[MT][MT], run[MT][MT];
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) run[p][q] = d4_t{0.0, 0.0, 0.0, 0.0};
  constexpr int G = ICP_REG_MACRO_G;  // correspondences whose gathers are in flight together (4: 288 registers per lane, one wave per SIMD)
  for (int f = 0; f < leaves; ++f) {
    const int k0 = f * kchunk, k1 = min(K, k0 + kchunk);
#pragma unroll
    for (int p = 0; p < MT; ++p)
#pragma unroll
      for (int q = 0; q < MT; ++q) acc[p][q] = d4_t{0.0, 0.0, 0.0, 0.0};
    for (int kb = k0; kb < k1; kb += G) {
      double A_op[G][MT], B_op[G][MT];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int k = min(kb + u, k1 - 1);  // (past the leaf's end: a repeated load, its products not issued)
        const double* q_ = Q + (size_t)3 * cb.id[k] * r;
        const double e0 = cb.e[3 * k], e1 = cb.e[3 * k + 1], e2 = cb.e[3 * k + 2];
        const double n0 = cb.nhat[3 * k], n1 = cb.nhat[3 * k + 1], n2 = cb.nhat[3 * k + 2];
        const double w = (kk == 3 ? kappa : wt) * (cb.keep[k] ? 1.0 : 0.0);
#pragma unroll
        for (int p = 0; p < MT; ++p) {
          const double a0 = fma(ma[p], q_[ca[p]], ea[p] * e0), a1 = fma(ma[p], q_[r + ca[p]], ea[p] * e1), a2 = fma(ma[p], q_[2 * r + ca[p]], ea[p] * e2);
          const double b0 = fma(mb[p], q_[cbi[p]], eb[p] * e0), b1 = fma(mb[p], q_[r + cbi[p]], eb[p] * e1), b2 = fma(mb[p], q_[2 * r + cbi[p]], eb[p] * e2);
          const double va = fma(a2, n2, fma(a1, n1, a0 * n0));
          const double vb = fma(b2, n2, fma(b1, n1, b0 * n0));
          A_op[u][p] = kk == 0 ? a0 : kk == 1 ? a1 : kk == 2 ? a2 : va;
          B_op[u][p] = (kk == 0 ? b0 : kk == 1 ? b1 : kk == 2 ? b2 : vb) * w;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (kb + u < k1) {
#pragma unroll
          for (int p = 0; p < MT; ++p)
#pragma unroll
            for (int q = 0; q < MT; ++q)
              if (on[p][q]) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_op[u][p], B_op[u][q], acc[p][q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < MT; ++p)
#pragma unroll
      for (int q = 0; q < MT; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) run[p][q][g] += acc[p][q][g];
  }
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      if (!on[p][q]) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * (MT * mi + p) + kk + 4 * g, col = 16 * (MT * mj + q) + i16;
        if (row < n && col < n) Mpart[(size_t)row * n + col] = run[p][q][g];
      }
    }
}

// … from the correspondences' OPERAND ROWS (StepRegressionArgs::X, k_wide_xrows).  regression_macro_fold gathers, per macro tile and
// correspondence, three basis rows at 2 x 32 columns behind the correspondence's model id — every lane all three rows, because the
// fourth operand row is their combination with the normal —: 1.7 GB through L2 for 25 face chains, which IS the launch's 280 µs, and
// two dependent loads (id, then rows) per round of a wave's chain.  With the four operand rows of a correspondence laid out once
// (2.6 MB per posterior), lane (column, j) loads ONE value per tile row / column block — a third of the traffic, every byte of a
// 512-byte request used —, no id in between, and six correspondences in flight instead of three.  Same values into the same matrix
// instructions in the same order: the bits of regression_tile.
template <int N> struct IntK { static constexpr int value = N; };
template <int MT>
__device__ __forceinline__ void regression_macro_fold_x(int mtile, int leaves, int K, int kchunk, int r, const double* __restrict__ X, int xrs,
                                                        const CorrBuffers& cb, double wt, double kappa, double* __restrict__ Mpart) {
  // (everything but the lane's place in the tile is the same for the whole wave — said so, it lives in scalar registers and the
  // tests around the matrix instructions are scalar branches instead of exec-mask sequences)
  mtile = __builtin_amdgcn_readfirstlane(mtile); leaves = __builtin_amdgcn_readfirstlane(leaves); K = __builtin_amdgcn_readfirstlane(K);
  kchunk = __builtin_amdgcn_readfirstlane(kchunk); r = __builtin_amdgcn_readfirstlane(r); xrs = __builtin_amdgcn_readfirstlane(xrs);
  const int n = r + 1, nt = (n + 15) >> 4;
  int mi = 0;
  while ((mi + 1) * (mi + 2) / 2 <= mtile) ++mi;
  const int mj = mtile - mi * (mi + 1) / 2;
  const int l = threadIdx.x & 63, i16 = l & 15, kk = l >> 4;
  static_assert(MT == 2, "the operand rows are stored for 2 x 2 macro tiles (StepRegressionArgs::X)");
  int ca[MT], cbi[MT];  // (the two tiles of a macro block interleaved: both operands of a lane side by side; blocks past the rank hold zeros)
#pragma unroll
  for (int p = 0; p < MT; ++p) {
    ca[p] = 32 * mi + 2 * i16 + p;
    cbi[p] = 32 * mj + 2 * i16 + p;
  }
  bool on[MT][MT];  // (uniform) tile (MT·mi + p, MT·mj + q) exists and lies in the lower triangle
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) on[p][q] = MT * mi + p < nt && MT * mj + q <= MT * mi + p;
  d4_t acc[MT][MT], run[MT][MT];
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) run[p][q] = d4_t{0.0, 0.0, 0.0, 0.0};
  const double wbase = kk == 3 ? kappa : wt;
  const global_ptr<const double> xl = as_global(X) + (size_t)kk * xrs;  // this lane's operand row of correspondence 0
  const global_ptr<const unsigned char> keep_g = as_global((const unsigned char*)cb.keep);
  // (the flags of a whole leaf in one 8-byte load where the leaves are eight long and the array allows it)
  const bool keep_wide = kchunk == 8 && (__builtin_amdgcn_readfirstlane((int)(uintptr_t)cb.keep) & 7) == 0;
  const global_ptr<double> Mg = as_global(Mpart);
  constexpr int G = 8;  // correspondences whose loads are in flight together (a leaf of the usual eight: one round of loads)
#ifndef ICP_FOLD_DEPTH
#define ICP_FOLD_DEPTH 4
#endif
  constexpr int D = ICP_FOLD_DEPTH;
  if (D > 1 && kchunk <= G) {
    // Leaves of at most eight correspondences (every posterior of up to 512): a leaf is ONE round of loads and 8 x 4 matrix instructions —
    // 0.4 µs of matrix pipe behind 2 µs of waiting, 50 times in a row for the face model's 400 correspondences, one wave per SIMD
    // (profiles/r06_pmc_sq.json: 0.56 waves per SIMD, MFMA busy 17 %).  The operands of the next D − 1 leaves are requested before a
    // leaf's products are issued (raw values: the weight is applied when they are used — the same product), a ring of D register
    // buffers; products, their order within an accumulator and the leaf sums are those of the plain loop below: the same bits.
    typedef double d2v_t __attribute__((ext_vector_type(2)));
    d2v_t A_raw[D][G], B_raw[D][G];
    unsigned long long keep_raw[D];  // (byte u: the flag of the leaf's correspondence u)
    auto request = [&](auto SLOT, int f) {
      constexpr int sl = decltype(SLOT)::value;
      const int k0 = f * kchunk, k1 = min(K, k0 + kchunk);
      if (keep_wide && k0 + 8 <= K) keep_raw[sl] = *(global_ptr<const unsigned long long>)(keep_g + k0);
      else {
        unsigned long long m = 0;
#pragma unroll
        for (int u = 0; u < G; ++u) m |= (unsigned long long)keep_g[min(k0 + u, k1 - 1)] << (8 * u);
        keep_raw[sl] = m;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int k = min(k0 + u, k1 - 1);
        const global_ptr<const double> xk = xl + (size_t)k * 4 * xrs;
        A_raw[sl][u] = *(global_ptr<const d2v_t>)(xk + ca[0]);
        B_raw[sl][u] = *(global_ptr<const d2v_t>)(xk + cbi[0]);
      }
    };
    auto products = [&](auto SLOT, int f) {
      constexpr int sl = decltype(SLOT)::value;
      const int k0 = f * kchunk, k1 = min(K, k0 + kchunk);
#pragma unroll
      for (int p = 0; p < MT; ++p)
#pragma unroll
        for (int q = 0; q < MT; ++q) acc[p][q] = d4_t{0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (k0 + u < k1) {
          const double w = wbase * (((keep_raw[sl] >> (8 * u)) & 0xff) ? 1.0 : 0.0);
          const double a[MT] = {A_raw[sl][u].x, A_raw[sl][u].y}, b[MT] = {B_raw[sl][u].x * w, B_raw[sl][u].y * w};
#pragma unroll
          for (int p = 0; p < MT; ++p)
#pragma unroll
            for (int q = 0; q < MT; ++q)
              if (on[p][q]) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[p], b[q], acc[p][q], 0, 0, 0);
        }
      }
#pragma unroll
      for (int p = 0; p < MT; ++p)
#pragma unroll
        for (int q = 0; q < MT; ++q)
#pragma unroll
          for (int g = 0; g < 4; ++g) run[p][q][g] += acc[p][q][g];
    };
    auto phase = [&](auto S, int f0) {
      constexpr int s = decltype(S)::value;
      const int f = f0 + s;
      if (f + D - 1 < leaves) request(IntK<(s + D - 1) % D>{}, f + D - 1);  // (into the buffer the previous phase used up)
      __builtin_amdgcn_sched_barrier(0);                                    // requested BEFORE this leaf's products, not next to their use
      if (f < leaves) products(S, f);
    };
    if (0 < leaves && D > 1) request(IntK<0>{}, 0);
    if (1 < leaves && D > 2) request(IntK<1 % D>{}, 1);
    if (2 < leaves && D > 3) request(IntK<2 % D>{}, 2);
    for (int f0 = 0; f0 < leaves; f0 += D) {
      phase(IntK<0>{}, f0);
      if constexpr (D > 1) phase(IntK<1 % D>{}, f0);
      if constexpr (D > 2) phase(IntK<2 % D>{}, f0);
      if constexpr (D > 3) phase(IntK<3 % D>{}, f0);
    }
  } else
  for (int f = 0; f < leaves; ++f) {
    const int k0 = f * kchunk, k1 = min(K, k0 + kchunk);
#pragma unroll
    for (int p = 0; p < MT; ++p)
#pragma unroll
      for (int q = 0; q < MT; ++q) acc[p][q] = d4_t{0.0, 0.0, 0.0, 0.0};
    for (int kb = k0; kb < k1; kb += G) {
      double A_op[G][MT], B_op[G][MT];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int k = min(kb + u, k1 - 1);  // (past the leaf's end: a repeated load, its products not issued)
        const global_ptr<const double> xk = xl + (size_t)k * 4 * xrs;
        const double w = wbase * (keep_g[k] ? 1.0 : 0.0);
#pragma unroll
        for (int p = 0; p < MT; ++p) {
          A_op[u][p] = xk[ca[p]];
          B_op[u][p] = xk[cbi[p]] * w;
        }
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        if (kb + u < k1) {
#pragma unroll
          for (int p = 0; p < MT; ++p)
#pragma unroll
            for (int q = 0; q < MT; ++q)
              if (on[p][q]) acc[p][q] = __builtin_amdgcn_mfma_f64_16x16x4f64(A_op[u][p], B_op[u][q], acc[p][q], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int p = 0; p < MT; ++p)
#pragma unroll
      for (int q = 0; q < MT; ++q)
#pragma unroll
        for (int g = 0; g < 4; ++g) run[p][q][g] += acc[p][q][g];
  }
#pragma unroll
  for (int p = 0; p < MT; ++p)
#pragma unroll
    for (int q = 0; q < MT; ++q) {
      if (!on[p][q]) continue;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * (MT * mi + p) + kk + 4 * g, col = 16 * (MT * mj + q) + i16;
        if (row < n && col < n) Mg[(size_t)row * n + col] = run[p][q][g];
      }
    }
}
// the operand rows of one correspondence, one thread per (correspondence, column): the values regression_load / regression_mac form
__device__ __forceinline__ void regression_xrows(int e, int K, int r, int xrs, const double* __restrict__ Q, const CorrBuffers& cb, double* __restrict__ X) {
  const int k = e / xrs, col = e - k * xrs;
  if (k >= K) return;
  const double* q = Q + (size_t)3 * cb.id[k] * r;
  const double ma = col < r ? 1.0 : 0.0, ea = col == r ? 1.0 : 0.0;
  const int cc = col < r ? col : 0;
  const double e0 = cb.e[3 * k], e1 = cb.e[3 * k + 1], e2 = cb.e[3 * k + 2];
  const double a0 = fma(ma, q[cc], ea * e0), a1 = fma(ma, q[r + cc], ea * e1), a2 = fma(ma, q[2 * r + cc], ea * e2);
  const double n0 = cb.nhat[3 * k], n1 = cb.nhat[3 * k + 1], n2 = cb.nhat[3 * k + 2];
  double* x = X + (size_t)k * 4 * xrs + ((col & ~31) | ((col & 15) << 1) | ((col >> 4) & 1));  // (interleaved: StepRegressionArgs::X)
  x[0] = a0; x[xrs] = a1; x[2 * (size_t)xrs] = a2; x[3 * (size_t)xrs] = fma(a2, n2, fma(a1, n1, a0 * n0));
}

// ---------------------------------------------------------------- dense helpers (one workgroup, matrix behind a generic pointer)

__device__ double block_sum(double v, double* s_red) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)((blockDim.x + 63) >> 6); ++w) t += s_red[w];
  return t;
}
__device__ double block_max(double v, double* s_red) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = s_red[0];
  for (int w = 1; w < (int)((blockDim.x + 63) >> 6); ++w) t = fmax(t, s_red[w]);
  return t;
}

// hardware reciprocal / reciprocal square root seeds + two Newton steps (≈ 1 ulp); the dependent chains of these small
// factorisations are latency bound, and the IEEE division / sqrt expansions are 3-5× longer
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  double e = fma(-x, y, 1.0);
  y = fma(y, e, y);
  e = fma(-x, y, 1.0);
  return fma(y, e, y);
}
__device__ __forceinline__ double fast_rsqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  return fma(y, e, y);
}

// y = A x for a row-major r×r matrix A (leading dimension lda; LDS or global), x and y in LDS.  2^tpr_log2 threads share
// a row (their partial sums meet through wave shuffles), blockDim/2^tpr_log2 rows per pass.  Ends with a barrier.
__device__ void block_matvec(int r, const double* A, int lda, const double* x, double* y, int tpr_log2) {
  const int tid = threadIdx.x, tpr = 1 << tpr_log2, sub = tid & (tpr - 1);
  const int rows_per_pass = blockDim.x >> tpr_log2;
  for (int row0 = 0; row0 < r; row0 += rows_per_pass) {  // uniform trip count: every lane reaches the shuffles
    const int row = row0 + (tid >> tpr_log2);
    double acc = 0.0;
    if (row < r) {
      // sixteen matrix entries in flight per lane, requested BEFORE the first one is used (left to itself the compiler keeps a load
      // next to its multiply-add: one round trip to L2 per entry — 50 in a row for a rank-200 row shared by four lanes); the sum
      // runs over the entries in their order either way
      const double* a = A + (size_t)row * lda;
      constexpr int CH = 16;
      for (int j0 = sub; j0 < r; j0 += CH * tpr) {
        double av[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int j = j0 + q * tpr;
          av[q] = j < r ? a[j] : 0.0;
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int j = j0 + q * tpr;
          if (j < r) acc = fma(av[q], x[j], acc);
        }
      }
    }
    for (int o = tpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < r && sub == 0) y[row] = acc;
  }
  __syncthreads();
}

// same with the matrix in the dynamic LDS segment at s_dyn[offA ...]: indexing s_dyn directly keeps the loads DS
// instructions (a generic pointer to LDS would make them FLAT, several times slower)
extern __shared__ __attribute__((aligned(16))) double s_dyn[];
__device__ void block_matvec_lds(int r, int offA, int lda, const double* x, double* y, int tpr_log2) {
  const int tid = threadIdx.x, tpr = 1 << tpr_log2, sub = tid & (tpr - 1);
  const int rows_per_pass = blockDim.x >> tpr_log2;
  for (int row0 = 0; row0 < r; row0 += rows_per_pass) {
    const int row = row0 + (tid >> tpr_log2);
    double acc = 0.0;
    if (row < r) {
      const int a = offA + row * lda;
#pragma unroll 8
      for (int j = sub; j < r; j += tpr) acc = fma(s_dyn[a + j], x[j], acc);
    }
    for (int o = tpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < r && sub == 0) y[row] = acc;
  }
  __syncthreads();
}

// threads per row for block_matvec: enough rows in flight to occupy the block, at least ~8 terms per thread
static inline int matvec_tpr_log2(int r, int block) {
  int t = 0;
  while (t < 6 && (r << (t + 1)) <= block && (r >> (t + 1)) >= 8) ++t;
  return t;
}

// copy a row-major r×r matrix into LDS with leading dimension ld
__device__ void stage_matrix(int r, const double* __restrict__ src, double* dst, int ld) {
  for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e - i * r;
    dst[(size_t)i * ld + j] = src[e];
  }
}
__device__ void stage_matrix_lds(int r, const double* __restrict__ src, int off, int ld) {
  for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e - i * r;
    s_dyn[off + i * ld + j] = src[e];
  }
}

// Root-free right-looking Cholesky of the leading n×n block of W (leading dimension ld), in place, carrying `extra`
// more rows below it through the same eliminations.  Column j is left UNSCALED (U[i][j] = L[i][j]·l_jj) and the
// trailing update uses U[i][j]·U[k][j]/U[j][j], so each column costs ONE reciprocal and ONE barrier.
// Afterwards L[i][j] = W[i][j]·dinv[j] with dinv[j] = 1/sqrt(W[j][j]).  2-D thread grid of tw×tw (tw² = blockDim).
__device__ bool block_cholesky_rootfree(double* W, int n, int ld, int extra, int tw_log2) {
  const int tid = threadIdx.x, tw = 1 << tw_log2, ty = tid >> tw_log2, tx = tid & (tw - 1);
  const int rows = n + extra;
  for (int j = 0; j < n; ++j) {
    const double ajj = W[(size_t)j * ld + j];
    if (!(ajj > 0.0)) return false;  // same value in every thread: uniform exit
    const double inv = fast_rcp(ajj);
    for (int i = j + 1 + ty; i < rows; i += tw) {
      const double uij = W[(size_t)i * ld + j] * inv;
      const int kmax = i < n ? i : n - 1;
      for (int k = j + 1 + tx; k <= kmax; k += tw) W[(size_t)i * ld + k] = fma(-uij, W[(size_t)k * ld + j], W[(size_t)i * ld + k]);
    }
    __syncthreads();
  }
  return true;
}

// ---------------------------------------------------------------- K5b: M = I + Σ partials, chol(M), α = M⁻¹ b

// Fast path (the factor fits LDS): every thread OWNS 2×4 tiles of the lower triangle of [M; bᵀ] (rows 2ti, 2ti+1,
// columns 4tj … 4tj+3) and keeps them in registers for the whole root-free factorisation.  Only the current pivot column
// is published through a double-buffered LDS vector, so a column step is: three 16-byte loads (the tile's two row entries
// and four column entries of the pivot column), one reciprocal, 2 + 8 multiply-adds per tile, one barrier.  A tile also
// covers entries above the diagonal and entries that are already final; those receive throw-away updates instead of
// predicates (their true values were stored the moment they became final).  The column loop is unrolled four times so
// that the tile column being published and the pivot-column buffer are compile-time constants.
#ifdef ICP_EIGEN_TIMING
extern __device__ long long g_eigen_stamps[64];
#define FAC_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_eigen_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define FAC_STAMP(i)
#endif

typedef double dense2 __attribute__((ext_vector_type(2)));

// tiles of the lower triangle, row-major over tile rows: tile row ti holds tile columns 0 … ti/2
static __host__ __device__ inline int factor_tile_count(int r) {
  const int tr = (r + 2) >> 1;  // tile rows over the r+1 rows (M and bᵀ)
  int c = 0;
  for (int t = 0; t < tr; ++t) c += (t >> 1) + 1;
  return c;
}

template <int TPT, int NT>
// tail_base >= 0: the caller runs a transition tail on this posterior right afterwards — the assembled M is ALSO written
// to s_dyn[tail_base + i·ld + j] and G⁻¹ (tail_Ginv) to s_dyn[tail_base + r·ld + …] (layout of tail_body), its loads
// issued beside the partial sums', so that the tail finds both matrices in LDS instead of fetching them again.
// Mplain != nullptr: the matrix comes assembled (row-major r × r, symmetric, the identity already in it) instead of as split-K
// partials — nothing is written to M, the appended row is zero and no α is solved (alpha_out unused): k_posterior_root.
__device__ __forceinline__ bool factor_reg_body(int r, const double* __restrict__ Mpart, int S, double* __restrict__ M,
                                                double* __restrict__ alpha_out, int* __restrict__ status, int tail_base = -1,
                                                const double* __restrict__ tail_Ginv = nullptr, const double* __restrict__ Mplain = nullptr,
                                                bool solve = true) {
  __shared__ __attribute__((aligned(16))) double s_col[2][520];
  __shared__ double s_dinv[512], s_v[512];
  const int tid = threadIdx.x, n = r + 1;
  const int ld = r | 1;
  double* W = s_dyn;  // (r+1) × ld: the finished (unscaled) factor, row r = forward-eliminated bᵀ
  FAC_STAMP(16);
  constexpr int kGinvPer = 16;  // elements of G⁻¹ per thread held in flight (covers r² <= 16·NT; the rest goes the slow way)
  double ginv[kGinvPer];
  if (tail_base >= 0) {
#pragma unroll
    for (int u = 0; u < kGinvPer; ++u) {
      const int e = tid + u * NT;
      ginv[u] = e < r * r ? tail_Ginv[e] : 0.0;
    }
  }
  const int n_tiles = factor_tile_count(r);
  double v[TPT][2][4];
  int R0[TPT], C0[TPT];
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    const int e = tid + NT * t;
    int ti = 0, base = 0;
    if (e < n_tiles) {
      while (base + (ti >> 1) + 1 <= e) { base += (ti >> 1) + 1; ++ti; }
    }
    R0[t] = e < n_tiles ? 2 * ti : -2;            // -2: slot unused
    C0[t] = e < n_tiles ? 4 * (e - base) : 0;
    // sum of the split partials: split-major so that the eight loads of a split are in flight together
    bool live[2][4];
    size_t off[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = R0[t] + a, k = C0[t] + c;
        live[a][c] = R0[t] >= 0 && i < n && k < r && k <= i;
        off[a][c] = live[a][c] ? (size_t)i * n + k : 0;
        v[t][a][c] = 0.0;
      }
    const size_t nn = (size_t)n * n;
    int sp = 0;
    if (Mplain) {  // (uniform) the assembled matrix: one load per entry, the mirrored pair averaged like the decompositions do
      sp = S;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int i = R0[t] + a, k = C0[t] + c;
          if (live[a][c] && i < r) v[t][a][c] = 0.5 * (Mplain[(size_t)i * r + k] + Mplain[(size_t)k * r + i]);
        }
    }
    for (; sp + 4 <= S; sp += 4) {  // four splits (32 loads) in flight; summed in split order
      double p[4][2][4];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) p[u][a][c] = Mpart[(size_t)(sp + u) * nn + off[a][c]];
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int c = 0; c < 4; ++c) v[t][a][c] += p[u][a][c];
    }
    for (; sp < S; ++sp) {
      const double* P = Mpart + (size_t)sp * nn;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int c = 0; c < 4; ++c) v[t][a][c] += P[off[a][c]];
    }
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = R0[t] + a, k = C0[t] + c;
        double x = live[a][c] ? v[t][a][c] : 0.0;
        if (live[a][c]) {
          if (i < r && !Mplain) {
            if (i == k) x += 1.0;
            M[(size_t)i * r + k] = x;
            M[(size_t)k * r + i] = x;
            if (tail_base >= 0) { s_dyn[tail_base + i * ld + k] = x; s_dyn[tail_base + k * ld + i] = x; }
          }
          if (k == 0) { s_col[0][i] = x; W[(size_t)i * ld] = x; }
        }
        v[t][a][c] = x;
      }
  }
  if (tail_base >= 0) {
    const int offG = tail_base + r * ld;
#pragma unroll
    for (int u = 0; u < kGinvPer; ++u) {
      const int e = tid + u * NT;
      if (e < r * r) { const int i = e / r; s_dyn[offG + i * ld + (e - i * r)] = ginv[u]; }
    }
    for (int e = tid + kGinvPer * NT; e < r * r; e += NT) { const int i = e / r; s_dyn[offG + i * ld + (e - i * r)] = tail_Ginv[e]; }
  }
  __syncthreads();
  FAC_STAMP(17);
  bool ok = true;
  for (int j4 = 0; j4 < r && ok; j4 += 4) {
#pragma unroll
    for (int cc = 0; cc < 4; ++cc) {
      const int j = j4 + cc;
      if (j >= r) break;                       // uniform
      const double* cur = s_col[cc & 1];      // j & 1 == cc & 1
      double* nxt = s_col[(cc + 1) & 1];
      const double ajj = cur[j];
      dense2 u[TPT], k01[TPT], k23[TPT];
#pragma unroll
      for (int t = 0; t < TPT; ++t) {
        const int rr = R0[t] < 0 ? 0 : R0[t];
        u[t] = *(const dense2*)&cur[rr];
        k01[t] = *(const dense2*)&cur[C0[t]];
        k23[t] = *(const dense2*)&cur[C0[t] + 2];
      }
      if (!(ajj > 0.0)) { ok = false; break; }  // same value in every thread: uniform exit
      const double inv = fast_rcp(ajj);
      const int kp = j + 1;                      // the column that becomes final in this step
#pragma unroll
      for (int t = 0; t < TPT; ++t) {
        const double m0 = -(u[t].x * inv), m1 = -(u[t].y * inv);
        v[t][0][0] = fma(m0, k01[t].x, v[t][0][0]); v[t][0][1] = fma(m0, k01[t].y, v[t][0][1]);
        v[t][0][2] = fma(m0, k23[t].x, v[t][0][2]); v[t][0][3] = fma(m0, k23[t].y, v[t][0][3]);
        v[t][1][0] = fma(m1, k01[t].x, v[t][1][0]); v[t][1][1] = fma(m1, k01[t].y, v[t][1][1]);
        v[t][1][2] = fma(m1, k23[t].x, v[t][1][2]); v[t][1][3] = fma(m1, k23[t].y, v[t][1][3]);
        constexpr int cp = 0;  // placeholder to keep the unrolled structure readable
        (void)cp;
        if (kp < r && R0[t] >= 0 && C0[t] == (kp & ~3)) {  // this tile holds column kp at tile column (cc+1)&3
          const double p0 = v[t][0][(cc + 1) & 3], p1 = v[t][1][(cc + 1) & 3];
          *(dense2*)&nxt[R0[t]] = dense2{p0, p1};
          if (R0[t] >= kp && R0[t] < n) W[(size_t)R0[t] * ld + kp] = p0;
          if (R0[t] + 1 >= kp && R0[t] + 1 < n) W[(size_t)(R0[t] + 1) * ld + kp] = p1;
        }
      }
      __syncthreads();
    }
  }
  if (!ok) {
    if (tid == 0) status[0] = 1;
    return false;
  }
  FAC_STAMP(18);
  if (tid == 0) status[0] = 0;
  if (!solve) return true;  // (uniform)
  // y = L⁻¹ b sits (unscaled) in row r: y_j = W[r][j]·dinv_j,  L[i][j] = W[i][j]·dinv_j
  for (int j = tid; j < r; j += NT) {
    const double d = fast_rsqrt(W[(size_t)j * ld + j]);
    s_dinv[j] = d;
    s_v[j] = W[(size_t)r * ld + j] * d;
  }
  __syncthreads();
  FAC_STAMP(19);
  if (r <= 64) {
    if (tid < 64) {  // one wave, registers + cross-lane reads: no barriers on the sequential chain; the rows of L are
      // fetched (and scaled) four steps ahead, so the chain per step is readlane -> multiply -> fma
      const int i = tid, ic = i < r ? i : r - 1;  // lanes past r mirror lane r-1 (their result is discarded)
      double x = s_v[ic];
      const double di = s_dinv[ic];
      constexpr int kA = 4;
      // branch-free: steps with j < 0 (padding of the last group) read row 0 and change nothing
      double lq[kA], dq[kA];
#pragma unroll
      for (int a = 0; a < kA; ++a) {
        const int jj = max(r - 1 - a, 0);
        lq[a] = W[(size_t)jj * ld + ic];  // raw: scaled at use, so nothing waits on the load here
        dq[a] = s_dinv[jj];
      }
      for (int j0 = r - 1; j0 >= 0; j0 -= kA) {
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          const int j = j0 - a;
          const double lij = lq[a] * di, dj = dq[a];
          const int jn = max(j - kA, 0);
          lq[a] = W[(size_t)jn * ld + ic];
          dq[a] = s_dinv[jn];
          const int lo = __builtin_amdgcn_readlane(__double2loint(x), j & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), j & 63);
          const double xj = __hiloint2double(hi, lo) * dj;
          const double upd = fma(-lij, xj, x);
          x = i == j ? xj : (i < j ? upd : x);
        }
      }
      if (i < r) alpha_out[i] = x;
    }
  } else {
    // Blocks of 64 unknowns from the bottom: a block's triangle by ONE wave exactly as above (registers, v_readlane, rows of L from
    // LDS four steps ahead), then every thread takes the block's contribution off the unknowns above it — NT/64 groups of lanes
    // share the block's columns, their partial sums meet in the (now idle) pivot-column buffer.  Two barriers per BLOCK where the
    // column-by-column form had two per column: 18.6 -> 5 µs at rank 101.
    double* s_bp = &s_col[0][0];  // [NT/64 <= 16][64]
    for (int b1 = r; b1 > 0; b1 -= 64) {
      const int b0 = b1 > 64 ? b1 - 64 : 0, nb = b1 - b0;
      if (tid < 64) {
        const int i = tid, ic = i < nb ? i : nb - 1;  // lanes past the block mirror its last lane (their result is discarded)
        double x = s_v[b0 + ic];
        const double di = s_dinv[b0 + ic];
        constexpr int kA = 4;
        double lq[kA], dq[kA];
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          const int jj = max(nb - 1 - a, 0);
          lq[a] = W[(size_t)(b0 + jj) * ld + b0 + ic];
          dq[a] = s_dinv[b0 + jj];
        }
        for (int j0 = nb - 1; j0 >= 0; j0 -= kA) {
#pragma unroll
          for (int a = 0; a < kA; ++a) {
            const int j = j0 - a;                       // (steps with j < 0, the padding of the last group, change nothing)
            const double lij = lq[a] * di, dj = dq[a];
            const int jn = max(j - kA, 0);
            lq[a] = W[(size_t)(b0 + jn) * ld + b0 + ic];
            dq[a] = s_dinv[b0 + jn];
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), j & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), j & 63);
            const double xj = __hiloint2double(hi, lo) * dj;
            const double upd = fma(-lij, xj, x);
            x = j < 0 ? x : (i == j ? xj : (i < j ? upd : x));
          }
        }
        if (i < nb) s_v[b0 + i] = x;
      }
      __syncthreads();
      if (b0 > 0) {
        // (EIGHT groups whatever the block size: the grouping decides how the sums are rounded, and the merged step's copy of this
        // body — 512 threads at ranks 65..116 — and the factor launch's — 1,024 — must give the same alpha bit for bit; with sixteen
        // groups in the latter they differed in the last place, found by the on-device loop of those ranks, round 5)
        constexpr int G = NT / 64 < 8 ? NT / 64 : 8;
        const int g = tid >> 6, li = tid & 63, per = (nb + G - 1) / G;
        for (int i0 = 0; i0 < b0; i0 += 64) {
          const int i = i0 + li;
          double acc = 0.0;
          if (i < b0 && g < G) {
            const int j1 = min(nb, (g + 1) * per);
            for (int j = g * per; j < j1; ++j) acc = fma(W[(size_t)(b0 + j) * ld + i], s_v[b0 + j], acc);
          }
          if (g < G) s_bp[g * 64 + li] = acc;
          __syncthreads();
          if (g == 0 && i < b0) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < G; ++q) t += s_bp[q * 64 + li];
            s_v[i] = fma(-s_dinv[i], t, s_v[i]);
          }
          __syncthreads();
        }
      }
    }
    for (int i = tid; i < r; i += NT) alpha_out[i] = s_v[i];
  }
  FAC_STAMP(20);
  return true;
}

// ---------------------------------------------------------------- a9 transition tails (batched, one workgroup each)
// (G + σ²M) γ = G d  ⇔  γ = d − σ² G⁻¹ M γ : fixed-point iteration with contraction factor ρ(σ² G⁻¹ M) (≈ 4e-8 for the
// femur model), run to machine precision; status != 0 if it does not contract, and the host then uses the direct
// (Cholesky) kernel below.  M and G⁻¹ are staged in LDS when they fit.

__device__ __forceinline__ void tail_body(int r, const double* __restrict__ alpha, const double* __restrict__ Mg, const double* __restrict__ c_from,
                                          const double* __restrict__ c_to, double step, double* __restrict__ out, int* __restrict__ status,
                                          const double* __restrict__ Ginv, double sigma2, int n_lds, int tpr_log2,
                                          int lds_base = 0, bool prestaged = false) {
  __shared__ double s_d[512], s_g[512], s_t[512], s_u[512], s_red3[3][16];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ld = r | 1;
  const int offM = lds_base, offG = lds_base + r * ld;
  if (!prestaged) {
    if (n_lds >= 1) stage_matrix_lds(r, Mg, offM, ld);
    if (n_lds >= 2) stage_matrix_lds(r, Ginv, offG, ld);
  }
  auto mul_M = [&](const double* x, double* y) {
    if (n_lds >= 1) block_matvec_lds(r, offM, ld, x, y, tpr_log2);
    else block_matvec(r, Mg, r, x, y, tpr_log2);
  };
  auto mul_Ginv = [&](const double* x, double* y) {
    if (n_lds >= 2) block_matvec_lds(r, offG, ld, x, y, tpr_log2);
    else block_matvec(r, Ginv, r, x, y, tpr_log2);
  };
  for (int i = tid; i < r; i += nt) {
    const double d = (c_from[i] + (c_to[i] - c_from[i]) / step) - alpha[i];  // :79 minus posterior mean
    s_d[i] = d;
    s_g[i] = d;
  }
  __syncthreads();
  // Stop on the PREDICTED error of the current iterate: with the contraction factor ρ_k = δ_k/δ_{k-1} observed so far,
  // ‖γ − g_k‖ <= δ_k·ρ/(1 − ρ).  For the femur models ρ ≈ 4e-8 and the second iterate is exact to rounding.  The quadratic
  // form uses M·g_{k-1} of the last iteration (it differs from M·g_k by O(δ_k), i.e. by rounding).
  int converged = 0;
  double delta_prev = 0.0, q = 0.0;
  for (int it = 0; it < 12 && !converged; ++it) {
    mul_M(s_g, s_t);
    mul_Ginv(s_t, s_u);
    double delta = 0.0, gmax = 0.0, part = 0.0;
    for (int i = tid; i < r; i += nt) {
      const double gn = fma(-sigma2, s_u[i], s_d[i]);
      delta = fmax(delta, fabs(gn - s_g[i]));
      gmax = fmax(gmax, fabs(gn));
      part = fma(gn, s_t[i], part);
      s_g[i] = gn;
    }
    // one combined reduction: max, max, sum
    for (int o = 32; o > 0; o >>= 1) {
      delta = fmax(delta, __shfl_down(delta, o, 64));
      gmax = fmax(gmax, __shfl_down(gmax, o, 64));
      part += __shfl_down(part, o, 64);
    }
    __syncthreads();
    if ((tid & 63) == 0) { s_red3[0][tid >> 6] = delta; s_red3[1][tid >> 6] = gmax; s_red3[2][tid >> 6] = part; }
    __syncthreads();
    delta = s_red3[0][0]; gmax = s_red3[1][0]; q = 0.0;
    for (int w = 0; w < (int)((nt + 63) >> 6); ++w) { delta = fmax(delta, s_red3[0][w]); gmax = fmax(gmax, s_red3[1][w]); q += s_red3[2][w]; }
    const double rho = it > 0 && delta_prev > 0.0 ? delta / delta_prev : 1.0;
    converged = gmax == 0.0 || delta == 0.0 || (it > 0 && rho < 0.5 && delta * rho <= 1e-16 * gmax * (1.0 - rho));
    delta_prev = delta;
    __syncthreads();
  }
  if (tid == 0) {
    out[0] = -0.5 * q - 0.5 * (double)r * 1.8378770664093453;  // ln(2π); no log-det term
    status[0] = converged ? 0 : 3;
  }
}

// ---------------------------------------------------------------- a8 propose (one workgroup)
// c_new = (G + σ²I)⁻¹ G w = w − σ² P w with P = (G + σ²I)⁻¹ precomputed;  w = α + D⁻¹ V (√S ∘ z);  c' = c + step·(c_new − c).
// c_out may be LDS or global; ends with a barrier.
// kStaged (k_propose only; blockDim a multiple of 256): the Cholesky-root branch stages each 64 × 64 triangle of L in LDS before its
// one-wave chain (rows from L2 four steps ahead were 8 µs of a block, from LDS 1 µs) and splits the update of the unknowns above a
// block over blockDim/256 thread groups.  One block (ranks <= 64): the same operations in the same order as the unstaged form.
template <bool kStaged = false>
__device__ __forceinline__ void propose_body(int r, const ProposeIn& in, double* c_out, int tpr_log2) {
  __shared__ double s_px[512], s_py[512], s_pw[512];
  const int tid = threadIdx.x, nt = blockDim.x;
  if (kStaged && in.root) {
    __shared__ double s_blk[64 * 65], s_part[4 * 256];
    constexpr int ldd = 65;
    for (int i = tid; i < r; i += nt) s_pw[i] = in.z[i];
    for (int b1 = r; b1 > 0; b1 -= 64) {
      const int b0 = b1 > 64 ? b1 - 64 : 0, nb = b1 - b0;
      for (int e = tid; e < 64 * 64; e += nt) {
        const int i = e >> 6, j = e & 63;
        if (j <= i && i < nb) s_blk[i * ldd + j] = in.V[(size_t)(b0 + i) * r + b0 + j];
      }
      __syncthreads();
      if (tid < 64) {
        const int i = tid, ic = i < nb ? i : nb - 1;  // lanes past the block mirror its last lane (their result is discarded)
        double x = s_pw[b0 + ic];
        const double di = in.S[b0 + ic];
        constexpr int kA = 4;
        double lq[kA];
#pragma unroll
        for (int a = 0; a < kA; ++a) lq[a] = s_blk[max(nb - 1 - a, ic) * ldd + ic];
        for (int j0 = nb - 1; j0 >= 0; j0 -= kA) {
#pragma unroll
          for (int a = 0; a < kA; ++a) {
            const int j = j0 - a;                       // (steps with j < 0, the padding of the last group, change nothing)
            const double lij = lq[a];
            lq[a] = s_blk[max(j - kA, ic) * ldd + ic];
            const int js = j & 63;
            const double xr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), js), __builtin_amdgcn_readlane(__double2loint(x), js));
            const double dj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(di), js), __builtin_amdgcn_readlane(__double2loint(di), js));
            const double xj = xr * dj;
            const double upd = fma(-lij, xj, x);
            x = j < 0 ? x : (i == j ? xj : (i < j ? upd : x));
          }
        }
        if (i < nb) s_pw[b0 + i] = x;
      }
      __syncthreads();
      if (b0 > 0) {  // the unknowns above lose this block's contribution: thread = (group of the block's columns, unknown)
        const int G = nt >> 8, g = tid >> 8, per = (nb + G - 1) / G;
        for (int i0 = 0; i0 < b0; i0 += 256) {
          const int i = i0 + (tid & 255);
          if (i < b0) {
            double acc = 0.0;
            const int j1 = min(nb, (g + 1) * per);
#pragma unroll 16
            for (int j = g * per; j < j1; ++j) acc = fma(in.V[(size_t)(b0 + j) * r + i], s_pw[b0 + j], acc);
            s_part[g * 256 + (tid & 255)] = acc;
          }
          __syncthreads();
          if (g == 0 && i < b0) {
            double t = 0.0;
            for (int q = 0; q < G; ++q) t += s_part[q * 256 + tid];
            s_pw[i] -= t;
          }
          __syncthreads();
        }
      }
    }
    for (int i = tid; i < r; i += nt) s_pw[i] += in.alpha[i];
    __syncthreads();
  } else if (in.root) {
    // Cholesky-root sampler: u = L⁻ᵀ z (in.V = L row-major, in.S = 1/diag L), 64 unknowns at a time from the bottom.  A block's
    // triangle by ONE wave — lane i carries u_i, rows of L arrive four steps ahead, the chain per step is readlane -> multiply ->
    // fma (the back substitution of factor_reg_body with z as the right-hand side) — then every thread takes the block's
    // contribution off the unknowns above it.  Ranks <= 64 are one block.  w = α + u.
    for (int i = tid; i < r; i += nt) s_pw[i] = in.z[i];
    __syncthreads();
    for (int b1 = r; b1 > 0; b1 -= 64) {
      const int b0 = b1 > 64 ? b1 - 64 : 0, nb = b1 - b0;
      if (tid < 64) {
        const int i = tid, ic = i < nb ? i : nb - 1;  // lanes past the block mirror its last lane (their result is discarded)
        double x = s_pw[b0 + ic];
        constexpr int kA = 4;
        double lq[kA], dq[kA];
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          const int jj = max(nb - 1 - a, 0);
          lq[a] = in.V[(size_t)(b0 + jj) * r + b0 + ic];
          dq[a] = in.S[b0 + jj];
        }
        for (int j0 = nb - 1; j0 >= 0; j0 -= kA) {
#pragma unroll
          for (int a = 0; a < kA; ++a) {
            const int j = j0 - a;                       // (steps with j < 0, the padding of the last group, change nothing)
            const double lij = lq[a], dj = dq[a];
            const int jn = max(j - kA, 0);
            lq[a] = in.V[(size_t)(b0 + jn) * r + b0 + ic];
            dq[a] = in.S[b0 + jn];
            const int lo = __builtin_amdgcn_readlane(__double2loint(x), j & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), j & 63);
            const double xj = __hiloint2double(hi, lo) * dj;
            const double upd = fma(-lij, xj, x);
            x = j < 0 ? x : (i == j ? xj : (i < j ? upd : x));
          }
        }
        if (i < nb) s_pw[b0 + i] = x;
      }
      __syncthreads();
      for (int i = tid; i < b0; i += nt) {  // the unknowns above lose this block's contribution (row b0 + j of L, coalesced over i)
        double acc = s_pw[i];
#pragma unroll 8
        for (int j = 0; j < nb; ++j) acc = fma(-in.V[(size_t)(b0 + j) * r + i], s_pw[b0 + j], acc);
        s_pw[i] = acc;
      }
      __syncthreads();
    }
    for (int i = tid; i < r; i += nt) s_pw[i] += in.alpha[i];
    __syncthreads();
  } else {
    for (int j = tid; j < r; j += nt) s_px[j] = sqrt(in.S[j]) * in.z[j];
    __syncthreads();
    block_matvec(r, in.V, r, s_px, s_py, tpr_log2);
    for (int i = tid; i < r; i += nt) s_pw[i] = fma(s_py[i], in.inv_sqrt_lambda[i], in.alpha[i]);
    __syncthreads();
  }
  block_matvec(r, in.P, r, s_pw, s_py, tpr_log2);
  for (int i = tid; i < r; i += nt) {
    const double cnew = fma(-in.sigma2, s_py[i], s_pw[i]);  // model.coefficients(...) with σ² = 1e-5 (NonRigidIcpProposal.scala:59)
    c_out[i] = in.c[i] + (cnew - in.c[i]) * in.step;         // :61-62
  }
  __syncthreads();
}

// direct version: factor G + σ²M, solve, quadratic form.  Used only if the iteration above reports non-contraction.
// Σ_k log N(sqrt(d2_k); mean, sigma) (Breeze Gaussian.logPdf), one workgroup
__device__ __forceinline__ void sum_gauss_logpdf_body(int K, const double* __restrict__ d2, double mean, double sigma, double* __restrict__ out) {
  __shared__ double s_red_g[16];
  const double lognorm = log(sqrt(2.0 * 3.14159265358979323846)) + log(sigma);  // Breeze Gaussian.logNormalizer
  double part = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double d = (sqrt(d2[k]) - mean) / sigma;
    part += -d * d / 2.0 - lognorm;
  }
  double t = block_sum(part, s_red_g);
  if (threadIdx.x == 0) out[0] = t;
}

}  // namespace icp
