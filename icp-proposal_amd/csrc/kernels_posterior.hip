// kernels_posterior.hip — correspondences -> GP regression -> r-space closed forms, and evaluator reductions.
//
// Math: SURVEY.md App. A.  With Q = Φ·diag(√λ), per-correspondence noise Σ_i = σ_t² I + (σ_n² − σ_t²) n̂n̂ᵀ
// (SurfaceNoiseHelpers.scala:32-60 for an orthonormal frame), so Σ_i⁻¹ = w_t I + κ n̂n̂ᵀ, w_t = 1/σ_t², κ = 1/σ_n² − 1/σ_t²:
//   M = I + Σ_i Q_iᵀ Σ_i⁻¹ Q_i,  b = Σ_i Q_iᵀ Σ_i⁻¹ (y_i − μ_i),  α = M⁻¹ b            (NonRigidIcpProposal.scala:152)
//   propose:   c_new = (G + σ²I)⁻¹ G (α + D⁻¹ V √S z),  D M⁻¹ D = V S Vᵀ, G = QᵀQ, σ² = 1e-5      (:53-68)
//   transition: log T = −½ γᵀMγ − (r/2) ln 2π,  (G + σ²M) γ = G (c̃ − α)                           (:71-85)
//     (equal to the reference's whitened-coefficient form for ANY square root of D M⁻¹ D; derivation in DESIGN.md)
// These kernels are latency-bound r×r work (r = 51…201): one workgroup per matrix, data in LDS when it fits.
#include <algorithm>
#include <atomic>
#include <cstdio>
#include <cstdlib>

#include <map>
#include <mutex>

#include "icp_kernels.hpp"
#include "icp_dense.hpp"

namespace icp {

#ifdef ICP_EIGEN_TIMING  // tools/eigen_bench only: phase stamps (100 MHz) of the last eigen kernel
__device__ long long g_eigen_stamps[64];
#define EIG_STAMP(i) do { if (threadIdx.x == 0) g_eigen_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define EIG_STAMP(i)
#endif
}  // namespace icp
#include "icp_tridiag.hpp"
namespace icp {

namespace {

constexpr int kBlock = 256;
constexpr int kEigenMaxSweeps = 40;
#ifndef ICP_LOOSE_TAU
#define ICP_LOOSE_TAU 4e-6
#endif
constexpr double kLooseTau = ICP_LOOSE_TAU;  // loose stopping test of the Jacobi kernels (see k_posterior_eigen_rr)

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- correspondences (one thread per correspondence)

// (init: the memo entry's copy of the state's coefficients and its cleared status words, by the launch's first workgroup — as
// runtime copy / fill operations they were two more dependent submissions of ≈ 5 µs each on every posterior's path)
__device__ __forceinline__ void entry_init(const EntryInit& init) {
  if (blockIdx.x != 0 || !init.coeffs_dst) return;
  for (int i = threadIdx.x; i < init.r; i += blockDim.x) init.coeffs_dst[i] = init.coeffs_src[i];
  if (threadIdx.x < 3) init.status[threadIdx.x] = 0;
}

__global__ void __launch_bounds__(kBlock) k_correspond_model(CorrTask c, const double* __restrict__ cp, EntryInit init) {
  entry_init(init);
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < c.K) correspond_model_one(c, k, ld3(cp + 3 * k));
}

__global__ void __launch_bounds__(kBlock) k_correspond_target(CorrTask c, const int* __restrict__ nn_id, EntryInit init) {
  entry_init(init);
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < c.K) correspond_target_one(c, k, nn_id[k]);
}

__global__ void __launch_bounds__(64) k_regression_mfma(int K, int kchunk, int r, const double* __restrict__ Q, CorrBuffers cb,
                                                         double wt, double kappa, double* __restrict__ Mpart) {
  // (blockIdx.x = split, .y = tile: with 64 splits every tile of a split runs on the XCD "split mod 8" — workgroups go to the XCDs
  // round robin by linear index — and the split's gathered basis rows are fetched into ONE L2; see step_regression_body)
  regression_tile(blockIdx.y, blockIdx.x, K, kchunk, r, Q, cb, wt, kappa, Mpart);
}

constexpr int kFactorMax = 2 * kWideMaxChains;  // posteriors per launch: both ICP directions of one or two states — or one per chain of a wide step
struct FactorArgs {
  const double* Mpart[kFactorMax];
  int splits[kFactorMax];
  double* M[kFactorMax];
  double* alpha[kFactorMax];
  int* status[kFactorMax];
  double* scratch[kFactorMax];  // (r+1)·r doubles, used only when the matrix does not fit in LDS
  // (optional, the Cholesky-root sampler at ranks above 64: icp_proposal_set_sampler) the factor itself: L row-major r × r with a zero
  // upper triangle, and 1/diag(L) — what k_posterior_root writes at ranks <= 64
  double* Lout[kFactorMax];
  double* Sout[kFactorMax];
};

// the factor out of the root-free form W (w_ij = l_ij·d_j, w_jj = d_j; row stride ld): L_ij = w_ij / sqrt(d_j), L_jj = sqrt(d_j)
__device__ __forceinline__ void emit_factor_rootfree(int r, const double* W, int ld, double* __restrict__ Lout, double* __restrict__ Sout) {
  for (int e = threadIdx.x; e < r * r; e += blockDim.x) {
    const int i = e / r, j = e - i * r;
    double v = 0.0;
    if (j <= i) {
      const double d = W[(size_t)j * ld + j], ri = fast_rsqrt(d);
      v = j == i ? d * ri : W[(size_t)i * ld + j] * ri;
    }
    Lout[e] = v;
  }
  if (Sout)
    for (int j = threadIdx.x; j < r; j += blockDim.x) Sout[j] = fast_rsqrt(W[(size_t)j * ld + j]);
}

constexpr int kFactorThreads = 256;

__global__ void __launch_bounds__(kFactorThreads) k_posterior_factor_generic(int r, FactorArgs fa, int use_lds) {
  __shared__ double s_dinv[512], s_v[512];
  const int tid = threadIdx.x, nt = blockDim.x, n = r + 1, p = blockIdx.x;
  const double* Mpart = fa.Mpart[p];
  const int S = fa.splits[p];
  const int ld = use_lds ? (r | 1) : r;   // odd leading dimension in LDS: conflict-free column walks
  double* W = use_lds ? s_dyn : fa.scratch[p];  // rows 0..r-1 = M, row r = bᵀ
  double* M = fa.M[p];
  for (int e = tid; e < n * r; e += nt) {
    const int i = e / r, j = e - i * r;   // i == r: the appended row bᵀ = Maug[r][0..r-1]
    double m = 0.0;
    const int hi = i < r ? max(i, j) : r, lo = i < r ? min(i, j) : j;  // (only the lower triangle of the partials is computed)
    for (int s = 0; s < S; ++s) m += Mpart[(size_t)s * n * n + (size_t)hi * n + lo];
    if (i < r) {
      m += i == j ? 1.0 : 0.0;
      M[e] = m;
    }
    W[(size_t)i * ld + j] = m;
  }
  __syncthreads();
  const bool ok = block_cholesky_rootfree(W, r, ld, 1, 4);
  if (tid == 0) fa.status[p][0] = ok ? 0 : 1;
  if (!ok) return;
  if (fa.Lout[p]) emit_factor_rootfree(r, W, ld, fa.Lout[p], fa.Sout[p]);
  // y = L⁻¹ b sits (unscaled) in row r: y_j = W[r][j]·dinv_j
  for (int j = tid; j < r; j += nt) {
    const double d = fast_rsqrt(W[(size_t)j * ld + j]);
    s_dinv[j] = d;
    s_v[j] = W[(size_t)r * ld + j] * d;
  }
  __syncthreads();
  // back substitution Lᵀ α = y with L[j][i] = W[j][i]·dinv_i:  α_j = (y_j − Σ_{k>j} L[k][j] α_k)·dinv_j
  if (r <= 64) {
    if (tid < 64) {  // one wave, registers + readlane: no barriers on the sequential chain
      const int i = tid;
      double v = i < r ? s_v[i] : 0.0;
      const double di = i < r ? s_dinv[i] : 0.0;
      for (int j = r - 1; j >= 0; --j) {
        const double xj = __shfl(v, j, 64) * s_dinv[j];
        if (i == j) v = xj;
        else if (i < j) v = fma(-(W[(size_t)j * ld + i] * di), xj, v);
      }
      if (i < r) fa.alpha[p][i] = v;
    }
  } else {
    for (int j = r - 1; j >= 0; --j) {
      if (tid == 0) s_v[j] = s_v[j] * s_dinv[j];
      __syncthreads();
      const double xj = s_v[j];
      for (int i = tid; i < j; i += nt) s_v[i] = fma(-(W[(size_t)j * ld + i] * s_dinv[i]), xj, s_v[i]);
      __syncthreads();
    }
    for (int i = tid; i < r; i += nt) fa.alpha[p][i] = s_v[i];
  }
}

template <int TPT, int NT>
__global__ void __launch_bounds__(NT) k_posterior_factor_reg(int r, FactorArgs fa) {
  const int p = blockIdx.x;
  const bool ok = factor_reg_body<TPT, NT>(r, fa.Mpart[p], fa.splits[p], fa.M[p], fa.alpha[p], fa.status[p]);
  if (ok && fa.Lout[p]) emit_factor_rootfree(r, s_dyn, r | 1, fa.Lout[p], fa.Sout[p]);  // (W of factor_reg_body: s_dyn, row stride r | 1)
}

// ---------------------------------------------------------------- K5b for ranks whose factor does not fit one CU's LDS (128..256)
// Right-looking blocked Cholesky of [M; bᵀ] in ONE launch, one workgroup per posterior: the matrix sits in `scratch` (global,
// L2-resident: 320 KB at rank 200), block columns of 64 go through LDS —
//   1 the 64×64 diagonal block is factored in LDS (two barriers per column);
//   2 every row below it (the appended row bᵀ included: forward substitution for free) is solved against that factor, one
//     thread per row, and kept in LDS as the panel P;
//   3 the trailing lower triangle takes −P·Pᵀ (operands from LDS, the matrix itself read and written in place).
// Then the blocked back substitution Lᵀα = y.  ≈ 0.15 ms at rank 200 (the generic kernel further up, every entry behind L2:
// 3.3 ms).
// Σ of the split-K partials of up to two posteriors into their first partial, on many CUs, in split order (the order of the
// factor kernels' own loops)
struct PartialSumArgs { int n; int nn; double* Mpart[kFactorMax]; int splits[kFactorMax]; };
__global__ void __launch_bounds__(256) k_sum_partials(PartialSumArgs a) {
  const int which = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (which >= a.n || e >= a.nn) return;
  double* P = a.Mpart[which];
  const int S = a.splits[which];
  double acc = 0.0;
  for (int s0 = 0; s0 < S; s0 += 8) {  // eight splits in flight
    double q[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) q[u] = P[(size_t)min(s0 + u, S - 1) * a.nn + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) if (s0 + u < S) acc += q[u];
  }
  P[e] = acc;
}

constexpr int kCholNB = 64;
constexpr int kCholMaxRank = 256;

// Blocked back substitution Lᵀ α = y on the SCALED factor in global memory (row-major, ld = r; rows 0..r-1: L with its diagonal,
// row r: y = L⁻¹b), block by block from the end, then α, the status and — Cholesky-root sampler — L and 1/diag(L) handed out.
// A block's triangle goes into LDS first (coalesced), then ONE wave runs its 64 steps out of registers: lane i carries α_i and
// 1/l_ii, a step is two readlane pairs, a multiply and a multiply-add, the rows of L come from LDS four steps ahead (the same chain
// as factor_reg_body's; a __shfl per step — an LDS round trip — made 5 µs of a block, loads from L2 four ahead 8 µs).  The
// unknowns above then lose the block's contribution with every thread at work: four groups of 256 threads take 16 of its
// columns each.
// D: [kCholNB][kCholNB + 1] doubles of LDS; s_y: 512 doubles; s_part: 4 × 256 doubles.
__device__ __forceinline__ void factor_backsolve_emit(int r, const double* __restrict__ W, double* D, double* s_y, double* s_part,
                                                      const FactorArgs& fa, int p) {
  const int tid = threadIdx.x, nt = blockDim.x;
  constexpr int ldd = kCholNB + 1;
  for (int j = tid; j < r; j += nt) s_y[j] = W[(size_t)r * r + j];
  const int last_kb = ((r - 1) / kCholNB) * kCholNB;
  for (int kb = last_kb; kb >= 0; kb -= kCholNB) {
    const int nbk = min(kCholNB, r - kb);
    for (int e = tid; e < kCholNB * kCholNB; e += nt) {
      const int i = e >> 6, j = e & 63;
      if (j <= i && i < nbk) D[i * ldd + j] = W[(size_t)(kb + i) * r + kb + j];
    }
    __syncthreads();
    if (tid < 64) {
      const int i = tid, ic = i < nbk ? i : nbk - 1;  // lanes past the block mirror its last lane (their result is discarded)
      double x = s_y[kb + ic];
      const double di = fast_rcp(D[ic * ldd + ic]);
      constexpr int kA = 4;
      double lq[kA];
#pragma unroll
      for (int a = 0; a < kA; ++a) lq[a] = D[max(nbk - 1 - a, ic) * ldd + ic];  // (row >= column: the triangle that was staged)
      for (int j0 = nbk - 1; j0 >= 0; j0 -= kA) {
#pragma unroll
        for (int a = 0; a < kA; ++a) {
          const int j = j0 - a;                         // (steps with j < 0, the padding of the last group, change nothing)
          const double lij = lq[a];
          lq[a] = D[max(j - kA, ic) * ldd + ic];
          const int js = j & 63;
          const double xr = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(x), js), __builtin_amdgcn_readlane(__double2loint(x), js));
          const double dj = __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(di), js), __builtin_amdgcn_readlane(__double2loint(di), js));
          const double xj = xr * dj;
          const double upd = fma(-lij, xj, x);
          x = j < 0 ? x : (i == j ? xj : (i < j ? upd : x));
        }
      }
      if (i < nbk) s_y[kb + i] = x;
    }
    __syncthreads();
    if (kb > 0) {  // the unknowns further up lose this block's contribution (kb <= 192 of them: thread = (column group, unknown))
      const int g = tid >> 8, i = tid & 255;
      if (i < kb) {
        double acc = 0.0;
#pragma unroll
        for (int jj = 0; jj < 16; ++jj) {
          const int j = 16 * g + jj;
          acc = j < nbk ? fma(W[(size_t)(kb + j) * r + i], s_y[kb + j], acc) : acc;
        }
        s_part[g * 256 + i] = acc;
      }
      __syncthreads();
      if (tid < kb) s_y[tid] -= (s_part[tid] + s_part[256 + tid]) + (s_part[512 + tid] + s_part[768 + tid]);
      __syncthreads();
    }
  }
  for (int j = tid; j < r; j += nt) fa.alpha[p][j] = s_y[j];
  if (tid == 0) fa.status[p][0] = 0;
  if (fa.Lout[p]) {  // (W holds the scaled factor here: diagonal sqrt(d), columns divided by it)
    double* __restrict__ Lo = fa.Lout[p];
    for (int e = tid; e < r * r; e += nt) {
      const int i = e / r, j = e - i * r;
      Lo[e] = j <= i ? W[e] : 0.0;
    }
    if (fa.Sout[p])
      for (int j = tid; j < r; j += nt) fa.Sout[p][j] = fast_rcp(W[(size_t)j * r + j]);
  }
}

__global__ void __launch_bounds__(1024) k_posterior_factor_blocked(int r, FactorArgs fa) {
  __shared__ double s_y[512], s_dinv[kCholNB], s_part[1024];
  __shared__ int s_fail;
  const int tid = threadIdx.x, nt = blockDim.x, n = r + 1, p = blockIdx.x;
  const double* __restrict__ Mpart = fa.Mpart[p];
  const int S = fa.splits[p];
  double* __restrict__ W = fa.scratch[p];  // rows 0..r-1: lower triangle of M -> L; row r: bᵀ -> y = L⁻¹b  (row-major, ld = r)
  double* __restrict__ M = fa.M[p];
  double* D = s_dyn;                       // [kCholNB][kCholNB + 1] diagonal block
  double* P = s_dyn + kCholNB * (kCholNB + 1);  // [rows below][kCholNB (+1: rows one bank apart)] panel
  constexpr int ldd = kCholNB + 1, ldp = kCholNB + 1;
#ifdef ICP_EIGEN_TIMING
  long long t_acc[4] = {0, 0, 0, 0}, t_last = __builtin_amdgcn_s_memrealtime();
#define CHOL_T(i) do { const long long t_now = __builtin_amdgcn_s_memrealtime(); t_acc[i] += t_now - t_last; t_last = t_now; } while (0)
#else
#define CHOL_T(i)
#endif
  if (tid == 0) s_fail = 0;
  // ---- M = I + Σ partials (lower triangle of the partials, split order), bᵀ = row r
  for (int e = tid; e < n * r; e += nt) {
    const int i = e / r, j = e - i * r;
    if (j > i) continue;
    double m = 0.0;
    const double* src = Mpart + (size_t)i * n + j;
    for (int s0 = 0; s0 < S; s0 += 8) {  // eight splits in flight (a loop of dependent loads took 0.4 ms here); summed in split order
      double q[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) q[u] = src[(size_t)min(s0 + u, S - 1) * n * n];
#pragma unroll
      for (int u = 0; u < 8; ++u) if (s0 + u < S) m += q[u];
    }
    if (i < r) {
      if (i == j) m += 1.0;
      M[(size_t)i * r + j] = m;
      M[(size_t)j * r + i] = m;
    }
    W[(size_t)i * r + j] = m;
  }
  __syncthreads();
  CHOL_T(0);
  for (int kb = 0; kb < r; kb += kCholNB) {
    const int nbk = min(kCholNB, r - kb), below0 = kb + nbk, n_below = n - below0;
    // 1: diagonal block
    for (int e = tid; e < nbk * nbk; e += nt) {
      const int i = e / nbk, j = e - i * nbk;
      if (j <= i) D[i * ldd + j] = W[(size_t)(kb + i) * r + kb + j];
    }
    __syncthreads();
    // root-free right-looking elimination (one reciprocal and ONE barrier per column), then the columns are scaled once:
    // L[i][j] = U[i][j]·rsqrt(U[j][j])
    if (!block_cholesky_rootfree(D, nbk, ldd, 0, 5)) { if (tid == 0) s_fail = 1; }
    __syncthreads();
    if (!s_fail) {
      const int ty = tid >> 5, tx = tid & 31;
      for (int j = tx; j < nbk; j += 32) {
        const double dinv = fast_rsqrt(D[j * ldd + j]);
        for (int i = j + 1 + ty; i < nbk; i += 32) D[i * ldd + j] *= dinv;
      }
      __syncthreads();
      for (int j = tid; j < nbk; j += nt) { const double u = D[j * ldd + j], ri = fast_rsqrt(u); D[j * ldd + j] = u * ri; s_dinv[j] = ri; }
      __syncthreads();
    }
    if (s_fail) break;
    for (int e = tid; e < nbk * nbk; e += nt) {
      const int i = e / nbk, j = e - i * nbk;
      if (j <= i) W[(size_t)(kb + i) * r + kb + j] = D[i * ldd + j];
    }
    CHOL_T(1);
    // 2: panel — row `below0 + t` solved against the block's factor: x_j = (a_j − Σ_{k<j} x_k·l_jk) / l_jj
    // (the rows' segments come into LDS first, all loads in flight, and go back when solved: a load per step of the
    // substitution — which the stores of the step before it pin in place — cost one trip to L2 per column)
    for (int e = tid; e < n_below * nbk; e += nt) {
      const int t = e / nbk, j = e - t * nbk;
      P[(size_t)t * ldp + j] = W[(size_t)(below0 + t) * r + kb + j];
    }
    __syncthreads();
    // eight lanes per row, entry k of the row with lane k mod 8: a step's dot product is at most eight multiply-adds per lane
    // and three DPP exchanges inside the group; the lane that owns x_j is the only one that needs it (128 rows per pass)
    for (int row0 = 0; row0 < n_below; row0 += 128) {
      const int t = row0 + (tid >> 3), u = tid & 7;
      const bool act = t < n_below;
      double* xrow = P + (size_t)(act ? t : 0) * ldp;
      double x[8];
#pragma unroll
      for (int q = 0; q < 8; ++q) x[q] = act && u + 8 * q < nbk ? xrow[u + 8 * q] : 0.0;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) {
        if (8 * jj < nbk) {
#pragma unroll
          for (int uu = 0; uu < 8; ++uu) {
            const int j = 8 * jj + uu;
            if (j < nbk) {
              const double* lj = D + j * ldd + u;
              double part = 0.0;
#pragma unroll
              for (int q = 0; q < jj; ++q) part = fma(x[q], lj[8 * q], part);
              part = u < uu ? fma(x[jj], lj[8 * jj], part) : part;
              part += tri::dpp_f64<0xB1>(part);
              part += tri::dpp_f64<0x4E>(part);
              part += tri::dpp_f64<0x141>(part);
              const double xj = (x[jj] - part) * s_dinv[j];
              x[jj] = u == uu ? xj : x[jj];
            }
          }
        }
      }
      if (act) {
#pragma unroll
        for (int q = 0; q < 8; ++q) if (u + 8 * q < nbk) xrow[u + 8 * q] = x[q];
      }
    }
    __syncthreads();
    for (int e = tid; e < n_below * nbk; e += nt) {
      const int t = e / nbk, j = e - t * nbk;
      W[(size_t)(below0 + t) * r + kb + j] = P[(size_t)t * ldp + j];
    }
    CHOL_T(2);
    // 3: trailing update of the rows below (lower triangle; the appended row r has every column < r)
    // (4×4 outputs per thread: eight LDS reads per sixteen multiply-adds instead of two per one)
    {
      const int nt4 = (n_below + 3) >> 2, ntiles = nt4 * (nt4 + 1) / 2;
      for (int tile = tid; tile < ntiles; tile += nt) {
        int bi = (int)((sqrtf(8.0f * (float)tile + 1.0f) - 1.0f) * 0.5f);
        while (bi * (bi + 1) / 2 > tile) --bi;
        while ((bi + 1) * (bi + 2) / 2 <= tile) ++bi;
        const int bj = tile - bi * (bi + 1) / 2;
        const double* xi[4];
        const double* xj[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          xi[a] = P + (size_t)min(4 * bi + a, n_below - 1) * ldp;
          xj[a] = P + (size_t)min(4 * bj + a, n_below - 1) * ldp;
        }
        double acc[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
        for (int k = 0; k < nbk; ++k) {
          double va[4], vb[4];
#pragma unroll
          for (int a = 0; a < 4; ++a) { va[a] = xi[a][k]; vb[a] = xj[a][k]; }
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[a][b] = fma(va[a], vb[b], acc[a][b]);
        }
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int ti = 4 * bi + a, tj = 4 * bj + b;
            const int gi = below0 + ti, gj = below0 + tj;
            if (ti < n_below && tj <= ti && gj < r) W[(size_t)gi * r + gj] -= acc[a][b];
          }
      }
    }
    __syncthreads();
    CHOL_T(3);
  }
#ifdef ICP_EIGEN_TIMING
  if (tid == 0 && blockIdx.x == 0) { for (int i = 0; i < 4; ++i) g_eigen_stamps[24 + i] = t_acc[i]; g_eigen_stamps[28] = t_last; }
#endif
  if (s_fail) {
    if (tid == 0) fa.status[p][0] = 1;
    return;
  }
  factor_backsolve_emit(r, W, D, s_y, s_part, fa, p);
#ifdef ICP_EIGEN_TIMING
  if (tid == 0 && blockIdx.x == 0) g_eigen_stamps[29] = __builtin_amdgcn_s_memrealtime();
#endif
}

// ---------------------------------------------------------------- K5c ranks 117..~250: the whole factorisation in REGISTERS
// The register-tiled right-looking elimination of factor_reg_body (2 × 4 tiles, one barrier per column, the pivot column passed
// through a double buffer in LDS) does not need the factor in LDS at all — only the finished columns went there, for the back
// substitution.  Here they go to global scratch (fire-and-forget stores of the SCALED column: l_ij = u_ij·rsqrt(u_jj), written by
// the threads 0..n/2 from the pivot-column buffer every thread reads anyway), TPT = 3 or 4 tiles per thread cover ranks up to 250
// (48-64 registers of matrix per thread), and the blocked back substitution above reads the factor back from L2.
// Tiles are dealt COLUMN-major: a thread's slots die one after the other as the elimination passes their columns, the whole
// workgroup skips a dead slot (its loads from the column buffer and its eight multiply-adds) — 57 % of the work at rank 200.
// 200 columns × ≈ 0.3 µs instead of the blocked kernel's diagonal blocks + panels + trailing updates (187 µs).
static __host__ __device__ inline int factor_tile_rows(int r) { return (r + 2) >> 1; }  // over the r+1 rows (M and bᵀ)
constexpr int kFactorAsmGroups = 4;  // workgroups beside the factorising one that write M

template <int TPT>
__global__ void __launch_bounds__(1024) k_posterior_factor_tiles(int r, FactorArgs fa) {
  __shared__ __attribute__((aligned(16))) double s_col[2][520];
  __shared__ double s_y[512], s_part[1024];
  constexpr int NT = 1024;
  const int tid = threadIdx.x, n = r + 1, p = blockIdx.x;
  const double* __restrict__ Mpart = fa.Mpart[p];   // one (summed) partial, (r+1) × (r+1)
  double* __restrict__ W = fa.scratch[p];            // (r+1) × r: the scaled factor, row r = y
  double* __restrict__ M = fa.M[p];
  if (blockIdx.y > 0) {
    // M = I + the summed partial, both triangles, for the kernels that follow (tails, decomposition): by workgroups of their own,
    // coalesced — written from the tiles (a row per lane: one memory transaction per element) it took 17 of the kernel's first 22 µs
    for (int e = (blockIdx.y - 1) * NT + tid; e < r * r; e += (gridDim.y - 1) * NT) {
      const int i = e / r, j = e - i * r;
      M[e] = Mpart[(size_t)max(i, j) * n + min(i, j)] + (i == j ? 1.0 : 0.0);
    }
    return;
  }
  // (the evaluator's searches fill the chip while this workgroup runs: 137-181 µs beside them, 113 alone — its waves issue first)
  __builtin_amdgcn_s_setprio(3);
  FAC_STAMP(16);
  const int tr = factor_tile_rows(r), n_tiles = factor_tile_count(r);
  double v[TPT][2][4];
  int R0[TPT], C0[TPT];
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
    const int e = tid + NT * t;
    int tc = 0, base = 0;  // tile column tc holds the tile rows 2·tc … tr−1
    if (e < n_tiles) {
      while (base + (tr - 2 * tc) <= e) { base += tr - 2 * tc; ++tc; }
    }
    R0[t] = e < n_tiles ? 2 * (2 * tc + (e - base)) : -2;  // -2: slot unused
    C0[t] = e < n_tiles ? 4 * tc : -8;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {  // every load of the thread in flight before the first store below
        const int i = R0[t] + a, k = C0[t] + c;
        const bool live = R0[t] >= 0 && i < n && k < r && k <= i;
        v[t][a][c] = live ? Mpart[(size_t)i * n + k] : 0.0;
      }
  }
#pragma unroll
  for (int t = 0; t < TPT; ++t) {
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int i = R0[t] + a, k = C0[t] + c;
        const bool live = R0[t] >= 0 && i < n && k < r && k <= i;
        if (live) {
          if (i < r && i == k) v[t][a][c] += 1.0;
          if (k == 0) s_col[0][i] = v[t][a][c];
        }
      }
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
      if (!(ajj > 0.0)) { ok = false; break; }  // same value in every thread: uniform exit
      const double inv = fast_rcp(ajj);
      const int kp = j + 1;                      // the column that becomes final in this step
      if (2 * tid < n) {                         // column j of the scaled factor, rows j … r
        const double rs = fast_rsqrt(ajj);
        const dense2 cj = *(const dense2*)&cur[2 * tid];
        const int i0 = 2 * tid;
        if (i0 >= j) W[(size_t)i0 * r + j] = cj.x * rs;
        if (i0 + 1 >= j && i0 + 1 < n) W[(size_t)(i0 + 1) * r + j] = cj.y * rs;
      }
#pragma unroll
      for (int t = 0; t < TPT; ++t) {
        if (C0[t] + 3 < j4) continue;            // every column of this tile is final (slots die workgroup-wide, a few columns apart)
        const dense2 u = *(const dense2*)&cur[R0[t]];
        const dense2 k01 = *(const dense2*)&cur[C0[t]];
        const dense2 k23 = *(const dense2*)&cur[C0[t] + 2];
        const double m0 = -(u.x * inv), m1 = -(u.y * inv);
        v[t][0][0] = fma(m0, k01.x, v[t][0][0]); v[t][0][1] = fma(m0, k01.y, v[t][0][1]);
        v[t][0][2] = fma(m0, k23.x, v[t][0][2]); v[t][0][3] = fma(m0, k23.y, v[t][0][3]);
        v[t][1][0] = fma(m1, k01.x, v[t][1][0]); v[t][1][1] = fma(m1, k01.y, v[t][1][1]);
        v[t][1][2] = fma(m1, k23.x, v[t][1][2]); v[t][1][3] = fma(m1, k23.y, v[t][1][3]);
        if (kp < r && C0[t] == (kp & ~3))        // this tile holds column kp at tile column (cc+1)&3
          *(dense2*)&nxt[R0[t]] = dense2{v[t][0][(cc + 1) & 3], v[t][1][(cc + 1) & 3]};
      }
      __syncthreads();
    }
  }
  if (!ok) {
    if (tid == 0) fa.status[p][0] = 1;
    return;
  }
  FAC_STAMP(18);
  FAC_STAMP(19);
  factor_backsolve_emit(r, W, s_dyn, s_y, s_part, fa, p);
  FAC_STAMP(20);
}

constexpr int kTailMax = 2 * kWideMaxChains;  // tails per launch (a wide step: forward and backward of every chain)
struct TailArgs {
  int n;
  const int* relay_in[kTailMax];
  int* relay_out[kTailMax];
  const double* alpha[kTailMax];
  const double* M[kTailMax];
  const double* c_from[kTailMax];
  const double* c_to[kTailMax];
  double step[kTailMax];
  double* out[kTailMax];
  int* status[kTailMax];
};

template <int NT>
__global__ void __launch_bounds__(NT) k_transition_tails(int r, TailArgs ta, const double* __restrict__ Ginv, double sigma2,
                                                          int n_lds, int tpr_log2) {
  const int t = blockIdx.x;
  if (ta.relay_in[t] && threadIdx.x < 3) ta.relay_out[t][threadIdx.x] = ta.relay_in[t][threadIdx.x];
  if (NT == 1024) __builtin_amdgcn_s_setprio(3);  // (ranks above 134: beside the evaluator's searches, like the factorisation)
  tail_body(r, ta.alpha[t], ta.M[t], ta.c_from[t], ta.c_to[t], ta.step[t], ta.out[t], ta.status[t], Ginv, sigma2, n_lds, tpr_log2);
}

__global__ void __launch_bounds__(1024) k_transition_tail_direct(int r, const double* __restrict__ alpha, const double* __restrict__ M,
                                                                  const double* __restrict__ G, double sigma2,
                                                                  const double* __restrict__ c_from, const double* __restrict__ c_to,
                                                                  double step, double* __restrict__ work, double* __restrict__ out,
                                                                  int* __restrict__ status, int use_lds) {
  __shared__ double s_d[512], s_dinv[512], s_red[16];
  const int tid = threadIdx.x, nt = blockDim.x;
  const int ld = use_lds ? (r | 1) : r;
  double* W = use_lds ? s_dyn : work;  // rows 0..r-1 = G + σ²M, row r = (G d)ᵀ
  for (int j = tid; j < r; j += nt) s_d[j] = (c_from[j] + (c_to[j] - c_from[j]) / step) - alpha[j];
  __syncthreads();
  for (int e = tid; e < r * r; e += nt) {
    const int i = e / r, j = e - i * r;
    W[(size_t)i * ld + j] = fma(sigma2, M[e], G[e]);
  }
  for (int i = tid; i < r; i += nt) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(G[(size_t)j * r + i], s_d[j], s);
    W[(size_t)r * ld + i] = s;
  }
  __syncthreads();
  const bool ok = block_cholesky_rootfree(W, r, ld, 1, 5);
  if (!ok) { if (tid == 0) status[0] = 1; return; }
  for (int j = tid; j < r; j += nt) {
    const double d = fast_rsqrt(W[(size_t)j * ld + j]);
    s_dinv[j] = d;
    s_d[j] = W[(size_t)r * ld + j] * d;
  }
  __syncthreads();
  for (int j = r - 1; j >= 0; --j) {
    if (tid == 0) s_d[j] = s_d[j] * s_dinv[j];
    __syncthreads();
    const double xj = s_d[j];
    for (int i = tid; i < j; i += nt) s_d[i] = fma(-(W[(size_t)j * ld + i] * s_dinv[i]), xj, s_d[i]);
    __syncthreads();
  }
  double part = 0.0;
  for (int i = tid; i < r; i += nt) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(M[(size_t)j * r + i], s_d[j], s);
    part = fma(s_d[i], s, part);
  }
  const double q = block_sum(part, s_red);
  if (tid == 0) { out[0] = -0.5 * q - 0.5 * (double)r * 1.8378770664093453; status[0] = 0; }
}

// ---------------------------------------------------------------- posterior KL basis: parallel two-sided Jacobi
// Eigen-decomposition of N = D⁻¹ M D⁻¹ (same eigenvectors as D M⁻¹ D, reciprocal eigenvalues) in one workgroup of
// 1024 threads, matrices in LDS (odd leading dimension).  Round-robin ordering: each round rotates n/2 disjoint
// index pairs concurrently:
//   phase 1: one thread per pair computes (c, s) from three matrix entries (reciprocal/rsqrt seeds + Newton: the
//            f64 division/sqrt expansions would dominate the round), barrier;
//   phase 2: every 2×2 block (rows of pair P1 × columns of pair P2) is transformed as R1ᵀ·B·R2 by ONE thread, so each
//            matrix element is read and written once per round; other threads rotate the column pairs of V; barrier.
// Warm start: if `Vwarm` is given, the iteration starts from Vwarmᵀ N Vwarm (nearly diagonal when Vwarm diagonalised a
// nearby posterior) with V = Vwarm, which cuts the number of sweeps roughly in half; the result is the same
// eigen-decomposition (to rounding) either way.

__global__ void __launch_bounds__(1024) k_posterior_eigen(int r, const double* __restrict__ M, const double* __restrict__ sqrt_lambda,
                                                           const double* __restrict__ Vwarm, double* __restrict__ Vout,
                                                           double* __restrict__ Vtout, double* __restrict__ Sout,
                                                           double* __restrict__ work, int* __restrict__ status, int a_in_lds, int v_in_lds,
                                                           const int* __restrict__ gate = nullptr) {
  __shared__ double s_red[16], s_mu[512], s_sgn[512], s_c[256], s_s[256];
  __shared__ int s_rank[512];
  __shared__ short s_p[256], s_q[256];
  // (as the tridiagonal route's fall-back above rank 200, where the in-place kernel's triangle no longer fits a CU's LDS: runs only if
  // the multisection could not separate the spectrum — status 2)
  if (gate && gate[0] != 2) return;
  const int tid = threadIdx.x, nt = blockDim.x;
  const int lda = a_in_lds ? (r | 1) : r, ldv = v_in_lds ? (r | 1) : r;
  double* A = a_in_lds ? s_dyn : work;
  double* V = v_in_lds ? (s_dyn + (a_in_lds ? (size_t)r * lda : 0)) : Vout;
  for (int e = tid; e < r * r; e += nt) {
    const int i = e / r, j = e - i * r;
    const double mij = 0.5 * (M[e] + M[(size_t)j * r + i]);
    A[(size_t)i * lda + j] = mij / (sqrt_lambda[i] * sqrt_lambda[j]);
    V[(size_t)i * ldv + j] = Vwarm ? Vwarm[e] : (i == j ? 1.0 : 0.0);
  }
  __syncthreads();
  if (Vwarm) {
    // A <- Vᵀ A V in two passes through `work` (T = A V, then A = Vᵀ T); `work` is free when A lives in LDS,
    // otherwise the warm start is skipped by the launcher.
    for (int e = tid; e < r * r; e += nt) {
      const int i = e / r, j = e - i * r;
      double s = 0.0;
      for (int k = 0; k < r; ++k) s = fma(A[(size_t)i * lda + k], V[(size_t)k * ldv + j], s);
      work[e] = s;
    }
    __syncthreads();
    for (int e = tid; e < r * r; e += nt) {
      const int i = e / r, j = e - i * r;
      double s = 0.0;
      for (int k = 0; k < r; ++k) s = fma(V[(size_t)k * ldv + i], work[(size_t)k * r + j], s);
      A[(size_t)i * lda + j] = s;
    }
    __syncthreads();
    for (int e = tid; e < r * r; e += nt) {  // symmetrise against rounding
      const int i = e / r, j = e - i * r;
      if (i < j) {
        const double v = 0.5 * (A[(size_t)i * lda + j] + A[(size_t)j * lda + i]);
        A[(size_t)i * lda + j] = v;
        A[(size_t)j * lda + i] = v;
      }
    }
    __syncthreads();
  }
  const int n2 = (r + 1) & ~1, half = n2 >> 1, mm = n2 - 1;
  // work items of a round: n_blocks 2×2 blocks of A (upper triangle of the pair×pair grid, mirrored) + r·half column
  // pairs of V.  Item -> thread mapping is fixed, so everything but the pair's current (p,q) is precomputed.
  const int n_blocks = half * (half + 1) / 2, n_items = n_blocks + r * half;
  constexpr int kItems = 2;  // items with precomputed descriptors; more (large ranks) go through the generic loop
  int it_a[kItems], it_b[kItems];  // block item: (P1 <= P2); V item: (k | 0x40000000, P)
#pragma unroll
  for (int m = 0; m < kItems; ++m) {
    const int w = tid + nt * m;
    it_a[m] = -1; it_b[m] = 0;
    if (w < n_blocks) {  // unrank the upper triangle row-major: P1 <= P2
      int P1 = 0, base = 0;
      while (base + (half - P1) <= w) { base += half - P1; ++P1; }
      it_a[m] = P1; it_b[m] = P1 + (w - base);
    } else if (w < n_items) {
      const int vi = w - n_blocks;
      it_a[m] = (vi / half) | 0x40000000; it_b[m] = vi % half;
    }
  }
  // round-robin state of the pair this thread computes in phase 1 (slot = tid): incremental, no modulo per round
  int ra = 0, rb = 0;
  if (tid < half) {
    if (tid == 0) { ra = mm; rb = 0; }
    else { ra = tid % mm; rb = (mm - tid) % mm; }
  }
  auto do_block = [&](int P1, int P2) {
    const int p1 = s_p[P1], q1 = s_q[P1], p2 = s_p[P2], q2 = s_q[P2];
    const double c1 = s_c[P1], s1 = s_s[P1], c2 = s_c[P2], s2 = s_s[P2];
    const bool hq1 = q1 < r, hq2 = q2 < r;
    const int opp = p1 * lda + p2, opq = p1 * lda + q2, oqp = q1 * lda + p2, oqq = q1 * lda + q2;
    const double bpp = A[opp];
    const double bpq = hq2 ? A[opq] : 0.0;
    const double bqp = hq1 ? A[oqp] : 0.0;
    const double bqq = (hq1 && hq2) ? A[oqq] : 0.0;
    // rows (pair P1): [p; q] <- [c −s; s c][p; q];  columns (pair P2): [p q] <- [p q][c s; −s c]
    const double tpp = c1 * bpp - s1 * bqp, tpq = c1 * bpq - s1 * bqq;
    const double tqp = s1 * bpp + c1 * bqp, tqq = s1 * bpq + c1 * bqq;
    const double npp = c2 * tpp - s2 * tpq, npq = s2 * tpp + c2 * tpq;
    const double nqp = c2 * tqp - s2 * tqq, nqq = s2 * tqp + c2 * tqq;
    A[opp] = npp;
    if (hq2) A[opq] = npq;
    if (hq1) A[oqp] = nqp;
    if (hq1 && hq2) A[oqq] = nqq;
    if (P1 != P2) {  // mirror block (A stays exactly symmetric)
      A[p2 * lda + p1] = npp;
      if (hq2) A[q2 * lda + p1] = npq;
      if (hq1) A[p2 * lda + q1] = nqp;
      if (hq1 && hq2) A[q2 * lda + q1] = nqq;
    }
  };
  auto do_vpair = [&](int k, int P) {
    const int p = s_p[P], q = s_q[P];
    if (q < r) {
      const double c = s_c[P], s = s_s[P];
      const int op = k * ldv + p, oq = k * ldv + q;
      const double vkp = V[op], vkq = V[oq];
      V[op] = c * vkp - s * vkq;
      V[oq] = s * vkp + c * vkq;
    }
  };
  int converged = 0, n_sweeps = 0;
  for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
    for (int rnd = 0; rnd < mm; ++rnd) {
      if (tid < half) {
        const int p = ra < rb ? ra : rb, q = ra < rb ? rb : ra;
        double c = 1.0, s = 0.0;
        if (q < r) {
          const double apq = A[p * lda + q], app = A[p * lda + p], aqq = A[q * lda + q];
          if (fabs(apq) > 1e-300 && apq * apq > 1e-36 * fabs(app * aqq)) {
            // t = sgn(a)·b / (|a| + sqrt(a² + b²)),  a = (aqq − app)/2, b = apq  (smaller root of t² + 2τt − 1 = 0)
            const double a = 0.5 * (aqq - app);
            const double h2 = fma(a, a, apq * apq);
            const double h = h2 * fast_rsqrt(h2);
            const double t = (a >= 0.0 ? apq : -apq) * fast_rcp(fabs(a) + h);
            c = fast_rsqrt(fma(t, t, 1.0));
            s = t * c;
          }
        }
        s_p[tid] = (short)p; s_q[tid] = (short)q; s_c[tid] = c; s_s[tid] = s;
        // next round's pair of this slot (circle method: every player but the fixed one advances by one seat)
        if (tid == 0) { rb = rb + 1 == mm ? 0 : rb + 1; }
        else { ra = ra + 1 == mm ? 0 : ra + 1; rb = rb + 1 == mm ? 0 : rb + 1; }
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kItems; ++m) {
        if (it_a[m] >= 0) {
          if (it_a[m] & 0x40000000) do_vpair(it_a[m] & 0x3FFFFFFF, it_b[m]);
          else do_block(it_a[m], it_b[m]);
        }
      }
      for (int w = tid + nt * kItems; w < n_items; w += nt) {  // large ranks only
        if (w < n_blocks) {
          int P1 = 0, base = 0;
          while (base + (half - P1) <= w) { base += half - P1; ++P1; }
          do_block(P1, P1 + (w - base));
        } else {
          const int vi = w - n_blocks;
          do_vpair(vi / half, vi % half);
        }
      }
      __syncthreads();
    }
    double off = 0.0, dg = 0.0;
    for (int e = tid; e < r * r; e += nt) {
      const int i = e / r, j = e - i * r;
      const double v = A[(size_t)i * lda + j];
      if (i == j) dg = fma(v, v, dg);
      else off = fma(v, v, off);
    }
    off = block_sum(off, s_red);
    dg = block_sum(dg, s_red);
    converged = off <= 1e-26 * dg;
    n_sweeps = sweep + 1;
  }
  if (tid == 0) { status[0] = converged ? 0 : 2; status[-1] = n_sweeps; }
  // eigenvalues of D M⁻¹ D are 1/μ; S descending = μ ascending (ties: lower original index first)
  for (int i = tid; i < r; i += nt) s_mu[i] = A[(size_t)i * lda + i];
  __syncthreads();
  for (int i = tid; i < r; i += nt) {
    int rank = 0;
    const double mi = s_mu[i];
    for (int j = 0; j < r; ++j) rank += (s_mu[j] < mi) || (s_mu[j] == mi && j < i);
    s_rank[i] = rank;
    int best = 0;  // canonical sign: the largest-|.| component of each eigenvector is positive
    double bv = fabs(V[i]);
    for (int k = 1; k < r; ++k) {
      const double a = fabs(V[(size_t)k * ldv + i]);
      if (a > bv) { bv = a; best = k; }
    }
    s_sgn[i] = V[(size_t)best * ldv + i] < 0.0 ? -1.0 : 1.0;
    Sout[rank] = 1.0 / mi;
  }
  __syncthreads();
  if (!v_in_lds) {  // V aliases Vout: permute through `work` (free if A sat in LDS; otherwise A lived there and is dead now)
    for (int e = tid; e < r * r; e += nt) work[e] = V[e];
    __syncthreads();
    V = work;
  }
  for (int e = tid; e < r * r; e += nt) {
    const int k = e / r, i = e - k * r;
    const double v = V[(size_t)k * ldv + i] * s_sgn[i];
    Vout[(size_t)k * r + s_rank[i]] = v;
    Vtout[(size_t)s_rank[i] * r + k] = v;
  }
}

// ---------------------------------------------------------------- posterior KL basis, ranks <= 64: fixed-position Jacobi
// Same method (cyclic two-sided Jacobi, round-robin pairing, warm start) re-laid for the LDS pipe, which bounds the kernel
// above: there every round gathers its pair indices and rotation parameters through dependent LDS reads and touches each
// matrix element with scalar 8-byte accesses.  Here the PAIRING never changes — pair K always sits at positions (2K, 2K+1)
// — and the matrix itself is permuted by the round-robin rotation while it is written back (Brent–Luk style), so
//   * every thread reads and writes the SAME addresses every round (all offsets precomputed in registers);
//   * the two elements of a pair are adjacent: one 16-byte read fetches both;
//   * A and V are double-buffered (read `cur`, write the permuted result to `nxt`): ONE barrier per round;
//   * the rotation of a pair of the NEXT round is computed in the same round by a dedicated thread, from the three
//     transformed entries it needs (evaluated with the expressions the block threads use, so both agree bit for bit).
// A is kept exactly symmetric (block (J,I) is computed as the transpose of block (I,J) by the same arithmetic).  An odd
// rank is padded with a dummy index (zero row/column, diagonal 1e300): its rotations are identities.

__device__ __forceinline__ int rr_dst(int pos, int m) {  // where the content of position `pos` goes after a round
  const int k = pos >> 1;
  if ((pos & 1) == 0) return k == 0 ? 0 : (k == m - 1 ? 2 * (m - 1) + 1 : 2 * (k + 1));
  return k == 0 ? 2 : 2 * (k - 1) + 1;
}
__device__ __forceinline__ int rr_src(int pos, int m) {  // inverse of rr_dst
  const int k = pos >> 1;
  if ((pos & 1) == 0) return k == 0 ? 0 : (k == 1 ? 1 : 2 * (k - 1));
  return k == m - 1 ? 2 * (m - 1) : 2 * (k + 1) + 1;
}

struct Rot { double c, s; };

// Rotation (nearly) annihilating apq, branch free; the dependent chain is two reciprocal square roots and no division:
//   a = aqq − app, b = 2·apq (the angle depends on their ratio only), h ≈ sqrt(a² + b²), u = h + |a|:
//   c = u/sqrt(u² + b²), s = sgn(a)·b/sqrt(u² + b²)          (t = s/c = sgn(a)·b/(|a| + h), the smaller root)
// c² + s² = 1 holds to rounding for ANY h, so h comes from the bare hardware seed (relative error 5e-8): the rotated
// off-diagonal entry is then 5e-8·apq instead of 0, which the next sweep removes — the callers store the computed entry,
// never an assumed zero.
__device__ __forceinline__ Rot jacobi_rotation(double app, double apq, double aqq) {
  const double a = aqq - app, b = apq + apq, b2 = b * b;
  const bool rot = apq * apq > 1e-36 * fabs(app * aqq);  // false for zero / underflowing entries and for the dummy index
  const double h2 = fma(a, a, b2);
  const double h = h2 * __builtin_amdgcn_rsq(h2);
  const double u = h + fabs(a);
  const double w = fma(u, u, b2);
  // 1/sqrt(w): seed (5e-8) and one third-order step, y·(1 + e + 1.5e²) with e = ½ − ½w·y²: error ~e³
  const double y0 = __builtin_amdgcn_rsq(w);
  const double e = fma(-(0.5 * w) * y0, y0, 0.5);
  const double y = fma(y0, e * fma(1.5, e, 1.0), y0);
  Rot R;
  R.c = rot ? u * y : 1.0;            // not rotating (negligible or zero entry, dummy index): NaN/inf above are discarded
  R.s = rot ? (a >= 0.0 ? b : -b) * y : 0.0;
  return R;
}

struct B22 { double a00, a01, a10, a11; };

// R1ᵀ·B·R2 with R = [c s; −s c]
__device__ __forceinline__ B22 rot_block(B22 b, double c1, double s1, double c2, double s2) {
  const double t00 = fma(c1, b.a00, -(s1 * b.a10)), t01 = fma(c1, b.a01, -(s1 * b.a11));
  const double t10 = fma(s1, b.a00, c1 * b.a10), t11 = fma(s1, b.a01, c1 * b.a11);
  return B22{fma(c2, t00, -(s2 * t01)), fma(s2, t00, c2 * t01), fma(c2, t10, -(s2 * t11)), fma(s2, t10, c2 * t11)};
}
typedef double dbl2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ dbl2 lds2(const double* p) { return *(const dbl2*)p; }

// Work split of one round (1024 threads launched; ranks <= 64).  Only the upper triangle of A is stored: element {R, C}
// lives at [min][max], so every element is written once per round and no mirror is kept.
//   wave 3         lane K prepares the rotation of pair K of the next round — the longest dependent chain of a round; it
//                  has its SIMD (waves 3, 7, 11, 15) to itself — and appends it to the rotation log in global memory
//   block waves    (0-2, 4-6, …) one 2×2 block (I <= J) of A per thread: read, rotate, write to the permuted places
// The rotation table holds R = [c s; −s c] column by column, [c, −s | s, c] per pair, so that a thread that needs column
// `side` of a pair's rotation reads it with one 16-byte load at a precomputed offset (no selects on the critical chain).
// V is not touched inside the loop: its 2·r·n2 stores per round cost more than the whole round (measured: ≥ 500 cycles
// of LDS time per round on one CU in every layout tried, against a ~1000-cycle round).  The rotations are logged instead,
// and k_eigen_vreplay applies them to V afterwards on many CUs at once (rows of V are independent).

// ---- progress word shared by the two roles of k_posterior_eigen_rr (meta[0]); every launch carries its own id so that
// whatever an earlier launch left there is never mistaken for news
constexpr int kPwRoundsMask = 0x3FFFF, kPwAbort = 1 << 18, kPwFinished = 1 << 19, kPwIdShift = 20, kPwIdMask = 0x7FF;
__device__ __forceinline__ void progress_publish(int* meta, int id, int rounds, int flags) {
  // no fence: everything the word announces (rotation log, final diagonal, correction) is written with write-through
  // stores that the storing waves have waited for (s_waitcnt vmcnt(0), then the workgroup's barrier) before this store
  __hip_atomic_store(meta, (id << kPwIdShift) | flags | rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// whole-wave shifts by one lane on the DPP path of the VALU (gfx9 wave_shr:1 / wave_shl:1): no LDS crossbar trip
__device__ __forceinline__ double wave_shr1_f64(double v) {  // lane l receives the value of lane l−1 (lane 0 keeps its own)
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x138, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x138, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_shl1_f64(double v) {  // lane l receives the value of lane l+1 (lane 63 keeps its own)
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), 0x130, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), 0x130, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}

// ---------------------------------------------------------------- V <- V·J_0·J_1···  (replay of the rotation log)
// Workgroups 1.. of a problem of k_posterior_eigen_rr (one per 32 rows of V), running BESIDE its Jacobi workgroup
// (workgroup 0) on other CUs: they follow the progress word — the producer's log wave advances it every round, a few
// rounds behind its write-through stores — and apply the rotations to the eigenvector matrix while the iteration goes on,
// so that when it ends only a handful of rounds, the first-order correction and the final sort are left.
// One wave per FOUR coordinates (rows of V): lane = pair + 32·(row pair), two rows per lane, 8 row-carrying waves per
// workgroup (one CU cannot keep pace with the iteration for all 64 rows: ≈ 60 cycles per row and round).  A lane keeps its
// pair's two entries of each row in registers; one round rotates the pair and hands the results to the neighbouring pairs
// (the round-robin move: first entries travel to pair+1, second entries to pair−1, with the two turn-arounds at the
// ends): two 64-bit DPP wave shifts per row and round, no LDS traffic for the data.  The rotations of the published
// rounds are staged in LDS once per pass.  At the end every workgroup applies the correction V <- V·(I + X) (see
// k_posterior_eigen_rr) to its rows, ranks the eigenvalues, and the two workgroups exchange, per column, their
// largest-|.| candidate (one 1-KB message each, write-through stores and a flag) to fix the signs — largest-|.| component
// of every eigenvector positive, the first among equals — before every lane writes its own entries of V and Vᵀ.
constexpr int kReplayStageRounds = 64;   // rounds staged per pass (>= one sweep for ranks <= 64)
constexpr int kReplayWaves = 8;          // waves of a replay workgroup that carry rows (the others help with staging)
constexpr int kReplayRows = 4 * kReplayWaves;  // rows of V per replay workgroup
constexpr int kEigMetaCorr = 7;          // meta word: the producer left a first-order correction X in `xcorr`
constexpr int kEigMetaMu = 64;           // (double*)meta + this: the final diagonal, by position
constexpr int kEigMetaXchg = 128;        // (double*)meta + this: [2 workgroups][64 values | 64 rows] sign candidates

__device__ __forceinline__ void sc1_store(double* p, double v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ double sc1_load(const double* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ void eigen_replay_consumer(int r, const double* Vwarm, const double* rotlog, int* meta, const double* xcorr /* [n2][n2] */,
                                      double* Vout, double* Vtout, double* Sout, int launch_id, int me, int nb, int* done_word,
                                      int done_value) {
  __shared__ int s_pw;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int n2 = (r + 1) & ~1, m = n2 >> 1;
  double* s_log = s_dyn;  // kReplayStageRounds × m entries of (c, −s)
  const int kc = lane >> 5, q = lane & 31, ka = kReplayRows * me + 4 * wave + 2 * kc, kb = ka + 1;
  const bool carry = wave < kReplayWaves && kReplayRows * me + 4 * wave < r;  // (uniform) this wave holds rows of V
  const bool act = q < m && carry;
  const int qc = q < m ? q : 0;
  // this lane's pair of each of its two rows: positions 2q (first) and 2q+1 (second)
  auto v0_at = [&](int k, int p) { return (k < r && p < r) ? (Vwarm ? Vwarm[(size_t)k * r + p] : (k == p ? 1.0 : 0.0)) : 0.0; };
  double a0 = act ? v0_at(ka, 2 * q) : 0.0, a1 = act ? v0_at(ka, 2 * q + 1) : 0.0;
  double b0 = act ? v0_at(kb, 2 * q) : 0.0, b1 = act ? v0_at(kb, 2 * q + 1) : 0.0;
  int done = 0;
  bool aborted = false;
  for (;;) {
    if (tid == 0) {  // follow the producer (relaxed polls: an acquiring load would invalidate this CU's L1 every time round)
      int pw;
      for (;;) {
        pw = __hip_atomic_load(meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (((pw >> kPwIdShift) & kPwIdMask) == launch_id && ((pw & kPwRoundsMask) > done || (pw & (kPwAbort | kPwFinished)))) break;
        __builtin_amdgcn_s_sleep(2);
      }
      s_pw = pw;
#ifdef ICP_EIGEN_TIMING
      if (me == 0 && (pw & kPwFinished)) g_eigen_stamps[40] = __builtin_amdgcn_s_memrealtime();
#endif
    }
    __syncthreads();
    const int pw = s_pw;
    if (pw & kPwAbort) { aborted = true; break; }  // cancelled decomposition: V stays untouched
    const int avail = pw & kPwRoundsMask;
    while (done < avail) {
      const int n = min(avail - done, kReplayStageRounds);
      // (the log is written with write-through stores and read with loads served by L2: its addresses are reused by every
      // decomposition, a plain load could hit a stale line of this CU's L1; the word is advanced behind the stores' return)
      for (int e = tid; e < 2 * n * m; e += blockDim.x) s_log[e] = sc1_load(rotlog + 2 * (size_t)done * m + e);
      __syncthreads();
      if (carry) {
        for (int rl = 0; rl < n; rl += 8) {
          dbl2 cs[8];
#pragma unroll
          for (int u = 0; u < 8; ++u) cs[u] = *(const dbl2*)&s_log[2 * (min(rl + u, n - 1) * m + qc)];
#pragma unroll
          for (int u = 0; u < 8; ++u) {
            if (rl + u < n) {  // uniform
              // rotated first / second entry of both rows
              const double fa = fma(cs[u].x, a0, cs[u].y * a1), ga = fma(-cs[u].y, a0, cs[u].x * a1);
              const double fb = fma(cs[u].x, b0, cs[u].y * b1), gb = fma(-cs[u].y, b0, cs[u].x * b1);
              // round-robin move (rr_dst): first entries go one pair up, except pair 0 (stays) and pair m−1 (becomes its
              // own second); second entries go one pair down, except pair 0 (becomes the first of pair 1)
              const double ua = wave_shr1_f64(q == 0 ? ga : fa), da = wave_shl1_f64(ga);
              const double ub = wave_shr1_f64(q == 0 ? gb : fb), db = wave_shl1_f64(gb);
              a0 = q == 0 ? fa : ua; a1 = q == m - 1 ? fa : da;
              b0 = q == 0 ? fb : ub; b1 = q == m - 1 ? fb : db;
            }
          }
        }
      }
      __syncthreads();
      done += n;
    }
    if (pw & kPwFinished) break;
  }
#ifdef ICP_EIGEN_TIMING
  if (me == 0 && tid == 0) g_eigen_stamps[41] = __builtin_amdgcn_s_memrealtime();
#endif
  // ---- correction, ranks, signs, output
  double* s_x = s_dyn;                       // [n2][n2] first-order correction (the log's region: every wave is past it)
  double* s_mu = s_dyn + 4096;               // [64] final diagonal by position
  int* s_rank = (int*)(s_dyn + 4096 + 64);   // [64]
  double* s_bv = s_dyn + 4096 + 128;         // [64] signed winner per position
  int* s_bk = (int*)(s_dyn + 4096 + 192);    // [64] … and its row
  double* s_pv = s_dyn + 4096 + 256;         // [8 waves][64 positions] candidate value (signed)
  int* s_pk = (int*)(s_dyn + 4096 + 256 + kReplayWaves * 64);  // … and its row
  // s_dyn[5632 …): [32 rows][64 positions] the workgroup's rows, for the correction
  double* xchg = (double*)meta + kEigMetaXchg;
  if (!aborted) {
    const int has_corr = __hip_atomic_load(meta + kEigMetaCorr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tid < n2) s_mu[tid] = sc1_load((const double*)meta + kEigMetaMu + tid);
    if (has_corr)
      for (int e = tid; e < n2 * n2; e += blockDim.x) s_x[e] = sc1_load(xcorr + e);
    __syncthreads();
    if (has_corr) {  // row·(I + X): entry p of a row gains Σ_i row[i]·X[i][p].  The rows go through LDS (every lane then reads
      // the SAME two entries of its row — a broadcast — beside its own two columns of X): a quarter of the LDS cycles of
      // fetching the entries from their lanes by shuffles, which is what this step is bound by
      if (act) {
        *(dbl2*)&s_dyn[5632 + (4 * wave + 2 * kc) * 64 + 2 * q] = dbl2{a0, a1};
        *(dbl2*)&s_dyn[5632 + (4 * wave + 2 * kc + 1) * 64 + 2 * q] = dbl2{b0, b1};
      }
      __syncthreads();
      if (carry) {
        double ca0 = 0.0, ca1 = 0.0, cb0 = 0.0, cb1 = 0.0;
        const int ra = 5632 + (4 * wave + 2 * kc) * 64, rb = ra + 64;
        for (int i = 0; i < n2; i += 2) {
          const dbl2 va = *(const dbl2*)&s_dyn[ra + i], vb = *(const dbl2*)&s_dyn[rb + i];
          const dbl2 xe = *(const dbl2*)&s_dyn[i * n2 + 2 * qc], xo = *(const dbl2*)&s_dyn[(i + 1) * n2 + 2 * qc];
          ca0 = fma(va.x, xe.x, ca0); ca1 = fma(va.x, xe.y, ca1); cb0 = fma(vb.x, xe.x, cb0); cb1 = fma(vb.x, xe.y, cb1);
          ca0 = fma(va.y, xo.x, ca0); ca1 = fma(va.y, xo.y, ca1); cb0 = fma(vb.y, xo.x, cb0); cb1 = fma(vb.y, xo.y, cb1);
        }
        if (act) { a0 += ca0; a1 += ca1; b0 += cb0; b1 += cb1; }
      }
    }
    // sign candidates per position: largest |.| over the rows, the lowest row among equals
    if (wave < kReplayWaves) {
      const bool va = act && ka < r, vb = act && kb < r;
      double m0 = va ? fabs(a0) : -1.0, m1 = va ? fabs(a1) : -1.0, c0 = a0, c1 = a1;
      int k0 = ka, k1 = ka;
      if (vb && fabs(b0) > m0) { m0 = fabs(b0); c0 = b0; k0 = kb; }
      if (vb && fabs(b1) > m1) { m1 = fabs(b1); c1 = b1; k1 = kb; }
      const double o0 = __shfl_xor(m0, 32, 64), o1 = __shfl_xor(m1, 32, 64), w0 = __shfl_xor(c0, 32, 64), w1 = __shfl_xor(c1, 32, 64);
      const int ok0 = __shfl_xor(k0, 32, 64), ok1 = __shfl_xor(k1, 32, 64);
      if (o0 > m0 || (o0 == m0 && ok0 < k0)) { m0 = o0; c0 = w0; k0 = ok0; }
      if (o1 > m1 || (o1 == m1 && ok1 < k1)) { m1 = o1; c1 = w1; k1 = ok1; }
      if (lane < 32 && q < m) {
        s_pv[wave * 64 + 2 * q] = m0 < 0.0 ? 0.0 : c0; s_pk[wave * 64 + 2 * q] = m0 < 0.0 ? 0x7fffffff : k0;
        s_pv[wave * 64 + 2 * q + 1] = m1 < 0.0 ? 0.0 : c1; s_pk[wave * 64 + 2 * q + 1] = m1 < 0.0 ? 0x7fffffff : k1;
      }
    }
    __syncthreads();
    if (tid < n2) {
      double bv = s_pv[tid];
      int bk = s_pk[tid];
      for (int w = 1; w < kReplayWaves; ++w) {
        const double v = s_pv[w * 64 + tid];
        const int kk = s_pk[w * 64 + tid];
        if (kk != 0x7fffffff && (bk == 0x7fffffff || fabs(v) > fabs(bv) || (fabs(v) == fabs(bv) && kk < bk))) { bv = v; bk = kk; }
      }
      s_bv[tid] = bv; s_bk[tid] = bk;
      if (nb > 1) { sc1_store(xchg + me * 128 + tid, bv); sc1_store(xchg + me * 128 + 64 + tid, (double)bk); }
      // eigenvalues of D M⁻¹ D are 1/μ; S descending = μ ascending (ties: lower position first); the dummy sorts last
      int rank = 0;
      const double mi = s_mu[tid];
      for (int j = 0; j < n2; ++j) rank += (s_mu[j] < mi) || (s_mu[j] == mi && j < tid);
      s_rank[tid] = rank;
      if (me == 0 && rank < r) sc1_store(Sout + rank, 1.0 / mi);
    }
    if (nb > 1) {  // exchange with the other workgroup: message out (write-through, drained), flag up; its flag, its message
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      const int o = 1 - me;
      if (tid == 0) {
        __hip_atomic_store(meta + 2 + me, launch_id, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(meta + 2 + o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != launch_id) __builtin_amdgcn_s_sleep(1);
      }
      __syncthreads();
      if (tid < n2) {
        const double v = sc1_load(xchg + o * 128 + tid);
        const int kk = (int)sc1_load(xchg + o * 128 + 64 + tid);
        double bv = s_bv[tid];
        const int bk = s_bk[tid];
        if (kk != 0x7fffffff && (bk == 0x7fffffff || fabs(v) > fabs(bv) || (fabs(v) == fabs(bv) && kk < bk))) bv = v;
        s_bv[tid] = bv;
      }
      if (tid == 0) __hip_atomic_store(meta + 2 + o, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // read: back to idle (ids repeat after 2047 launches)
    }
    __syncthreads();
    if (act) {  // (write-through stores: nothing to write back before the completion word)
      const int p0 = 2 * q, p1 = 2 * q + 1, r0 = s_rank[p0], r1 = s_rank[p1];
      const bool n0 = s_bv[p0] < 0.0, n1 = s_bv[p1] < 0.0;
      if (ka < r) {
        if (r0 < r) { const double v = n0 ? -a0 : a0; sc1_store(Vout + (size_t)ka * r + r0, v); sc1_store(Vtout + (size_t)r0 * r + ka, v); }
        if (r1 < r) { const double v = n1 ? -a1 : a1; sc1_store(Vout + (size_t)ka * r + r1, v); sc1_store(Vtout + (size_t)r1 * r + ka, v); }
      }
      if (kb < r) {
        if (r0 < r) { const double v = n0 ? -b0 : b0; sc1_store(Vout + (size_t)kb * r + r0, v); sc1_store(Vtout + (size_t)r0 * r + kb, v); }
        if (r1 < r) { const double v = n1 ? -b1 : b1; sc1_store(Vout + (size_t)kb * r + r1, v); sc1_store(Vtout + (size_t)r1 * r + kb, v); }
      }
    }
  }
#ifdef ICP_EIGEN_TIMING
  if (me == 0 && tid == 0) g_eigen_stamps[42] = __builtin_amdgcn_s_memrealtime();
#endif
  // every wave's (write-through) stores have left before the workgroup is counted out; the last workgroup out puts the
  // shared words back to idle and raises the completion word
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    if (nb == 1 || __hip_atomic_fetch_add(meta + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nb - 1) {
      __hip_atomic_store(meta + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(meta, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (the producer has long finished; nobody reads it any more)
      // this decomposition is complete (or dropped): whoever waits for it alone need not wait for the rest of the launch
      if (done_word) __hip_atomic_store(done_word, done_value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

constexpr int kRrLd = 66;                 // row stride of A (doubles): rows 16 B apart modulo the 256-B bank window
constexpr int kRrSzA = 64 * kRrLd;        // one buffer of A, sized for rank 64 whatever r is: every offset below is a constant
constexpr int kRrSzC = 4 * 32;            // one rotation table
constexpr int kRrOC = 0, kRrOA = 2 * kRrSzC, kRrOV = kRrOA + 2 * kRrSzA;  // table[2] | A[2] | Vt (warm start) | T (its transform)
constexpr int kRrLogWave = 14;            // never a block wave (at most 9 of those, on waves 0-2, 4-6, 8-10)
constexpr int kRrPollWave = 13;           // … nor this one: a speculative decomposition's cancel word is polled here (a slow read of
                                          // pinned memory, which must not sit in the log wave's memory queue: it counts its stores)
constexpr int kRrLogLag = 8;              // the progress word trails the log wave's write-through stores by this many rounds
template <int N> struct IntC { static constexpr int value = N; };

// One launch decomposes up to two posteriors side by side (the two ICP directions of a chain step): problem p owns the
// workgroups [p·per, (p+1)·per), the first of which iterates while the others replay.
// (struct EigenProblem: icp_kernels.hpp — the on-device chain loop patches these records in device memory)
// The batch record.  EigenBatch<2>: the two directions of one chain step, by value in the kernel arguments.  EigenBatchMem: the
// decompositions of a batch of chains (icp_chain_step_batched) — any number of them in ONE launch, the records read in place from
// pinned host memory (136 bytes per workgroup, once); every workgroup announces itself in `arrive` when it starts, so that the
// batch's launch sequence can be held back until all of them are resident (k_step_batch_args: its first launch fills the chip with
// workgroups that spin on these decompositions' completion words, and must not get there first).
template <int CAP> struct EigenBatch {
  int n; EigenProblem p[CAP];
  __device__ __forceinline__ void announce() const {}
  __device__ __forceinline__ bool skipped(int) const { return false; }
};
struct EigenBatchMem {
  int n; const EigenProblem* p; int* arrive;
  const int* skip = nullptr;  // (optional) skip[problem] != 0: nothing to decompose this time (the on-device chain loop launches the
                              // decompositions of every chain every step; only the chains that moved have one)
  __device__ __forceinline__ void announce() const {
    if (arrive && threadIdx.x == 0) __hip_atomic_fetch_add(arrive, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __device__ __forceinline__ bool skipped(int which) const { return skip != nullptr && skip[which] != 0; }
};
static_assert(sizeof(EigenBatch<2>) + 64 <= 4096, "the batch record must fit the kernel argument segment");

template <class Batch>
__global__ void __launch_bounds__(1024) k_posterior_eigen_rr(int r, const double* __restrict__ sqrt_lambda_launch, int ldk, int max_sweeps,
                                                              int no_corr /* 1: sweep to the strict test (A/B, tests) */, Batch batch) {
  batch.announce();
  const int per = 1 + (r + kReplayRows - 1) / kReplayRows;  // workgroups per problem: the iteration + the replay (32 rows each)
  const int which = (int)blockIdx.x / per, local = (int)blockIdx.x - which * per;
  if (batch.skipped(which)) return;  // (uniform per workgroup)
  const EigenProblem pb = batch.p[which];
  const double* __restrict__ sqrt_lambda = pb.sqrt_lambda ? pb.sqrt_lambda : sqrt_lambda_launch;
  const double* __restrict__ M = pb.M;
  const double* Vwarm = pb.Vwarm;
  double* Vout = pb.Vout;
  double* Vtout = pb.Vtout;
  double* __restrict__ Sout = pb.Sout;
  int* __restrict__ status = pb.status;
  double* rotlog = pb.rotlog;
  int* meta = pb.meta;
  double* vpos = pb.vpos;
  const EigenSpec spec = pb.spec;
  const int launch_id = pb.launch_id;
  int* host_status = pb.host_status;
  if (Vwarm && !(Vwarm[0] == Vwarm[0])) Vwarm = nullptr;  // the basis of a decomposition that gave up (see below): cold start
  if (local != 0) {
    eigen_replay_consumer(r, Vwarm, rotlog, meta, vpos, Vout, Vtout, Sout, launch_id, local - 1, per - 1, pb.done_word, pb.done_value);
    return;
  }
  __shared__ double s_red[16], s_red2[16];
  __shared__ int s_cancel, s_bad[16];
  const int tid = threadIdx.x, nt = blockDim.x, lane = tid & 63, wave = tid >> 6;
  const int n2 = (r + 1) & ~1, m = n2 >> 1;
  // buffers are addressed as s_dyn[offset] with integer offsets: a table of pointers would turn every access into a
  // FLAT instruction (address space lost), several times slower than the DS path.  The offsets of the round loop are
  // compile-time constants plus one per-thread register, so they fold into the DS instructions' immediate fields.
  constexpr int ld = kRrLd, szA = kRrSzA, szC = kRrSzC, oA = kRrOA, oV = kRrOV, oC = kRrOC;
  const int szV = n2 * ldk, oT = oV + szV;
#define LDS_A(b, i) s_dyn[oA + (b) * szA + (i)]
#define LDS_VT(i) s_dyn[oV + (i)]
#define LDS_T(i) s_dyn[oT + (i)]
#define LDS_C(b, i) s_dyn[oC + (b) * szC + (i)]
  EIG_STAMP(0);
  // a speculative decomposition polls its cancel word (pinned host memory: a slow read, so one thread of an otherwise
  // idle wave fetches it while the others work, and the block looks at the copy at the next convenient barrier)
  const bool is_poll = spec.cancel != nullptr && tid == 64 * kRrPollWave + 63;
  if (tid == 0) s_cancel = 0;
  // ---- N = D⁻¹ M D⁻¹ (symmetrised), padded; Vt = (warm start or identity)ᵀ, padded with zeros
  for (int e = tid; e < szV; e += nt) LDS_VT(e) = 0.0;
  __syncthreads();
  // ---- fixed work of this thread (indices only: nothing here depends on the matrix, so a speculative launch does it —
  // and the staging of the warm-start basis — while it still waits for its input)
  const int nA = m * (m + 1) / 2, nbw = (nA + 63) >> 6;
  const int widx = (wave & 3) == 3 ? -1 : wave - (wave >> 2);  // index among the waves of SIMDs 0-2 (12 of them)
  const int bidx = (widx >= 0 && widx < nbw) ? widx * 64 + lane : nA;
  const bool is_blk = bidx < nA;
  int bI = 0, bJ = 0, b_rd = 0, w00 = 0, w01 = 0, w10 = 0, w11 = 0;
  if (is_blk) {  // unrank the upper triangle row-major
    int base = 0;
    while (base + (m - bI) <= bidx) { base += m - bI; ++bI; }
    bJ = bI + (bidx - base);
    b_rd = 2 * bI * ld + 2 * bJ;
    const int R0 = rr_dst(2 * bI, m), R1 = rr_dst(2 * bI + 1, m), C0 = rr_dst(2 * bJ, m), C1 = rr_dst(2 * bJ + 1, m);
    w00 = min(R0, C0) * ld + max(R0, C0); w01 = min(R0, C1) * ld + max(R0, C1);
    w10 = min(R1, C0) * ld + max(R1, C0); w11 = min(R1, C1) * ld + max(R1, C1);
  }
  const bool is_rot = wave == 3 && lane < m;
  int rp_dp = 0, rp_dq = 0, rp_ob = 0, rp_cp = 0, rp_cq = 0, rp_cl = 0, rp_ch = 0, rp_k = 0;
  if (wave == 3) {
    __builtin_amdgcn_s_setprio(3);
    rp_k = is_rot ? lane : 0;
    const int p = rr_src(2 * rp_k, m), q = rr_src(2 * rp_k + 1, m);
    const int ip = p >> 1, ap = p & 1, iq = q >> 1, aq = q & 1, lo = min(ip, iq), hi = max(ip, iq);
    const int ra = ip < iq ? ap : aq, ca = ip < iq ? aq : ap;  // (row in pair lo, column in pair hi) of the new off-diagonal entry
    rp_dp = 2 * ip * ld + 2 * ip; rp_dq = 2 * iq * ld + 2 * iq; rp_ob = 2 * lo * ld + 2 * hi;
    rp_cp = 4 * ip + 2 * ap; rp_cq = 4 * iq + 2 * aq; rp_cl = 4 * lo + 2 * ra; rp_ch = 4 * hi + 2 * ca;
  }
  for (int e = tid; e < r * r; e += nt) {  // i = coordinate, j = position
    const int i = e / r, j = e - i * r;
    LDS_VT(j * ldk + i) = Vwarm ? Vwarm[e] : (i == j ? 1.0 : 0.0);
  }
  EIG_STAMP(50);
  if (spec.ready) {  // enqueued ahead of its input: wait for the launch that announces it (or for the cancellation).
    // Should that launch not come forward within 5 ms — kernels of different streams forced to run one at a time by a
    // tool, say — give up and say so in the pinned status: the host then repeats the decomposition the ordinary way.
    if (is_poll) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {
        if (__hip_atomic_load(spec.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - spec.ready_seq >= 0) break;
        if (__hip_atomic_load(spec.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == spec.seq) { s_cancel = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 500000) { s_cancel = 2; break; }
        __builtin_amdgcn_s_sleep(32);
      }
      if (spec.wait_ticks) atomicAdd((unsigned long long*)spec.wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0));
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (acquire side for the plain loads of the partials below)
  }
  EIG_STAMP(51);
  if (is_poll && !spec.ready && __hip_atomic_load(spec.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == spec.seq) s_cancel = 1;
  if (spec.splits > 0) {
    // M = I + Σ_s partial_s from the split-K partials of the regression launch (lower triangle of (r+1)² matrices, summed in
    // split order from 0.0 like the factorisation does).  One 16-byte piece (row i, columns 2jp, 2jp+1) per thread and row
    // half, every split's load in flight at once: the partials sit in other CUs' L2 slices, and this CU's share of them
    // (13 × 21 KB at rank 51) is what the step costs — dependent loads took 8-10 µs here, this takes ≈ 2.
    for (int e = tid; e < n2 * n2; e += nt) {
      const int i = e / n2, j = e - i * n2;
      LDS_A(0, i * ld + j) = (i == j && i >= r) ? 1e300 : 0.0;
    }
    __syncthreads();
    const size_t nn = (size_t)(r + 1) * (r + 1);
    const int jp = tid & 31, j0 = 2 * jp;
    dbl2 acc[2] = {dbl2{0.0, 0.0}, dbl2{0.0, 0.0}};
    bool live[2];
    size_t off[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int i = (tid >> 5) + 32 * h;
      live[h] = i < r && j0 <= i;
      off[h] = live[h] ? (size_t)i * (r + 1) + j0 : 0;  // (r + 1 even or odd: the piece is read as two 8-byte halves when unaligned)
    }
    const bool aligned = ((r + 1) & 1) == 0;
    int sp = 0;
    for (; sp + 8 <= spec.splits; sp += 8) {
      dbl2 p[8][2];
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const double* src = M + (size_t)(sp + q8) * nn + off[h];
          p[q8][h] = aligned ? *(const dbl2*)src : dbl2{src[0], src[1]};
        }
#pragma unroll
      for (int q8 = 0; q8 < 8; ++q8)
#pragma unroll
        for (int h = 0; h < 2; ++h) acc[h] += p[q8][h];
    }
    for (; sp < spec.splits; ++sp) {
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const double* src = M + (size_t)sp * nn + off[h];
        acc[h] += aligned ? *(const dbl2*)src : dbl2{src[0], src[1]};
      }
    }
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (live[h]) {
        const int i = (tid >> 5) + 32 * h;
        const double si = sqrt_lambda[i];
        double v0 = acc[h].x + (i == j0 ? 1.0 : 0.0);
        v0 = v0 / (si * sqrt_lambda[j0]);
        LDS_A(0, i * ld + j0) = v0; LDS_A(0, j0 * ld + i) = v0;
        if (j0 + 1 <= i) {
          double v1 = acc[h].y + (i == j0 + 1 ? 1.0 : 0.0);
          v1 = v1 / (si * sqrt_lambda[j0 + 1]);
          LDS_A(0, i * ld + j0 + 1) = v1; LDS_A(0, (j0 + 1) * ld + i) = v1;
        }
      }
  } else {
    for (int e = tid; e < n2 * n2; e += nt) {
      const int i = e / n2, j = e - i * n2;
      double v = i == j ? 1e300 : 0.0;
      if (i < r && j < r) v = 0.5 * (M[(size_t)i * r + j] + M[(size_t)j * r + i]) / (sqrt_lambda[i] * sqrt_lambda[j]);
      LDS_A(0, i * ld + j) = v;
    }
  }
  __syncthreads();
  if (s_cancel) {  // cancelled (or timed out) before it started: nothing is written
    if (tid == 0) {
      progress_publish(meta, launch_id, 0, kPwAbort);
      if (s_cancel == 2) {  // timed out: tell the host, and mark the basis that was never written so that no later
        // decomposition takes it for a warm start (a NaN in its first entry; a finished decomposition overwrites it)
        if (host_status) __hip_atomic_store(host_status, kEigenGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        Vout[0] = __builtin_nan("");
      }
    }
    return;
  }
  EIG_STAMP(1);
  if (Vwarm) {  // A <- Vᵀ A V (nearly diagonal when V diagonalised a nearby posterior) on the f64 matrix cores: one 16×16
    // output tile per wave, the contraction in steps of 4 (v_mfma_f64_16x16x4_f64: lane l supplies A[l&15][l>>4] and
    // B[l>>4][l&15], result register g is D[(l>>4) + 4g][l&15]).  Both products read their operands along rows of LDS
    // images (row = l&15, k = l>>4: rows are 16 B apart modulo the 256-B bank window, conflict free); indices >= r (the
    // dummy of an odd rank, the padding of the tiles) enter as zeros.
    const int tI = wave >> 2, tJ = wave & 3, nT = (n2 + 15) >> 4, l15 = lane & 15, l4 = lane >> 4;
    {  // Tt[j][i] = Σ_k Vt[j][k]·A[k][i]   (tile rows j, tile columns i)
      const int j = 16 * tI + l15, i = 16 * tJ + l15;
      const bool vj = j < r, vi = i < r;
      const int jc = vj ? j : 0, ic = vi ? i : 0;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      if (tI < nT && tJ < nT) {
        // eight steps' operands at a time (one trip of LDS latency), then their MFMAs back to back
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const int k = 4 * (h + st) + l4;
          const bool vk = k < r;
          const int kk = vk ? k : 0;
          const double a = LDS_VT(jc * ldk + kk), b = LDS_A(0, kk * ld + ic);
          av[st] = (vj && vk) ? a : 0.0; bv[st] = (vi && vk) ? b : 0.0;
        }
#pragma unroll
        for (int st = 0; st < 8; ++st)
          if (4 * (h + st) < n2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * tI + l4 + 4 * g;
          if (row < n2 && i < n2) LDS_T(row * ldk + i) = acc[g];
        }
      }
    }
    __syncthreads();
    {  // A[i][j] = Σ_k Vt[i][k]·Tt[j][k], tiles of the upper triangle only
      const int i = 16 * tI + l15, j = 16 * tJ + l15;
      const bool vi = i < r, vj = j < r;
      const int ic = vi ? i : 0, jc = vj ? j : 0;
      d4_t acc = {0.0, 0.0, 0.0, 0.0};
      if (tI <= tJ && tJ < nT) {
#pragma unroll
        for (int h = 0; h < 16; h += 8) {
        double av[8], bv[8];
#pragma unroll
        for (int st = 0; st < 8; ++st) {
          const int k = 4 * (h + st) + l4;
          const bool vk = k < r;
          const int kk = vk ? k : 0;
          const double a = LDS_VT(ic * ldk + kk), b = LDS_T(jc * ldk + kk);
          av[st] = (vi && vk) ? a : 0.0; bv[st] = (vj && vk) ? b : 0.0;
        }
#pragma unroll
        for (int st = 0; st < 8; ++st)
          if (4 * (h + st) < n2) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[st], bv[st], acc, 0, 0, 0);
        }
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int row = 16 * tI + l4 + 4 * g;
          if (row <= j && j < r) LDS_A(0, row * ld + j) = acc[g];
        }
      }
    }
    __syncthreads();
  }
  EIG_STAMP(2);

  if (wave == 3) {
    // rotations of the first round, straight from the diagonal blocks
    const int o = 2 * rp_k * ld + 2 * rp_k;
    const Rot R = jacobi_rotation(LDS_A(0, o), LDS_A(0, o + 1), LDS_A(0, o + ld + 1));
    if (is_rot) { LDS_C(0, 4 * rp_k) = R.c; LDS_C(0, 4 * rp_k + 1) = -R.s; LDS_C(0, 4 * rp_k + 2) = R.s; LDS_C(0, 4 * rp_k + 3) = R.c; }
  }
  const bool is_log = wave == kRrLogWave && lane < m;
  const size_t lstride = (size_t)2 * m;  // doubles per logged round: (c, −s) of every pair
  __syncthreads();
  EIG_STAMP(3);

  // One round: reads buffers `cur`, writes buffers `cur ^ 1`.  `cur` is a template constant (the loop below alternates
  // the two instantiations), so no address is computed inside the loop at all.
  int n_rounds = 0, pub_floor = 0;
  auto round = [&](auto CUR) {
    constexpr int cur = decltype(CUR)::value;
    constexpr int ac = oA + cur * szA, an = oA + (cur ^ 1) * szA, cc = oC + cur * szC, cn = oC + (cur ^ 1) * szC;
    if (is_blk) {
      const dbl2 r0 = lds2(&s_dyn[ac + b_rd]), r1 = lds2(&s_dyn[ac + b_rd + ld]);
      const dbl2 c1 = lds2(&s_dyn[cc + 4 * bI]), c2 = lds2(&s_dyn[cc + 4 * bJ]);  // (c, −s)
      const bool dg = bI == bJ;  // diagonal block: its lower entry is not stored
      const B22 n = rot_block(B22{r0.x, r0.y, dg ? r0.y : r1.x, r1.y}, c1.x, -c1.y, c2.x, -c2.y);
      // (diagonal block: w01 and w10 are the same address and a01, a10 agree to rounding — either store serves)
      s_dyn[an + w00] = n.a00; s_dyn[an + w01] = n.a01; s_dyn[an + w10] = n.a10; s_dyn[an + w11] = n.a11;
    } else if (wave == 3) {
      // the next round pairs the contents of old positions p (pair ip, side ap) and q (pair iq, side aq); their three
      // entries after this round's rotations, by the block threads' own expressions
      const dbl2 dp0 = lds2(&s_dyn[ac + rp_dp]), dp1 = lds2(&s_dyn[ac + rp_dp + ld]);
      const dbl2 dq0 = lds2(&s_dyn[ac + rp_dq]), dq1 = lds2(&s_dyn[ac + rp_dq + ld]);
      const dbl2 b0 = lds2(&s_dyn[ac + rp_ob]), b1 = lds2(&s_dyn[ac + rp_ob + ld]);
      const dbl2 kp = lds2(&s_dyn[cc + rp_cp]), kq = lds2(&s_dyn[cc + rp_cq]);  // rotation column (p, q) of each factor
      const dbl2 kl = lds2(&s_dyn[cc + rp_cl]), kh = lds2(&s_dyn[cc + rp_ch]);
      __builtin_amdgcn_sched_barrier(0);  // all ten reads in flight together: ONE trip of LDS latency on the chain
      // entry = Σ (rotation entry products)·(block entries), as two independent multiply-add pairs and one add; it
      // only steers the next angle, so it need not match the block threads' rounding
      const double app = fma(kp.x * kp.x, dp0.x, (kp.x * kp.y) * dp0.y) + fma(kp.y * kp.x, dp0.y, (kp.y * kp.y) * dp1.y);
      const double aqq = fma(kq.x * kq.x, dq0.x, (kq.x * kq.y) * dq0.y) + fma(kq.y * kq.x, dq0.y, (kq.y * kq.y) * dq1.y);
      const double apq = fma(kh.x * kl.x, b0.x, (kh.x * kl.y) * b1.x) + fma(kh.y * kl.x, b0.y, (kh.y * kl.y) * b1.y);
      const Rot R = jacobi_rotation(app, apq, aqq);
      if (is_rot) {
        *(dbl2*)&s_dyn[cn + 4 * rp_k] = dbl2{R.c, -R.s};
        *(dbl2*)&s_dyn[cn + 4 * rp_k + 2] = dbl2{R.s, R.c};
      }
    } else if (wave == kRrLogWave) {  // the rotations this round applies, for the replay workgroups (negligible ones as
      // identities): two write-through stores per pair; every fourth round the progress word is advanced to kRrLogLag rounds
      // behind — 2 + ¼ memory operations per round in this wave's queue, so all but the youngest 2·lag + lag/4 − 1 of them
      // being done means the rounds up to n_rounds − lag have arrived
      if (is_log) {
        const dbl2 k = lds2(&s_dyn[cc + 4 * lane]);
        const dbl2 w = fabs(k.y) >= 2e-17 ? k : dbl2{1.0, 0.0};
        double* dst = rotlog + (size_t)n_rounds * lstride + 2 * lane;
        sc1_store(dst, w.x); sc1_store(dst + 1, w.y);
      }
      if ((n_rounds & 3) == 3) {
        asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * kRrLogLag + kRrLogLag / 4 - 1) : "memory");
        const int upto = n_rounds + 1 - kRrLogLag;
        if (lane == 0 && upto > pub_floor) __hip_atomic_store(meta, (launch_id << kPwIdShift) | upto, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
    }
    __syncthreads();
    ++n_rounds;
  };
  int converged = 0, n_sweeps = 0, in_sweep = 0, use_corr = 0;
  // (a warm-started iteration has never met either test after one sweep, a cold one never before its third: those passes
  // — two barriers and a reduction each — are skipped; were the matrix diagonal already, one more sweep would be harmless)
  const int first_test = Vwarm ? 1 : 2;
  auto sweep_end = [&](int cur) -> bool {  // -> stop?
    EIG_STAMP(4 + 2 * n_sweeps);
    in_sweep = 0;
    if (n_sweeps < first_test && n_sweeps + 1 < max_sweeps) {
      ++n_sweeps;
      EIG_STAMP(3 + 2 * n_sweeps);
      return s_cancel != 0;
    }
    // this sweep's rotations are in the log (written through; the log wave's own progress stores have landed, too): the
    // replay workgroups may have all of them (published behind the barrier below)
    if (wave == kRrLogWave) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pub_floor = n_rounds;
    // Two ways to be done, both from one pass over the stored upper triangle (thread = row, 4 columns; one barrier):
    //   strict  off(A)² <= 1e-26·Σ diag²: nothing left to do;
    //   loose   every |A_ij| <= 4e-6·|A_jj − A_ii|: what one more sweep would do to the eigenvectors is, to first order,
    //           V <- V·(I + X) with X_ij = A_ij/(A_jj − A_ii) (antisymmetric), all |X_ij| <= 4e-6 — the replay workgroups
    //           apply that instead (error of the correction ~ X²: 1e-11, against 19 µs for the sweep).  The Jacobi sweeps
    //           converge quadratically, so the sweep before the last is the one that meets this test.
    double off = 0.0, dg = 0.0;
    bool bad = false;
    {
      const int i = tid >> 4, j0 = (tid & 15) << 2;
      if (i < n2 && j0 + 3 >= i && j0 < n2) {
        const dbl2 u = lds2(&LDS_A(cur, i * ld + j0)), w = lds2(&LDS_A(cur, i * ld + j0 + 2));
        const double dii = LDS_A(cur, i * ld + i);
        const double v[4] = {u.x, u.y, w.x, w.y};
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int j = j0 + c;
          if (j < n2) {
            if (j == i) { if (v[c] < 1e299) dg = fma(v[c], v[c], dg); }
            else if (j > i) {
              off = fma(2.0 * v[c], v[c], off);
              bad = bad || fabs(v[c]) > kLooseTau * fabs(LDS_A(cur, j * ld + j) - dii);
            }
          }
        }
      }
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
    const bool wave_bad = __any(bad);
    if (lane == 0) { s_red[wave] = off; s_red2[wave] = dg; s_bad[wave] = wave_bad ? 1 : 0; }
    __syncthreads();
    if (tid == 0) progress_publish(meta, launch_id, n_rounds, 0);
    off = 0.0; dg = 0.0;
    int any_bad = 0;
#pragma unroll
    for (int w = 0; w < 16; ++w) { off += s_red[w]; dg += s_red2[w]; any_bad |= s_bad[w]; }
    const int strict = off <= 1e-26 * dg;
    const int loose = !any_bad && off <= dg;  // (off <= dg: false for NaN)
    converged = strict || (loose && !no_corr);
    use_corr = converged && !strict;
    if (tid == 0 && n_sweeps < 8) ((double*)(meta + 80))[n_sweeps] = off / dg;  // diagnostic: off(A)²/Σdiag² after each sweep
    ++n_sweeps;
    EIG_STAMP(3 + 2 * n_sweeps);
    return converged || n_sweeps >= max_sweeps || s_cancel;  // (s_cancel: stored by the poll thread rounds ago)
  };
  int cur = 0, polled = 0x80000000;
  const int poll_use = (n2 - 1) >> 2;
  if (max_sweeps > 0)
    for (;;) {
      // the poll of a sweep is issued at its start and looked at half a sweep later, when the word has long arrived
      // (waiting for it on the spot would hold every wave at this round's barrier for a microsecond or two)
      if (is_poll && (in_sweep >> 1) == 0) polled = __hip_atomic_load(spec.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      if (is_poll && (in_sweep >> 1) == poll_use && polled == spec.seq) s_cancel = 1;
      round(IntC<0>{}); cur = 1;
      if (++in_sweep == n2 - 1 && sweep_end(cur)) break;
      round(IntC<1>{}); cur = 0;
      if (++in_sweep == n2 - 1 && sweep_end(cur)) break;
    }
  if (s_cancel && !converged) {  // given up: no status, no eigenvalues; the replay workgroups drop what they have
    if (tid == 0) progress_publish(meta, launch_id, n_rounds, kPwAbort);
    return;
  }
  if (tid == 0) {
    ++meta[100 + min(n_sweeps, 15)];  // diagnostic: histogram of sweep counts on this work buffer
    status[0] = converged ? 0 : 2; status[-1] = n_sweeps;
    if (host_status) __hip_atomic_store(host_status, converged ? 0 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  EIG_STAMP(62);
  // ---- hand-over to the replay workgroup: the final diagonal by position (it ranks the eigenvalues: those of D M⁻¹ D are
  // 1/μ, S descending = μ ascending, the dummy sorts last and is dropped) and, when the iteration stopped on the loose
  // test, the first-order correction X
  double* xg = vpos;
  if (tid < n2) sc1_store((double*)meta + kEigMetaMu + tid, LDS_A(cur, tid * ld + tid));
  if (use_corr) {
    const int i = tid >> 4, j0 = (tid & 15) << 2;
    if (i < n2 && j0 + 3 >= i && j0 < n2) {
      const double dii = LDS_A(cur, i * ld + i);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int j = j0 + c;
        if (j < n2 && j >= i) {
          const double a = LDS_A(cur, i * ld + j);
          const double x = (j == i || a == 0.0) ? 0.0 : a / (LDS_A(cur, j * ld + j) - dii);
          sc1_store(xg + i * n2 + j, x);
          if (j != i) sc1_store(xg + j * n2 + i, -x);
        }
      }
    }
  }
  if (tid == 0) __hip_atomic_store(meta + kEigMetaCorr, use_corr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every wave's stores, before the barrier behind which they are released
  __syncthreads();
  if (tid == 0) progress_publish(meta, launch_id, n_rounds, kPwFinished);  // (max_sweeps == 0: nothing to replay)
  EIG_STAMP(63);
#undef LDS_A
#undef LDS_VT
#undef LDS_T
#undef LDS_C
}

// ---------------------------------------------------------------- posterior KL basis, ranks 65..200: in-place parallel Jacobi
// One CU's LDS cannot hold these matrices twice (the fixed-position kernel above reads one copy and writes the permuted
// other), but it holds the strict upper triangle ONCE, packed, beside the diagonal (rank 200: 159 KB + 1.6 KB): the classical
// parallel-order Jacobi iteration updates it in place — pair P of a round rotates (p, q) of the round-robin tournament, block
// (P1, P2) owns the four entries A[{p1,q1}][{p2,q2}] and nobody else touches them in that round.
//   workgroup 0      phase 1: one thread per pair computes (c, s) from three entries and logs it; barrier;
//                    phase 2: every thread transforms its (up to five) 2×2 blocks, R1ᵀ·B·R2, in place; barrier.
//                    ≈ 2.3 µs per round at rank 200 (LDS cycles: 5,050 blocks × 8 accesses), 199 rounds per sweep.
//   workgroups 1..   64 coordinates (rows of V) each, the slab in LDS: they follow the published sweeps and apply every
//                    round's rotations to their rows (the tournament's pairs are recomputed, only (c, s) is read from the log).
// Warm start: the launcher transforms N by the basis of a nearby posterior first (k_eigen_big_warm, two plain GEMM passes on
// many CUs), the slabs start from that basis.  Sort, signs and the two output layouts are taken by k_eigen_big_finish (one
// wave per eigenvector) behind this launch.  Ranks above 200 take the generic kernel further up.
constexpr int kBigBlocksPerThread = 5;   // 1024 threads × 5 >= 100·101/2 blocks (rank 200)
constexpr int kBigMaxRank = 200;
constexpr int kBigSlabRows = 64;
constexpr int kBigStageRounds = 16;      // rounds of (c, s) staged per pass by a replay workgroup

__device__ __forceinline__ int big_idx(int i, int j, int n) {  // packed strict upper triangle, i < j
  return i * (2 * n - i - 1) / 2 + (j - i - 1);
}
// round-robin tournament (circle method) on n2 players: slot 0 holds (mm, 0) in round 0 and keeps its first player; every
// other seat advances by one per round.  State (ra, rb) of a slot; the pair is (min, max)
__device__ __forceinline__ void rr_init(int slot, int mm, int& ra, int& rb) {
  if (slot == 0) { ra = mm; rb = 0; }
  else { ra = slot % mm; rb = (mm - slot) % mm; }
}
__device__ __forceinline__ void rr_advance(int slot, int mm, int& ra, int& rb) {
  if (slot == 0) { rb = rb + 1 == mm ? 0 : rb + 1; }
  else { ra = ra + 1 == mm ? 0 : ra + 1; rb = rb + 1 == mm ? 0 : rb + 1; }
}

// SQUARE: the upper triangle inside a full n × ld image (ranks <= 140: it fits, and an entry's address is i·ld + j); otherwise the
// packed triangle, row bases carried along with the tournament's players
template <bool SQUARE>
__global__ void __launch_bounds__(1024) k_eigen_big(int r, const double* __restrict__ A0 /* r×r, symmetric */, const double* __restrict__ Vwarm,
                                                     double* __restrict__ Vwork /* [coordinate][index] eigenvectors, unsorted */,
                                                     double* __restrict__ mu_out, double* rotlog, double* xcorr /* r×r */, int* meta,
                                                     int max_sweeps, int no_corr, int launch_id, int* __restrict__ status,
                                                     const int* __restrict__ gate /* optional: run only if *gate == 2 */) {
  if (gate && gate[0] != 2) return;  // (the fall-back of the tridiagonal route: its eigenvalues were told apart)
  const int tid = threadIdx.x, nt = blockDim.x;
  const int n = r, n2 = (r + 1) & ~1, half = n2 >> 1, mm = n2 - 1;
  __shared__ short s_p[128], s_q[128];
  __shared__ double s_red[16], s_red2[16];
  __shared__ int s_pw, s_bad[16];
  if (blockIdx.x != 0) {
    // ---------------- replay: rows [row0, row0 + rows) of V in LDS, rotated as the sweeps are published
    const int row0 = ((int)blockIdx.x - 1) * kBigSlabRows, rows = min(kBigSlabRows, r - row0);
    double* s_cs = s_dyn + kBigSlabRows * n;  // kBigStageRounds × half × (c, s); later: 16 columns of the correction
    for (int e = tid; e < rows * n; e += nt) {
      const int k = e / n, p = e - k * n;
      s_dyn[e] = Vwarm ? Vwarm[(size_t)(row0 + k) * r + p] : (row0 + k == p ? 1.0 : 0.0);
    }
    int ra = 0, rb = 0;
    if (tid < half) rr_init(tid, mm, ra, rb);
    // items of a round: (row k, pair P); the same ones every round
    constexpr int kItems = 7;  // 64 rows × 100 pairs / 1024 threads
    int itk[kItems], itP[kItems];
#pragma unroll
    for (int m = 0; m < kItems; ++m) {
      const int it = tid + m * nt;
      itk[m] = it < rows * half ? it / half : -1;
      itP[m] = it < rows * half ? it - itk[m] * half : 0;
    }
    int done = 0;
    for (;;) {
      if (tid == 0) {
        int pw;
        for (;;) {
          pw = __hip_atomic_load(meta, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          if (((pw >> kPwIdShift) & kPwIdMask) == launch_id && ((pw & kPwRoundsMask) > done || (pw & (kPwAbort | kPwFinished)))) break;
          __builtin_amdgcn_s_sleep(8);
        }
        s_pw = pw;
      }
      __syncthreads();
      const int pw = s_pw;
      const int avail = pw & kPwRoundsMask;
      while (done < avail) {
        const int nr = min(avail - done, kBigStageRounds);
        for (int e = tid; e < 2 * nr * half; e += nt) s_cs[e] = sc1_load(rotlog + 2 * (size_t)done * half + e);
        __syncthreads();
        for (int rl = 0; rl < nr; ++rl) {
          if (tid < half) {
            const int p = ra < rb ? ra : rb, q = ra < rb ? rb : ra;
            s_p[tid] = (short)p; s_q[tid] = (short)q;
            rr_advance(tid, mm, ra, rb);
          }
          __syncthreads();
#pragma unroll
          for (int m = 0; m < kItems; ++m) {
            if (itk[m] < 0) continue;
            const int k = itk[m], P = itP[m];
            const int p = s_p[P], q = s_q[P];
            if (q < r) {
              const dbl2 cs = *(const dbl2*)&s_cs[2 * (rl * half + P)];
              const double vp = s_dyn[k * n + p], vq = s_dyn[k * n + q];
              s_dyn[k * n + p] = fma(cs.x, vp, -(cs.y * vq));   // columns: [p q] <- [p q]·[c s; −s c]
              s_dyn[k * n + q] = fma(cs.y, vp, cs.x * vq);
            }
          }
          __syncthreads();
        }
        done += nr;
      }
      if (pw & (kPwFinished | kPwAbort)) break;
    }
    if (__hip_atomic_load(meta + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == launch_id) {
      // the iteration stopped on the loose test: rows·(I + X), sixteen columns of X through LDS at a time (k_posterior_eigen_rr)
      for (int j0 = 0; j0 < n; j0 += 16) {
        __syncthreads();
        for (int e = tid; e < n * 16; e += nt) {
          const int i = e >> 4, j = j0 + (e & 15);
          s_cs[e] = j < n ? sc1_load(xcorr + (size_t)i * n + j) : 0.0;
        }
        __syncthreads();
        const int k = tid >> 4, j = j0 + (tid & 15);
        if (k < rows && j < n) {
          double acc = 0.0;
          for (int i = 0; i < n; ++i) acc = fma(s_dyn[k * n + i], s_cs[i * 16 + (tid & 15)], acc);
          Vwork[(size_t)(row0 + k) * r + j] = s_dyn[k * n + j] + acc;
        }
      }
    } else {
      for (int e = tid; e < rows * n; e += nt) Vwork[(size_t)row0 * r + e] = s_dyn[e];
    }
    return;
  }
  // ---------------- the iteration
  const int ld = SQUARE ? (n | 1) : 0;
  const int n_off = SQUARE ? n * ld : n * (n - 1) / 2;
  double* s_diag = s_dyn + n_off;
  double* s_cs = s_diag + n;  // [half] (c, s) of the round's pairs
  auto at = [&](int i, int j) { return SQUARE ? i * ld + j : big_idx(i, j, n); };  // i < j
  for (int e = tid; e < n * n; e += nt) {
    const int i = e / n, j = e - i * n;
    if (i < j) s_dyn[at(i, j)] = 0.5 * (A0[e] + A0[(size_t)j * n + i]);
    else if (i == j) s_diag[i] = A0[e];
  }
  const int n_blocks = half * (half + 1) / 2;
  // this thread's blocks (P1 <= P2) and the players sitting at their four seats, advanced round by round in registers
  int bP1[kBigBlocksPerThread], bP2[kBigBlocksPerThread], a1[kBigBlocksPerThread], b1[kBigBlocksPerThread], a2[kBigBlocksPerThread],
      b2[kBigBlocksPerThread];
#pragma unroll
  for (int m = 0; m < kBigBlocksPerThread; ++m) {
    const int w = tid + nt * m;
    bP1[m] = -1; bP2[m] = 0; a1[m] = b1[m] = a2[m] = b2[m] = 0;
    if (w < n_blocks) {  // unrank the upper triangle of the pair × pair grid, row-major
      int P1 = 0, base = 0;
      while (base + (half - P1) <= w) { base += half - P1; ++P1; }
      bP1[m] = P1; bP2[m] = P1 + (w - base);
      rr_init(bP1[m], mm, a1[m], b1[m]);
      rr_init(bP2[m], mm, a2[m], b2[m]);
    }
  }
  int ra = 0, rb = 0;
  if (tid < half) rr_init(tid, mm, ra, rb);
  __syncthreads();
  int converged = 0, use_corr = 0, n_sweeps = 0, n_rounds = 0;
  for (int sweep = 0; sweep < max_sweeps && !converged; ++sweep) {
    for (int rnd = 0; rnd < mm; ++rnd) {
      if (tid < half) {
        const int p = ra < rb ? ra : rb, q = ra < rb ? rb : ra;
        Rot R{1.0, 0.0};
        if (q < r) R = jacobi_rotation(s_diag[p], s_dyn[at(p, q)], s_diag[q]);
        *(dbl2*)&s_cs[2 * tid] = dbl2{R.c, R.s};
        *(dbl2*)(rotlog + 2 * ((size_t)n_rounds * half + tid)) = dbl2{R.c, R.s};
        rr_advance(tid, mm, ra, rb);
      }
      __syncthreads();
#pragma unroll
      for (int m = 0; m < kBigBlocksPerThread; ++m) {
        if (bP1[m] < 0) continue;
        const int P1 = bP1[m], P2 = bP2[m];
        const int p1 = min(a1[m], b1[m]), q1 = max(a1[m], b1[m]), p2 = min(a2[m], b2[m]), q2 = max(a2[m], b2[m]);
        rr_advance(P1, mm, a1[m], b1[m]);
        rr_advance(P2, mm, a2[m], b2[m]);
        const dbl2 r1 = *(const dbl2*)&s_cs[2 * P1], r2 = *(const dbl2*)&s_cs[2 * P2];
        if (P1 == P2) {
          if (q1 < r) {
            const int o = at(p1, q1);
            const double apq = s_dyn[o];
            const B22 nb = rot_block(B22{s_diag[p1], apq, apq, s_diag[q1]}, r1.x, r1.y, r1.x, r1.y);
            s_diag[p1] = nb.a00; s_diag[q1] = nb.a11; s_dyn[o] = nb.a01;
          }
        } else {
          // entries A[x][y], x in {p1, q1}, y in {p2, q2}; a bye (q >= r) has no row / column and an identity rotation
          // (taken block by block: holding all five blocks' entries at once — one trip of LDS latency for the lot — needs
          // more than the 128 registers a 1024-thread workgroup has, and the spills cost more than the latency: 648 -> 899 µs at rank 101)
          const bool h1 = q1 < r, h2 = q2 < r;
          const int opp = p1 < p2 ? at(p1, p2) : at(p2, p1);
          const int opq = h2 ? (p1 < q2 ? at(p1, q2) : at(q2, p1)) : 0;
          const int oqp = h1 ? (q1 < p2 ? at(q1, p2) : at(p2, q1)) : 0;
          const int oqq = (h1 && h2) ? (q1 < q2 ? at(q1, q2) : at(q2, q1)) : 0;
          B22 b{s_dyn[opp], h2 ? s_dyn[opq] : 0.0, h1 ? s_dyn[oqp] : 0.0, (h1 && h2) ? s_dyn[oqq] : 0.0};
          const B22 nb = rot_block(b, r1.x, r1.y, r2.x, r2.y);
          s_dyn[opp] = nb.a00;
          if (h2) s_dyn[opq] = nb.a01;
          if (h1) s_dyn[oqp] = nb.a10;
          if (h1 && h2) s_dyn[oqq] = nb.a11;
        }
      }
      __syncthreads();
      ++n_rounds;
    }
    // convergence (thread = row of the triangle): strict off(A)² <= 1e-26·Σ diag², or loose — every |A_ij| <= 4e-6·|A_jj − A_ii|:
    // the replay workgroups then apply V <- V·(I + X), X_ij = A_ij/(A_jj − A_ii), in place of one more sweep (see k_posterior_eigen_rr)
    double off = 0.0, dg = 0.0;
    bool bad = false;
    if (tid < n) {
      const double dii = s_diag[tid];
      for (int j = tid + 1; j < n; ++j) {
        const double v = s_dyn[at(tid, j)];
        off = fma(2.0 * v, v, off);
        bad = bad || fabs(v) > 4e-6 * fabs(s_diag[j] - dii);
      }
      dg = dii * dii;
    }
    for (int o = 32; o > 0; o >>= 1) { off += __shfl_xor(off, o, 64); dg += __shfl_xor(dg, o, 64); }
    const bool wave_bad = __any(bad);
    if ((tid & 63) == 0) { s_red[tid >> 6] = off; s_red2[tid >> 6] = dg; s_bad[tid >> 6] = wave_bad ? 1 : 0; }
    __syncthreads();
    off = 0.0; dg = 0.0;
    int any_bad = 0;
    for (int w = 0; w < 16; ++w) { off += s_red[w]; dg += s_red2[w]; any_bad |= s_bad[w]; }
    const int strict = off <= 1e-26 * dg;
    converged = strict || (!any_bad && off <= dg && !no_corr);
    use_corr = converged && !strict;
    n_sweeps = sweep + 1;
    const bool last = converged || sweep + 1 >= max_sweeps;
    if (last && use_corr && tid < n) {
      const double dii = s_diag[tid];
      sc1_store(xcorr + (size_t)tid * n + tid, 0.0);
      for (int j = tid + 1; j < n; ++j) {
        const double a = s_dyn[at(tid, j)];
        const double x = a == 0.0 ? 0.0 : a / (s_diag[j] - dii);
        sc1_store(xcorr + (size_t)tid * n + j, x);
        sc1_store(xcorr + (size_t)j * n + tid, -x);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // this sweep's rotations are in the log (plain stores, every storing wave past the barrier above): released, then announced
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      if (last) __hip_atomic_store(meta + 1, use_corr ? launch_id : 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(meta, (launch_id << kPwIdShift) | (last ? kPwFinished : 0) | n_rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
  }
  if (tid < n) mu_out[tid] = s_diag[tid];
  if (tid == 0) { status[0] = converged ? 0 : 2; status[-1] = n_sweeps; }
}

// T = A·V (pass 0) or A' = Vᵀ·T (pass 1): plain one-thread-per-entry products (r <= 200: 8 MFLOP, spread over the chip)
__global__ void __launch_bounds__(256) k_eigen_big_warm(int r, const double* __restrict__ X, const double* __restrict__ V, double* __restrict__ out,
                                                         int pass) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= r * r) return;
  const int i = e / r, j = e - i * r;
  double s = 0.0;
  if (pass == 0) { for (int k = 0; k < r; ++k) s = fma(X[(size_t)i * r + k], V[(size_t)k * r + j], s); }
  else { for (int k = 0; k < r; ++k) s = fma(V[(size_t)k * r + i], X[(size_t)k * r + j], s); }
  out[e] = s;
}

// one wave per index p: rank of its eigenvalue (S descending = mu ascending, ties: lower index first), sign by the
// largest-|.| component (the first among equals), the two output layouts
__global__ void __launch_bounds__(64) k_eigen_big_finish(int r, const double* __restrict__ Vwork, const double* __restrict__ mu,
                                                          double* __restrict__ Vout, double* __restrict__ Vtout, double* __restrict__ Sout,
                                                          const int* __restrict__ status, int* __restrict__ host_status,
                                                          const int* __restrict__ gate, int gate_value) {
  if (gate && ((gate[0] >> kPwIdShift) & kPwIdMask) != gate_value) return;  // (gate = the iteration's progress word: did THIS launch's run?)
  const int p = blockIdx.x, l = threadIdx.x;
  const double mp = mu[p];
  int cnt = 0;
  for (int j = l; j < r; j += 64) { const double mj = mu[j]; cnt += (mj < mp) || (mj == mp && j < p); }
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  const int rank = cnt;
  double bv = -1.0;
  int bi = 0x7fffffff;
  for (int k = l; k < r; k += 64) {
    const double a = fabs(Vwork[(size_t)k * r + p]);
    if (a > bv) { bv = a; bi = k; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  const double sgn = Vwork[(size_t)bi * r + p] < 0.0 ? -1.0 : 1.0;
  for (int k = l; k < r; k += 64) {
    const double v = Vwork[(size_t)k * r + p] * sgn;
    Vout[(size_t)k * r + rank] = v;
    Vtout[(size_t)rank * r + k] = v;
  }
  if (l == 0) Sout[rank] = 1.0 / mp;
  if (p == 0 && l == 0 && host_status) __hip_atomic_store(host_status, status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// ---------------------------------------------------------------- a8 propose
// c_new = (G + σ²I)⁻¹ G w = w − σ² P w with P = (G + σ²I)⁻¹ precomputed;  w = α + D⁻¹ V (√S ∘ z)

template <int NT>
__global__ void __launch_bounds__(NT) k_propose(int r, ProposeIn in, double* __restrict__ c_out, int tpr_log2, const int* relay_in,
                                                int* relay_out) {
  if (relay_in && threadIdx.x < 3) relay_out[threadIdx.x] = relay_in[threadIdx.x];
  propose_body<true>(r, in, c_out, tpr_log2);
}

// ---------------------------------------------------------------- deterministic ICP helpers

__global__ void __launch_bounds__(kBlock) k_gather_points(int K, const double* __restrict__ x, const int* __restrict__ ids,
                                                           double* __restrict__ P) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  const int i = ids[k];
  P[3 * k] = x[3 * i]; P[3 * k + 1] = x[3 * i + 1]; P[3 * k + 2] = x[3 * i + 2];
}

__global__ void __launch_bounds__(kBlock) k_correspond_plain(int K, const int* __restrict__ ids, const double* __restrict__ pts,
                                                              const double* __restrict__ ref, const double* __restrict__ mean, CorrBuffers cb) {
  const int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  const int id = ids[k];
  cb.id[k] = id; cb.aux[k] = -1; cb.keep[k] = 1;
  for (int d = 0; d < 3; ++d) {
    const double p = pts[3 * k + d];
    cb.pt[3 * k + d] = p;
    cb.nhat[3 * k + d] = 0.0;
    cb.e[3 * k + d] = (p - ref[3 * id + d]) - mean[3 * id + d];  // the regression sees world-space points (IcpBasedSurfaceFitting.scala:81)
  }
}

__global__ void __launch_bounds__(256) k_mean_step(int r, const double* __restrict__ alpha, const double* __restrict__ P, double sigma2,
                                                    double step, double* __restrict__ c, int tpr_log2) {
  __shared__ double s_a[512], s_y[512];
  for (int i = threadIdx.x; i < r; i += blockDim.x) s_a[i] = alpha[i];
  __syncthreads();
  block_matvec(r, P, r, s_a, s_y, tpr_log2);
  for (int i = threadIdx.x; i < r; i += blockDim.x) {
    const double cnew = fma(-sigma2, s_y[i], s_a[i]);   // model.coefficients(posterior.mean) (:84)
    c[i] = c[i] + (cnew - c[i]) * step;                 // :85
  }
}

// ---------------------------------------------------------------- posterior variability maps

__global__ void __launch_bounds__(kBlock) k_accumulate(int n, const double* __restrict__ src, double scale_after, double* __restrict__ acc) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  double v = acc[i] + src[i];
  if (scale_after != 0.0) v *= scale_after;
  acc[i] = v;
}

// lane = vertex; the sample loop runs in the reference's order (mean = (Σ s)·(1/n), then the centred second moments)
__global__ void __launch_bounds__(kBlock) k_variability(int N, int S, const double* __restrict__ X, int mode,
                                                         const double* __restrict__ normals, double* __restrict__ out) {
  const int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= N) return;
  const size_t stride = (size_t)3 * N;
  double m0 = 0.0, m1 = 0.0, m2 = 0.0;
  for (int s = 0; s < S; ++s) {
    const double* x = X + (size_t)s * stride + 3 * i;
    m0 += x[0]; m1 += x[1]; m2 += x[2];
  }
  const double inv_n = 1.0 / S, inv_n1 = 1.0 / (S - 1);
  m0 *= inv_n; m1 *= inv_n; m2 *= inv_n;
  if (mode == 0) {
    double c0 = 0.0, c1 = 0.0, c2 = 0.0;
    for (int s = 0; s < S; ++s) {
      const double* x = X + (size_t)s * stride + 3 * i;
      const double v0 = x[0] - m0, v1 = x[1] - m1, v2 = x[2] - m2;
      c0 += v0 * v0; c1 += v1 * v1; c2 += v2 * v2;
    }
    out[i] = (c0 * inv_n1 + c1 * inv_n1) + c2 * inv_n1;   // trace(cov) (:43)
  } else {
    const double n0 = normals[3 * i], n1 = normals[3 * i + 1], n2 = normals[3 * i + 2];
    double acc = 0.0;
    for (int s = 0; s < S; ++s) {
      const double* x = X + (size_t)s * stride + 3 * i;
      const double p = (n0 * (x[0] - m0) + n1 * (x[1] - m1)) + n2 * (x[2] - m2);
      acc += p * p;                                        // :69
    }
    out[i] = acc * inv_n1;
  }
}

// ---------------------------------------------------------------- evaluator reductions

__global__ void __launch_bounds__(kBlock) k_sum_gauss_logpdf(int K, const double* __restrict__ d2, double mean, double sigma,
                                                              double* __restrict__ out) {
  sum_gauss_logpdf_body(K, d2, mean, sigma, out);
}

__global__ void __launch_bounds__(1024) k_dist_stats(int K, const double* __restrict__ d2, const unsigned char* __restrict__ flags,
                                                        const int* __restrict__ idx, int n_flags, double* __restrict__ out) {
  __shared__ double s_red[16];
  double sum = 0.0, mx = -__builtin_inf(), cnt = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    bool drop = false;
    if (flags) {
      int i = idx ? idx[k] : k;
      drop = (i >= 0 && i < n_flags) ? flags[i] != 0 : false;
    }
    if (!drop) {
      double d = sqrt(d2[k]);
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

// the maximum alone (the Hausdorff evaluator needs nothing else of the list): any number of workgroups, the non-negative doubles'
// bit patterns through a 64-bit atomic maximum — order-independent, so exact; `out_max` must be zero (or a distance) beforehand
__global__ void __launch_bounds__(1024) k_dist_max(int K, const double* __restrict__ d2, double* __restrict__ out_max) {
  __shared__ double s_red[16];
  const int k = blockIdx.x * 1024 + threadIdx.x;
  double mx = k < K ? d2[k] : 0.0;
  mx = block_max(mx, s_red);
  if (threadIdx.x == 0) atomicMax((unsigned long long*)out_max, d2bits(sqrt(mx)));
}

}  // namespace

void launch_correspond_model(hipStream_t st, int K, const double* x, const double* cp, const int* nnv,
                             const unsigned char* tgt_boundary, int boundary_aware, const Pose& pose,
                             const double* ref, const double* mean, const int* tris, const int* adj_off,
                             const int* adj, const CorrBuffers& cb, const EntryInit& init) {
  if (K <= 0) return;
  CorrTask c{K, cb, x, nullptr, tgt_boundary, nnv, boundary_aware, pose, ref, mean, tris, adj_off, adj};
  ProfScope _ps(st, KID_CORRESPOND);
  hipLaunchKernelGGL(k_correspond_model, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, c, cp, init);
}

void launch_correspond_target(hipStream_t st, int K, const double* x, const double* tpts, const int* nn_id,
                              const unsigned char* model_boundary, int boundary_aware, const Pose& pose,
                              const double* ref, const double* mean, const int* tris, const int* adj_off,
                              const int* adj, const CorrBuffers& cb, const EntryInit& init) {
  if (K <= 0) return;
  CorrTask c{K, cb, x, tpts, model_boundary, nullptr, boundary_aware, pose, ref, mean, tris, adj_off, adj};
  ProfScope _ps(st, KID_CORRESPOND);
  hipLaunchKernelGGL(k_correspond_target, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, c, nn_id, init);
}

int regression_splits(int K) {
  int s = (K + 7) / 8;  // ~8 correspondences per wave: two gather rounds on its dependent chain, many waves to overlap them
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

int regression_fold(int K, int r, int n_posteriors_in_launch) {
  // (developer switches: ICP_REGRESSION_FOLD_TILES = output tiles from which a launch folds, ICP_REGRESSION_FOLD_K = … or posteriors of at
  // most this many correspondences fold whatever the launch carries)
  static const int min_tiles = dev_env("ICP_REGRESSION_FOLD_TILES") ? std::atoi(dev_env("ICP_REGRESSION_FOLD_TILES")) : 512;
  static const int small_k = dev_env("ICP_REGRESSION_FOLD_K") ? std::atoi(dev_env("ICP_REGRESSION_FOLD_K")) : 0;
  const int S = regression_splits(K);
  if (S <= 1) return 1;
  // femur-size matrices (4 x 4 tiles, 13 leaves) keep their split-K: 64 chains a launch measured 186k it/s folded against 204k split
  // (same box, alternating) — their partials are small, and 16,640 one-leaf waves hide the gathers' latency better than 1,280 thirteen-leaf ones
  static const int min_rank_tiles = dev_env("ICP_REGRESSION_FOLD_RANK_TILES") ? std::atoi(dev_env("ICP_REGRESSION_FOLD_RANK_TILES")) : 6;
  if (((r + 1 + 15) >> 4) < min_rank_tiles && K > small_k) return 1;
  const long tiles = (long)regression_tiles(r) * std::max(n_posteriors_in_launch, 1);
  return (tiles >= min_tiles || K <= small_k) ? S : 1;
}

int regression_macro(int r, int fold) {
  static const int forced = dev_env("ICP_REGRESSION_MACRO") ? std::atoi(dev_env("ICP_REGRESSION_MACRO")) : 0;  // (developer switch: 1, 2)
  if (fold <= 1) return 1;
  const int nt = (r + 1 + 15) >> 4;
  if (forced >= 1 && forced <= 2) return nt >= forced ? forced : 1;
  // (measured, 30 face-model chains a launch — 13 x 13 tiles, K = 400: single tiles 361 µs, 2 x 2 macro tiles 206 µs, 3 x 3 336 µs (few waves,
  // each a long chain of gathers); femur-size matrices (4 x 4 tiles) stay with single tiles: three macro units per posterior are too few waves)
  return nt >= 6 ? 2 : 1;
}
int regression_units(int r, int leaves, int fold, int macro) {
  if (fold > 1 && macro > 1) return regression_macro_tiles(r, macro);
  return regression_tiles(r) * (leaves / std::max(fold, 1));
}

void launch_regression(hipStream_t st, int K, int r, const double* Q, const CorrBuffers& cb, double w_tangent,
                       double kappa, double* Mpart, int* splits_out) {
  const int S = regression_splits(K);
  int kchunk = (K + S - 1) / S;
  if (kchunk < 1) kchunk = 1;
  *splits_out = S;
  { ProfScope _ps(st, KID_REGRESSION);
    hipLaunchKernelGGL(k_regression_mfma, dim3(S, regression_tiles(r)), dim3(64), 0, st, K, kchunk, r, Q, cb, w_tangent, kappa, Mpart); }
}

static void set_dyn_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}
// the same for a kernel whose largest request is known up front: one runtime call per process instead of one per launch
static void set_dyn_lds_once(const void* fn, size_t max_bytes, bool* done) {
  if (*done) return;
  (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)max_bytes);
  *done = true;
}

template <int TPT, int NT>
static void launch_factor_reg(hipStream_t st, int r, int n_post, const FactorArgs& fa) {
  const size_t shmem = sizeof(double) * (size_t)(r + 1) * (r | 1);
  set_dyn_lds((const void*)k_posterior_factor_reg<TPT, NT>, shmem);
  hipLaunchKernelGGL((k_posterior_factor_reg<TPT, NT>), dim3(n_post), dim3(NT), shmem, st, r, fa);
}

void library_release_stream(hipStream_t) {}  // (no library handles any more: kept for the callers' stream teardown)

__global__ void __launch_bounds__(256) k_assemble_posterior_matrix(int r, const double* __restrict__ P, double* __restrict__ M) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= r * r) return;
  const int i = e / r, j = e - i * r;
  const int hi = max(i, j), lo = min(i, j);
  M[e] = P[(size_t)hi * (r + 1) + lo] + (i == j ? 1.0 : 0.0);  // (the values the factor kernels' own assembly writes)
}

void launch_sum_partials(hipStream_t st, int r, double* Mpart, int splits) {
  if (splits <= 1) return;
  PartialSumArgs ra{};
  ra.nn = (r + 1) * (r + 1);
  ra.n = 1;
  ra.Mpart[0] = Mpart;
  ra.splits[0] = splits;
  hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(ra.nn, 256), 1), dim3(256), 0, st, ra);
}
void launch_sum_partials_many(hipStream_t st, int r, int n, double* const* Mpart, const int* splits) {
  PartialSumArgs ra{};
  ra.nn = (r + 1) * (r + 1);
  for (int i = 0; i < n && i < kFactorMax; ++i)
    if (splits[i] > 1) { ra.Mpart[ra.n] = Mpart[i]; ra.splits[ra.n] = splits[i]; ++ra.n; }
  if (ra.n) hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(ra.nn, 256), ra.n), dim3(256), 0, st, ra);
}
void launch_assemble_posterior_matrix(hipStream_t st, int r, const double* Mpart_summed, double* M) {
  hipLaunchKernelGGL(k_assemble_posterior_matrix, dim3(cdiv(r * r, 256)), dim3(256), 0, st, r, Mpart_summed, M);
}

int posterior_factor_max() { return kFactorMax; }

void launch_posterior_factor(hipStream_t st, int r, int n_post, const PosteriorFactorIO* io) {
  FactorArgs fa{};
  for (int p = 0; p < n_post; ++p) {
    fa.Mpart[p] = io[p].Mpart; fa.splits[p] = io[p].splits; fa.M[p] = io[p].M; fa.alpha[p] = io[p].alpha;
    fa.status[p] = io[p].status; fa.scratch[p] = io[p].scratch;
    fa.Lout[p] = io[p].Lout; fa.Sout[p] = io[p].Sout;
  }
  const int ld = r | 1;
  const size_t tiles = (size_t)factor_tile_count(r);
  const bool w_fits = (size_t)(r + 1) * ld <= (size_t)kLdsDoubles - 2000;  // + the static LDS of the kernel
  const bool blocked = !(w_fits && tiles <= 2048) && r <= kCholMaxRank;
  if (blocked) {
    // the split-K partials are summed by a launch of their own, on many CUs, into the first one (same order of summation as the
    // kernel's own loop, which one workgroup's share of the memory system made 44 us of at rank 200)
    for (int p0 = 0; p0 < n_post; p0 += kFactorMax) {
      PartialSumArgs ra{};
      ra.nn = (r + 1) * (r + 1);
      for (int p = p0; p < std::min(n_post, p0 + kFactorMax); ++p)
        if (fa.splits[p] > 1) {
          ra.Mpart[ra.n] = const_cast<double*>(fa.Mpart[p]);
          ra.splits[ra.n] = fa.splits[p];
          ++ra.n;
          fa.splits[p] = 1;
        }
      if (ra.n) hipLaunchKernelGGL(k_sum_partials, dim3(cdiv(ra.nn, 256), ra.n), dim3(256), 0, st, ra);
    }
  }
  ProfScope _ps(st, KID_FACTOR);
  if (w_fits && tiles <= 256) launch_factor_reg<1, 256>(st, r, n_post, fa);
  else if (w_fits && tiles <= 1024) launch_factor_reg<1, 1024>(st, r, n_post, fa);
  else if (w_fits && tiles <= 2048) launch_factor_reg<2, 1024>(st, r, n_post, fa);
  else if (tiles <= 4096 && !dev_env("ICP_FACTOR_BLOCKED")) {  // ranks 117..~250: register tiles, the factor to global scratch
    const size_t shmem = sizeof(double) * (size_t)kCholNB * (kCholNB + 1);
    if (tiles <= 3072) {
      set_dyn_lds((const void*)k_posterior_factor_tiles<3>, shmem);
      hipLaunchKernelGGL(k_posterior_factor_tiles<3>, dim3(n_post, 1 + kFactorAsmGroups), dim3(1024), shmem, st, r, fa);
    } else {
      set_dyn_lds((const void*)k_posterior_factor_tiles<4>, shmem);
      hipLaunchKernelGGL(k_posterior_factor_tiles<4>, dim3(n_post, 1 + kFactorAsmGroups), dim3(1024), shmem, st, r, fa);
    }
  } else if (r <= kCholMaxRank && sizeof(double) * ((size_t)kCholNB * (kCholNB + 1) + (size_t)(r + 1) * (kCholNB + 1)) <= (size_t)147 * 1024) {
    // blocked, matrix in `scratch`, block columns through LDS — while the panel fits beside the kernel's 13 KB of static arrays (ranks up
    // to 224; round 6: ranks 253..256, past the register tiles' 4,096, used to come here with 166 KB of dynamic LDS — a launch that
    // fails, M never written —: they take the generic kernel below)
    const size_t shmem = sizeof(double) * ((size_t)kCholNB * (kCholNB + 1) + (size_t)(r + 1) * (kCholNB + 1));
    static size_t lds_granted = 0;
    if (shmem > lds_granted) { set_dyn_lds((const void*)k_posterior_factor_blocked, shmem); lds_granted = shmem; }
    hipLaunchKernelGGL(k_posterior_factor_blocked, dim3(n_post), dim3(1024), shmem, st, r, fa);
  } else {
    const int use_lds = (size_t)(r + 1) * ld <= (size_t)kLdsDoubles;
    const size_t shmem = use_lds ? sizeof(double) * (size_t)(r + 1) * ld : 0;
    set_dyn_lds((const void*)k_posterior_factor_generic, shmem);
    hipLaunchKernelGGL(k_posterior_factor_generic, dim3(n_post), dim3(kFactorThreads), shmem, st, r, fa, use_lds);
  }
}

void launch_transition_tails(hipStream_t st, int r, int n, const TransitionTailIO* io, const double* Ginv, double sigma2) {
  TailArgs ta{};
  ta.n = n;
  for (int t = 0; t < n; ++t) {
    ta.alpha[t] = io[t].alpha; ta.M[t] = io[t].M; ta.c_from[t] = io[t].c_from; ta.c_to[t] = io[t].c_to;
    ta.step[t] = io[t].step; ta.out[t] = io[t].out; ta.status[t] = io[t].status;
    ta.relay_in[t] = io[t].relay_in; ta.relay_out[t] = io[t].relay_out;
  }
  const int ld = r | 1;
  const size_t one = (size_t)r * ld;
  const int n_lds = 2 * one <= (size_t)kLdsDoubles - 2560 ? 2 : (one <= (size_t)kLdsDoubles - 2560 ? 1 : 0);
  const size_t shmem = sizeof(double) * one * n_lds;
  ProfScope _ps(st, KID_TAIL);
  if (n_lds == 0) {
    // neither matrix fits LDS (ranks above 134): every product of the iteration streams 8·r² bytes from L2, 1024 threads keep four
    // times the loads in flight (50 -> ≈ 20 µs at rank 200).  Below, 256 threads: the arithmetic of the merged step's own tails.
    hipLaunchKernelGGL(k_transition_tails<1024>, dim3(n), dim3(1024), 0, st, r, ta, Ginv, sigma2, 0, 4);  // (16 lanes per row: 128-byte segments)
  } else {
    set_dyn_lds((const void*)k_transition_tails<256>, shmem);
    hipLaunchKernelGGL(k_transition_tails<256>, dim3(n), dim3(256), shmem, st, r, ta, Ginv, sigma2, n_lds, matvec_tpr_log2(r, 256));
  }
}

void launch_transition_tail_direct(hipStream_t st, int r, const TransitionTailIO& io, const double* G, double sigma2, double* work) {
  const int ld = r | 1;
  const int use_lds = (size_t)(r + 1) * ld <= (size_t)kLdsDoubles - 1200;
  const size_t shmem = use_lds ? sizeof(double) * (size_t)(r + 1) * ld : 0;
  set_dyn_lds((const void*)k_transition_tail_direct, shmem);
  { ProfScope _ps(st, KID_TAIL);
    hipLaunchKernelGGL(k_transition_tail_direct, dim3(1), dim3(1024), shmem, st, r, io.alpha, io.M, G, sigma2, io.c_from, io.c_to,
                       io.step, work, io.out, io.status, use_lds); }
}

static size_t jacobi_work_doubles(int r) {
  const size_t n2 = ((size_t)r + 1) & ~(size_t)1;
  if (r > 64) {  // in-place Jacobi (k_eigen_big): A0 | T | Vwork | mu | rotation log of every sweep | meta — or the generic kernel's r×r scratch
    const size_t log = (size_t)kEigenMaxSweeps * (n2 - 1) * n2;  // (c, s) per pair and round
    return 3 * (size_t)r * r + n2 + log + 64;
  }
  const size_t log = ((size_t)kEigenMaxSweeps * (n2 - 1) + 2) * n2;  // 2 doubles per pair and round
  return log + n2 * 64 + 128 + 256;  // fixed-position variant: log + correction + meta (see launch_eigen_rr)
}
// the tridiagonal route's part of `work`, behind the Jacobi kernels': d | e | beta | mu | sync words | reflectors
// (round 6: 208 — the reference's own largest model, femur_gp_model_200-components.h5, has 201 components: apps/femur/CreateGPModel.scala:93)
constexpr int kTriMaxRank = 256;  // = tri::kTriMaxN: four row slots of 64
static size_t tri_work_doubles(int r) { return r <= kTriMaxRank ? 4 * (size_t)tri::kTriMaxN + 8 + (size_t)r * 256 : 0; }
size_t eigen_work_doubles(int r) { return jacobi_work_doubles(r) + tri_work_doubles(r); }  // `work` of launch_posterior_eigen

// Householder tridiagonalisation on one workgroup, then one wave per eigenpair (icp_tridiag.hpp)
static bool tridiag_route(int r) {
  static const int forced = dev_env("ICP_EIGEN_TRIDIAG") ? std::atoi(dev_env("ICP_EIGEN_TRIDIAG")) : -1;
  if (r < 3 || r > kTriMaxRank) return false;
  return forced >= 0 ? forced != 0 : r > 64;
}
static void launch_eigen_big(hipStream_t st, int r, const double* M, const double* sqrt_lambda, const double* Vwarm, double* V, double* Vt,
                             double* S, double* work, int* status, int* host_status, const int* gate);
static void launch_eigen_tridiag(hipStream_t st, int r, const double* M, const double* sqrt_lambda, double* V, double* Vt, double* S, double* work,
                                 int* status, int* host_status, int* done_word, int done_value, int part = 0) {
  double* base = work + jacobi_work_doubles(r);
  double *d = base, *e = base + tri::kTriMaxN, *beta = base + 2 * tri::kTriMaxN, *mu = base + 3 * tri::kTriMaxN;
  int* sync = (int*)(base + 4 * tri::kTriMaxN);
  double* Hv = base + 4 * tri::kTriMaxN + 8;
  // the refinement step's matrices live where the Jacobi kernels' log would be (this route replaces them): N | X | Xt | T | S | R
  const size_t rr = (size_t)r * r;
  double *Nm = work, *X = work + rr, *Xt = work + 2 * rr, *T = work + 3 * rr, *Sm = work + 4 * rr, *R = work + 5 * rr;
  tri::TridiagIO ti{r, M, sqrt_lambda, d, e, beta, Hv, Nm};
  // (the reflector blocks' T factors live where the refinement's R will be: written behind the solve)
  tri::TriSolveIO so{r, d, e, beta, Hv, X, Xt, S, mu, R, sync, status, nullptr, nullptr, 0};
  const tri::TriWyIO wyio{r, beta, Hv, R, sync};
  const int nwg = (r + 3) / 4, nwy = (r - 2 + tri::kWyBlock - 1) / tri::kWyBlock;
  if (part != 2) {  // the reduction
    if (r <= 64) hipLaunchKernelGGL((tri::k_tridiag<4, 1, 16, 0>), dim3(1), dim3(256), 0, st, ti);
    else if (r <= 128) hipLaunchKernelGGL((tri::k_tridiag<4, 2, 32, 0>), dim3(1), dim3(256), 0, st, ti);
    else if (r <= 192) hipLaunchKernelGGL((tri::k_tridiag<8, 3, 24, 0>), dim3(1), dim3(512), 0, st, ti);
    else if (r <= 200) hipLaunchKernelGGL((tri::k_tridiag<8, 4, 25, 7>), dim3(1), dim3(512), 0, st, ti);
    else if (r <= 208) hipLaunchKernelGGL((tri::k_tridiag<8, 4, 26, 6>), dim3(1), dim3(512), 0, st, ti);
    else hipLaunchKernelGGL((tri::k_tridiag<8, 4, 32, 0>), dim3(1), dim3(512), 0, st, ti);
  }
  if (part == 1) return;
  // (ranks above 64: the T factors by the solve launch's own trailing workgroups — tri_solve_or_wy)
  if (r <= 64) hipLaunchKernelGGL(tri::k_tri_wy<1>, dim3(nwy), dim3(64), 0, st, wyio, wyio);
  if (r <= 64) hipLaunchKernelGGL(tri::k_tri_solve<1>, dim3(nwg), dim3(256), tri::tri_solve_lds_bytes(r), st, so, so);
  else if (r <= 128) hipLaunchKernelGGL(tri::k_tri_solve<2>, dim3(nwg + nwy), dim3(256), tri::tri_solve_lds_bytes(r), st, so, so);
  else if (r <= 192) hipLaunchKernelGGL(tri::k_tri_solve<3>, dim3(nwg + nwy), dim3(256), tri::tri_solve_lds_bytes(r), st, so, so);
  else {
    static bool lds_set = false;  // (above rank 201 the launch's dynamic LDS passes 48 KiB: 61 KiB at rank 256)
    set_dyn_lds_once((const void*)tri::k_tri_solve<4>, tri::tri_solve_lds_bytes(tri::kTriMaxN), &lds_set);
    hipLaunchKernelGGL(tri::k_tri_solve<4>, dim3(nwg + nwy), dim3(256), tri::tri_solve_lds_bytes(r), st, so, so);
  }
  if (r > 64) {  // (ranks above 64: the back-transformation is a launch of its own, sixteen eigenvectors per wave on the matrix cores)
    const tri::TriBackIO bk{r, Hv, R, X, Xt, status, sync};
    const int nb16 = (r + 15) / 16;
    if (r <= 128) hipLaunchKernelGGL(tri::k_tri_back<2>, dim3(nb16), dim3(256), 0, st, bk, bk);
    else if (r <= 192) hipLaunchKernelGGL(tri::k_tri_back<3>, dim3(nb16), dim3(256), 0, st, bk, bk);
    else hipLaunchKernelGGL(tri::k_tri_back<4>, dim3(nb16), dim3(256), 0, st, bk, bk);
  }
  // one refinement step: T = N·X and R = I − XᵀX, S = XᵀT, E, then V = X + X·E (and Vt)
  const int nt = (r + 15) / 16;
  const int* skip = sync + 3;  // (written by the solve launch: 1 = every gap wide enough, the refinement's launches return at once)
  const tri::TriGemm gT{Nm, X, T, 0, nullptr, nullptr, skip}, gR{X, X, R, 1, nullptr, nullptr, skip}, gS{X, T, Sm, 0, nullptr, nullptr, skip};
  hipLaunchKernelGGL(tri::k_tri_gemm, dim3(nt, nt, 2), dim3(64), 0, st, r, gT, gR);
  hipLaunchKernelGGL(tri::k_tri_gemm, dim3(nt, nt, 1), dim3(64), 0, st, r, gS, gS);
  hipLaunchKernelGGL(tri::k_tri_correction, dim3((unsigned)((rr + 255) / 256)), dim3(256), 0, st, r, (const double*)Sm, (const double*)R, T, S, skip);
  const tri::TriGemm gV{Xt, T, V, 2, X, Vt, skip};
  hipLaunchKernelGGL(tri::k_tri_gemm, dim3(nt, nt, 1), dim3(64), 0, st, r, gV, gV);
  // eigenvalues that multisection could not tell apart (status 2: a spectrum with (near-)multiple eigenvalues, e.g. a posterior without
  // correspondences over a model with equal variances): the Jacobi iteration takes over, cold, in the same stream — its launches
  // return at once otherwise
  if (r > 64 && r <= kBigMaxRank) launch_eigen_big(st, r, M, sqrt_lambda, nullptr, V, Vt, S, work, status, nullptr, status);
  else if (r > kBigMaxRank)  // (matrix behind L2: `work`'s head, whose refinement matrices a failed multisection has no use for)
    hipLaunchKernelGGL(k_posterior_eigen, dim3(1), dim3(1024), 0, st, r, M, sqrt_lambda, (const double*)nullptr, V, Vt, S, work, status, 0, 0,
                       (const int*)status);
  if (host_status || done_word) hipLaunchKernelGGL(tri::k_tri_done, dim3(1), dim3(1), 0, st, (const int*)status, host_status, done_word, done_value);
}

// The same route for n decompositions side by side (the chains of a wide step): every launch of the sequence takes all of them —
// the one-workgroup reductions run on n CUs at once.  No gated Jacobi fall-back in the sequence (see icp_kernels.hpp).
bool eigen_tridiag_many_supported(int r) { return r > 64 && r <= kTriMaxRank; }
void launch_posterior_eigen_tridiag_many(hipStream_t st, int r, int n_all, const EigenRequest* rq_all, const double* const* parts_all,
                                         const int* skip_all, int part) {
  const size_t rr = (size_t)r * r;
  const int nwg = (r + 3) / 4, nt = (r + 15) / 16;
  for (int q0 = 0; q0 < n_all; q0 += tri::kTriMany) {
    const int n = std::min(tri::kTriMany, n_all - q0);
    const EigenRequest* rq = rq_all + q0;
    const int* skip = skip_all ? skip_all + q0 : nullptr;
    tri::TridiagMany tm{};
    tri::TriSolveMany sm{};
    tri::TriBackMany bm{};
    tri::TriGemmMany g1{}, g2{}, g3{};
    tri::TriCorrMany cm{};
    tri::TriDoneMany dm{};
    tri::AssembleMany am{};
    bool assemble = false;
    for (int q = 0; q < n; ++q) {
      double* work = rq[q].work;
      double* base = work + jacobi_work_doubles(r);
      double *d = base, *e = base + tri::kTriMaxN, *beta = base + 2 * tri::kTriMaxN, *mu = base + 3 * tri::kTriMaxN;
      int* sync = (int*)(base + 4 * tri::kTriMaxN);
      double* Hv = base + 4 * tri::kTriMaxN + 8;
      double *Nm = work, *X = work + rr, *Xt = work + 2 * rr, *T = work + 3 * rr, *Sm = work + 4 * rr, *R = work + 5 * rr;
      const double* sl = rq[q].sqrt_lambda;
      tm.p[q] = tri::TridiagIO{r, rq[q].M, sl, d, e, beta, Hv, Nm};
      sm.p[q] = tri::TriSolveIO{r, d, e, beta, Hv, X, Xt, rq[q].S, mu, R, sync, rq[q].status, nullptr, nullptr, 0};
      bm.p[q] = tri::TriBackIO{r, Hv, R, X, Xt, rq[q].status, sync};
      // (test-hooks build, ICP_TEST_TRI_REFINE_ALWAYS=1: the refinement step whatever the gaps are — it is the rare path otherwise)
      static const bool refine_always = dev_env("ICP_TEST_TRI_REFINE_ALWAYS") && std::atoi(dev_env("ICP_TEST_TRI_REFINE_ALWAYS")) != 0;
      const int* skip = refine_always ? nullptr : sync + 3;
      g1.g[2 * q] = tri::TriGemm{Nm, X, T, 0, nullptr, nullptr, skip};
      g1.g[2 * q + 1] = tri::TriGemm{X, X, R, 1, nullptr, nullptr, skip};
      g2.g[q] = tri::TriGemm{X, T, Sm, 0, nullptr, nullptr, skip};
      cm.S[q] = Sm; cm.R[q] = R; cm.E[q] = T; cm.Sout[q] = rq[q].S; cm.skip[q] = skip;
      g3.g[q] = tri::TriGemm{Xt, T, rq[q].V, 2, X, rq[q].Vt, skip};
      dm.status[q] = rq[q].status; dm.host_status[q] = rq[q].host_status; dm.done_word[q] = rq[q].done_word; dm.done_value[q] = rq[q].done_value;
      am.P[q] = parts_all ? parts_all[q0 + q] : nullptr;
      am.M[q] = const_cast<double*>(rq[q].M);
      assemble = assemble || am.P[q] != nullptr;
    }
    ProfScope _ps(st, KID_EIGEN);
    if (part != 2) {  // the reduction (part 1 of a split sequence: the long one-workgroup launch, before anybody knows whom to skip)
      if (assemble) hipLaunchKernelGGL(tri::k_assemble_many, dim3((unsigned)((rr + 255) / 256), n), dim3(256), 0, st, r, am, skip);
      if (r <= 128) hipLaunchKernelGGL((tri::k_tridiag_many<4, 2, 32, 0>), dim3(n), dim3(256), 0, st, tm, skip);
      else if (r <= 192) hipLaunchKernelGGL((tri::k_tridiag_many<8, 3, 24, 0>), dim3(n), dim3(512), 0, st, tm, skip);
      else if (r <= 200) hipLaunchKernelGGL((tri::k_tridiag_many<8, 4, 25, 7>), dim3(n), dim3(512), 0, st, tm, skip);
      else if (r <= 208) hipLaunchKernelGGL((tri::k_tridiag_many<8, 4, 26, 6>), dim3(n), dim3(512), 0, st, tm, skip);
      else hipLaunchKernelGGL((tri::k_tridiag_many<8, 4, 32, 0>), dim3(n), dim3(512), 0, st, tm, skip);
    }
    if (part == 1) continue;
    const int nwy = (r - 2 + tri::kWyBlock - 1) / tri::kWyBlock;
    // (the reflector blocks' T factors: the solve launch's trailing workgroups — tri_solve_or_wy)
    if (r <= 128) hipLaunchKernelGGL(tri::k_tri_solve_many<2>, dim3(nwg + nwy, n), dim3(256), tri::tri_solve_lds_bytes(r), st, sm, skip);
    else if (r <= 192) hipLaunchKernelGGL(tri::k_tri_solve_many<3>, dim3(nwg + nwy, n), dim3(256), tri::tri_solve_lds_bytes(r), st, sm, skip);
    else {
      static bool lds_set = false;
      set_dyn_lds_once((const void*)tri::k_tri_solve_many<4>, tri::tri_solve_lds_bytes(tri::kTriMaxN), &lds_set);
      hipLaunchKernelGGL(tri::k_tri_solve_many<4>, dim3(nwg + nwy, n), dim3(256), tri::tri_solve_lds_bytes(r), st, sm, skip);
    }
    if (r <= 128) hipLaunchKernelGGL(tri::k_tri_back_many<2>, dim3(nt, n), dim3(256), 0, st, bm, skip);
    else if (r <= 192) hipLaunchKernelGGL(tri::k_tri_back_many<3>, dim3(nt, n), dim3(256), 0, st, bm, skip);
    else hipLaunchKernelGGL(tri::k_tri_back_many<4>, dim3(nt, n), dim3(256), 0, st, bm, skip);
    hipLaunchKernelGGL(tri::k_tri_gemm_many, dim3(nt, nt, 2 * n), dim3(64), 0, st, r, g1, skip, 2);
    hipLaunchKernelGGL(tri::k_tri_gemm_many, dim3(nt, nt, n), dim3(64), 0, st, r, g2, skip, 1);
    hipLaunchKernelGGL(tri::k_tri_correction_many, dim3((unsigned)((rr + 255) / 256), n), dim3(256), 0, st, r, cm, skip);
    hipLaunchKernelGGL(tri::k_tri_gemm_many, dim3(nt, nt, n), dim3(64), 0, st, r, g3, skip, 1);
    bool any_done = false;  // (nobody to tell — the on-device loop reads the status words on the device —: no launch)
    for (int q = 0; q < n; ++q) any_done = any_done || dm.host_status[q] || dm.done_word[q];
    if (any_done) hipLaunchKernelGGL(tri::k_tri_done_many, dim3(n), dim3(1), 0, st, dm, skip);
  }
}

// N = D⁻¹ M D⁻¹ (symmetrised) for the in-place kernel
__global__ void __launch_bounds__(256) k_eigen_big_prepare(int r, const double* __restrict__ M, const double* __restrict__ sqrt_lambda,
                                                           double* __restrict__ A, const int* __restrict__ gate) {
  if (gate && gate[0] != 2) return;
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= r * r) return;
  const int i = e / r, j = e - i * r;
  A[e] = 0.5 * (M[(size_t)i * r + j] + M[(size_t)j * r + i]) / (sqrt_lambda[i] * sqrt_lambda[j]);
}

static void launch_eigen_big(hipStream_t st, int r, const double* M, const double* sqrt_lambda, const double* Vwarm, double* V, double* Vt,
                             double* S, double* work, int* status, int* host_status, const int* gate) {
  const size_t n2 = ((size_t)r + 1) & ~(size_t)1, rr = (size_t)r * r;
  double* A0 = work;
  double* T = work + rr;
  double* Vwork = work + 2 * rr;
  double* mu = work + 3 * rr;
  double* rotlog = mu + n2;
  int* meta = (int*)(rotlog + (size_t)kEigenMaxSweeps * (n2 - 1) * n2);
  const int eb = (int)((rr + 255) / 256);
  // (as the tridiagonal route's fall-back — `gate` — the matrix is in place already: k_tridiag has written N = D⁻¹MD⁻¹, entry by
  // entry the values of the launch below, to the head of `work` for its refinement step, which only reads it)
  if (!gate) hipLaunchKernelGGL(k_eigen_big_prepare, dim3(eb), dim3(256), 0, st, r, M, sqrt_lambda, A0, gate);
  if (Vwarm) {  // A0 <- Vwarmᵀ·A0·Vwarm
    hipLaunchKernelGGL(k_eigen_big_warm, dim3(eb), dim3(256), 0, st, r, (const double*)A0, Vwarm, T, 0);
    hipLaunchKernelGGL(k_eigen_big_warm, dim3(eb), dim3(256), 0, st, r, (const double*)T, Vwarm, A0, 1);
  }
  const size_t half = n2 / 2;
  const bool square = r <= 140;  // the full n × (n|1) image fits one CU's LDS
  const size_t lds_iter = sizeof(double) * ((square ? (size_t)r * (r | 1) : (size_t)r * (r - 1) / 2) + r + 2 * half);
  const size_t lds_replay = sizeof(double) * ((size_t)kBigSlabRows * r + std::max(2 * (size_t)kBigStageRounds * half, (size_t)16 * r));
  const size_t shmem = std::max(lds_iter, lds_replay);
  static size_t lds_granted[2] = {0, 0};
  if (shmem > lds_granted[square]) {
    if (square) set_dyn_lds((const void*)k_eigen_big<true>, shmem); else set_dyn_lds((const void*)k_eigen_big<false>, shmem);
    lds_granted[square] = shmem;
  }
  static std::atomic<int> launch_counter{0};
  const int launch_id = 1 + (int)((unsigned)(++launch_counter) % kPwIdMask);
  static const int sweeps_cap = dev_env("ICP_EIGEN_MAX_SWEEPS") ? std::atoi(dev_env("ICP_EIGEN_MAX_SWEEPS")) : kEigenMaxSweeps;
  static const int no_corr = dev_env("ICP_EIGEN_NO_CORRECTION") != nullptr;
  const int nb = (r + kBigSlabRows - 1) / kBigSlabRows;
  double* xcorr = T;  // (the warm transform's scratch is free once the iteration starts)
  if (square)
    hipLaunchKernelGGL(k_eigen_big<true>, dim3(1 + nb), dim3(1024), shmem, st, r, (const double*)A0, Vwarm, Vwork, mu, rotlog, xcorr, meta,
                       std::min(sweeps_cap, kEigenMaxSweeps), no_corr, launch_id, status, gate);
  else
    hipLaunchKernelGGL(k_eigen_big<false>, dim3(1 + nb), dim3(1024), shmem, st, r, (const double*)A0, Vwarm, Vwork, mu, rotlog, xcorr, meta,
                       std::min(sweeps_cap, kEigenMaxSweeps), no_corr, launch_id, status, gate);
  // (as a fall-back the sort runs only if the iteration did: its progress word carries this launch's id then)
  hipLaunchKernelGGL(k_eigen_big_finish, dim3(r), dim3(64), 0, st, r, (const double*)Vwork, (const double*)mu, V, Vt, S, (const int*)status,
                     host_status, gate ? (const int*)meta : nullptr, launch_id);
}

void eigen_debug_dump(const double* work, int r) {  // developer aid: convergence trace of the last decomposition on `work`
  const size_t n2 = ((size_t)r + 1) & ~(size_t)1;
  const size_t log_doubles = ((size_t)kEigenMaxSweeps * (n2 - 1) + 2) * n2;
  double tr[8];
  (void)hipMemcpy(tr, (const char*)(work + log_doubles + n2 * 64) + 80 * sizeof(int), sizeof(tr), hipMemcpyDeviceToHost);
  std::fprintf(stderr, "[icp eigen] off^2/diag^2 after sweeps 1..: %.2e %.2e %.2e %.2e %.2e\n", tr[0], tr[1], tr[2], tr[3], tr[4]);
  int hist[16];
  (void)hipMemcpy(hist, (const char*)(work + log_doubles + n2 * 64) + 100 * sizeof(int), sizeof(hist), hipMemcpyDeviceToHost);
  std::fprintf(stderr, "[icp eigen] decompositions by sweep count 1..8: %d %d %d %d %d %d %d %d\n", hist[1], hist[2], hist[3], hist[4], hist[5], hist[6], hist[7], hist[8]);
}

bool eigen_speculation_supported(int r) { return r >= 3 && r <= 64 && dev_env("ICP_EIGEN_GENERIC") == nullptr; }

namespace {
// workgroups per problem: the iteration + the replay (32 rows each)
inline int eigen_rr_per(int r) { return 1 + (r + kReplayRows - 1) / kReplayRows; }
inline EigenProblem eigen_rr_problem(int r, const EigenRequest& rq) {
  const int n2 = (r + 1) & ~1;
  // work = [rotation log | sign exchange | meta: progress word, counters, rank per position]
  const size_t log_doubles = ((size_t)kEigenMaxSweeps * (n2 - 1) + 2) * n2;
  static std::atomic<int> launch_counter{0};  // (any value the previous launch on this `work` did not use would do)
  double* vpos = rq.work + log_doubles;
  const int launch_id = 1 + (int)((unsigned)(++launch_counter) % kPwIdMask);  // never 0: the idle value of the progress word
  return EigenProblem{rq.M, rq.Vwarm, rq.V, rq.Vt, rq.S, rq.status, rq.work, (int*)(vpos + (size_t)n2 * 64), vpos,
                      rq.spec ? *rq.spec : EigenSpec{0, nullptr, 0, nullptr, 0, nullptr}, launch_id, rq.host_status, rq.done_word,
                      rq.done_value, rq.sqrt_lambda};
}
template <class Batch>
void launch_eigen_rr_batch(hipStream_t st, int r, const double* sqrt_lambda, int n, const Batch& batch) {
  // fixed-position variant: A, V and the rotation table double-buffered in LDS
  const int n2 = (r + 1) & ~1;
  const int ldk = 66;  // ldk: 64 coordinates per position row, rows 16 B apart modulo the 256-B bank window
  const size_t szV = (size_t)n2 * ldk;
  const size_t shmem = sizeof(double) * ((size_t)kRrOV + 2 * szV);
  static const int sweeps_cap = dev_env("ICP_EIGEN_MAX_SWEEPS") ? std::atoi(dev_env("ICP_EIGEN_MAX_SWEEPS")) : kEigenMaxSweeps;
  static bool lds_set = false;
  set_dyn_lds_once((const void*)k_posterior_eigen_rr<Batch>, sizeof(double) * ((size_t)kRrOV + 2 * 64 * 66), &lds_set);
  ProfScope _ps(st, KID_EIGEN);
  // per problem: workgroup 0 iterates, workgroup 1 replays its rotations on V as the sweeps are published
  static const int no_corr = dev_env("ICP_EIGEN_NO_CORRECTION") != nullptr;
  hipLaunchKernelGGL(k_posterior_eigen_rr<Batch>, dim3(n * eigen_rr_per(r)), dim3(1024), shmem, st, r, sqrt_lambda, ldk,
                     std::min(sweeps_cap, kEigenMaxSweeps), no_corr, batch);
}
template <int CAP>
void launch_eigen_rr(hipStream_t st, int r, const double* sqrt_lambda, int n, const EigenRequest* rq) {
  EigenBatch<CAP> batch{};
  batch.n = n;
  for (int i = 0; i < n; ++i) batch.p[i] = eigen_rr_problem(r, rq[i]);
  launch_eigen_rr_batch(st, r, sqrt_lambda, n, batch);
}
}  // namespace

// ---------------------------------------------------------------- opt-in: the Cholesky-root sampler (ranks <= 64)
// The reference draws posterior.sample() in the eigenbasis of the posterior covariance (D M⁻¹ D = V S Vᵀ: the numbers z multiply
// the columns of V√S) — that is what the kernels above are for, and what parity with the reference needs.  ANY square root W of
// D M⁻¹ D gives a sample of the same distribution, and the transition density does not depend on the root: with
// M = L Lᵀ, W = D L⁻ᵀ, i.e. D⁻¹ W z = L⁻ᵀ z — ONE back substitution per proposal (propose_body), no decomposition, no inverse.
// This kernel writes the factor where the decomposition would write its basis — V := L (lower triangular, row-major), S := 1/diag(L)
// — with the same completion protocol and the same front end for a launch enqueued ahead of its input, so the machinery around
// it (speculation, completion words, batches) is unchanged.  One workgroup per posterior; the factorisation is the chain step's own
// (factor_reg_body: 2×4 register tiles, one barrier per column).  No iteration, no warm start, no state from one posterior to the
// next: ≈ 15 µs at rank 51 against 70-110 µs for the warm-started decomposition.
// (icp_proposal_set_sampler; NOT the default: the chain it produces is a different realisation of the same Markov kernel.)
struct RootBatch2 {
  int n; EigenProblem p[2];
  __device__ __forceinline__ void announce() const {}
  __device__ __forceinline__ bool skipped(int) const { return false; }
};
typedef EigenBatchMem RootBatchMem;

template <class Batch, int NT>
__global__ void __launch_bounds__(NT) k_posterior_root(int r, Batch batch) {
  batch.announce();
  if (batch.skipped(blockIdx.x)) return;
  const EigenProblem pb = batch.p[blockIdx.x];
  __shared__ int s_cancel;
  const int tid = threadIdx.x;
  const EigenSpec spec = pb.spec;
  EIG_STAMP(0);
  if (tid == 0) s_cancel = 0;
  __syncthreads();
  if (tid == NT - 1) {  // (the protocol of k_posterior_eigen_rr: wait for the input, or for the cancellation, or give up after 5 ms)
    if (spec.ready) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {
        if (__hip_atomic_load(spec.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - spec.ready_seq >= 0) break;
        if (spec.cancel && __hip_atomic_load(spec.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == spec.seq) { s_cancel = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 500000) { s_cancel = 2; break; }
        __builtin_amdgcn_s_sleep(32);
      }
      if (spec.wait_ticks) atomicAdd((unsigned long long*)spec.wait_ticks, (unsigned long long)(__builtin_amdgcn_s_memrealtime() - t0));
    } else if (spec.cancel) {
      if (__hip_atomic_load(spec.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == spec.seq) s_cancel = 1;
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (acquire side for the plain loads of the partials below)
  if (s_cancel) {
    if (tid == 0) {
      if (s_cancel == 2) {
        if (pb.host_status) __hip_atomic_store(pb.host_status, kEigenGaveUp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        pb.Vout[0] = __builtin_nan("");
      }
      __threadfence();
      if (pb.done_word) __hip_atomic_store(pb.done_word, pb.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  EIG_STAMP(1);
  // the factor kernel's own body; its by-products (assembled M, α) go to this problem's scratch (`rotlog` = the proposal's work buffer)
  double* scratch = pb.rotlog;
  const bool ok = factor_reg_body<1, NT>(r, pb.M, spec.splits, scratch, scratch + (size_t)r * r, (int*)(scratch + (size_t)r * r + r),
                                         -1, nullptr, spec.splits > 0 ? nullptr : pb.M, false);
  EIG_STAMP(3);
  // ---- V := L = L̃·D̃^{1/2} (L_ik = w_ik / sqrt(d_k), L_kk = sqrt(d_k)), Vt := Lᵀ, S := 1 / L_kk
  const int ld = r | 1;
  const double* W = s_dyn;
  if (ok) {
    for (int e = tid; e < r * r; e += NT) {
      const int i = e / r, k = e - i * r;
      const double v = k <= i ? W[(size_t)i * ld + k] * fast_rsqrt(W[(size_t)k * ld + k]) : 0.0;
      pb.Vout[(size_t)i * r + k] = v;
      pb.Vtout[(size_t)k * r + i] = v;
    }
    if (tid < r) pb.Sout[tid] = fast_rsqrt(W[(size_t)tid * ld + tid]);
  }
  EIG_STAMP(4);
  if (tid == 0) {
    pb.status[0] = ok ? 0 : 2; pb.status[-1] = 0;
    if (pb.host_status) __hip_atomic_store(pb.host_status, ok ? 0 : 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  }
  __threadfence();
  __syncthreads();
  if (tid == 0 && pb.done_word) __hip_atomic_store(pb.done_word, pb.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  EIG_STAMP(5);
}

namespace {
template <class Batch>
void launch_root_batch(hipStream_t st, int r, int n, const Batch& batch) {
  ProfScope _ps(st, KID_EIGEN);
  const size_t shmem = sizeof(double) * (size_t)(r + 1) * (r | 1);
  if (factor_tile_count(r) <= 256) {
    set_dyn_lds((const void*)k_posterior_root<Batch, 256>, shmem);
    hipLaunchKernelGGL((k_posterior_root<Batch, 256>), dim3(n), dim3(256), shmem, st, r, batch);
  } else {
    set_dyn_lds((const void*)k_posterior_root<Batch, 1024>, shmem);
    hipLaunchKernelGGL((k_posterior_root<Batch, 1024>), dim3(n), dim3(1024), shmem, st, r, batch);
  }
}
}  // namespace

// ranks <= 64 by the tridiagonal route: one launch reduces (a workgroup per decomposition, with the front end of a decomposition
// enqueued ahead of its input), one solves (a wave per eigenpair); no refinement step (gaps at these ranks leave the vectors
// orthogonal to 1e-13), completion words from the solve launch's last wave
static void launch_eigen_tridiag_small(hipStream_t st, int r, const double* sqrt_lambda, int n, const EigenRequest* rq) {
  tri::TriSmallBatch b{};
  b.r = r;
  tri::TriSolveIO so[2] = {};
  tri::TriWyIO wy[2] = {};
  for (int i = 0; i < n; ++i) {
    double* base = rq[i].work + jacobi_work_doubles(r);
    double *d = base, *e = base + tri::kTriMaxN, *beta = base + 2 * tri::kTriMaxN, *mu = base + 3 * tri::kTriMaxN;
    int* sync = (int*)(base + 4 * tri::kTriMaxN);
    double* Hv = base + 4 * tri::kTriMaxN + 8;
    const EigenSpec* sp = rq[i].spec;
    tri::TriSmallProblem& p = b.p[i];
    p.M = rq[i].M;
    p.splits = sp ? sp->splits : 0;
    p.ready = sp ? sp->ready : nullptr;
    p.ready_seq = sp ? sp->ready_seq : 0;
    p.cancel = sp ? sp->cancel : nullptr;
    p.seq = sp ? sp->seq : 0;
    p.sqrt_lambda = rq[i].sqrt_lambda ? rq[i].sqrt_lambda : sqrt_lambda;
    p.out = tri::TridiagIO{r, nullptr, nullptr, d, e, beta, Hv, nullptr};
    p.sync = sync;
    // (the reflector blocks' T factors: at the head of `work`, the Jacobi kernels' area, which this route leaves alone)
    so[i] = tri::TriSolveIO{r, d, e, beta, Hv, rq[i].V, rq[i].Vt, rq[i].S, mu, rq[i].work, sync, rq[i].status, rq[i].host_status, rq[i].done_word,
                            rq[i].done_value};
    wy[i] = tri::TriWyIO{r, beta, Hv, rq[i].work, sync};
  }
  if (n == 1) { b.p[1] = b.p[0]; so[1] = so[0]; wy[1] = wy[0]; }
  ProfScope _ps(st, KID_EIGEN);
  hipLaunchKernelGGL(tri::k_tridiag_small, dim3(n), dim3(256), 0, st, b);
  hipLaunchKernelGGL(tri::k_tri_wy<1>, dim3((r - 2 + tri::kWyBlock - 1) / tri::kWyBlock, n), dim3(64), 0, st, wy[0], wy[1]);
  hipLaunchKernelGGL(tri::k_tri_solve<1>, dim3((r + 3) / 4, n), dim3(256), tri::tri_solve_lds_bytes(r), st, so[0], so[1]);
}

bool launch_posterior_eigen_pair(hipStream_t st, int r, const double* sqrt_lambda, int n, const EigenRequest* rq) {
  static const bool force_generic = dev_env("ICP_EIGEN_GENERIC") != nullptr;
  if (!(r >= 3 && r <= 64 && !force_generic) || n < 1 || n > 2) return false;
  // The direct route at these ranks is a measured alternative, not the default: from input to completion word its two
  // launches take ≈ 110 µs whatever the state, the warm-started iteration ≈ 100 even at four sweeps (70 at two); and without the
  // refinement step its eigenvectors of close eigenvalues are a little further from the oracle's than the iteration's (femur-50
  // golden proposals: 1.4e-7 against a tolerance of 1.2e-7).  Requests ask for it (EigenRequest::direct) under developer switches only.
  static const int forced = dev_env("ICP_EIGEN_TRIDIAG") ? std::atoi(dev_env("ICP_EIGEN_TRIDIAG")) : -1;
  if (rq[0].root) {  // the Cholesky-root sampler (icp_proposal_set_sampler): no decomposition at all
    RootBatch2 b{};
    b.n = n;
    for (int i = 0; i < n; ++i) b.p[i] = eigen_rr_problem(r, rq[i]);
    launch_root_batch(st, r, n, b);
    return true;
  }
  bool direct = false;
  for (int i = 0; i < n; ++i) direct = direct || rq[i].direct;
  if (forced >= 0) direct = forced != 0;
  if (direct) launch_eigen_tridiag_small(st, r, sqrt_lambda, n, rq);
  else launch_eigen_rr<2>(st, r, sqrt_lambda, n, rq);
  return true;
}

size_t eigen_many_record_bytes(int n) { return sizeof(EigenProblem) * (size_t)n; }

EigenProblem eigen_problem_of(int r, const EigenRequest& rq) { return eigen_rr_problem(r, rq); }

void launch_posterior_eigen_resident(hipStream_t st, int r, int n, const EigenProblem* records, const int* skip, int root) {
  if (n < 1) return;
  EigenBatchMem b{n, records, nullptr};
  b.skip = skip;
  if (root) launch_root_batch(st, r, n, b);
  else launch_eigen_rr_batch(st, r, nullptr, n, b);
}

int launch_posterior_eigen_many(hipStream_t st, int r, int n, const EigenRequest* rq, void* pinned_records, int* arrive) {
  static const bool force_generic = dev_env("ICP_EIGEN_GENERIC") != nullptr;
  if (!(r >= 3 && r <= 64 && !force_generic) || n < 1) return -1;
  // ONE launch while all of its workgroups can be resident together on an otherwise idle chip (a replay workgroup waits for its
  // neighbour at the sign exchange): 240 workgroups; more problems than that follow in a second launch on the same stream
  // (test hook ICP_TEST_EIGEN_CHUNK: round 2's 24 per launch, for tools/r3_timeout_repro.py)
  static const int chunk_hook = dev_env("ICP_TEST_EIGEN_CHUNK") ? std::atoi(dev_env("ICP_TEST_EIGEN_CHUNK")) : 0;
  const int per = eigen_rr_per(r), chunk = chunk_hook > 0 ? chunk_hook : 240 / per;
  EigenProblem* rec = (EigenProblem*)pinned_records;
  for (int i = 0; i < n; ++i) rec[i] = eigen_rr_problem(r, rq[i]);
  if (rq[0].root) {  // (all requests of a batch share the sampler: checked by the caller)
    launch_root_batch(st, r, n, RootBatchMem{n, rec, arrive});
    return n;
  }
  for (int i = 0; i < n; i += chunk) {
    const int m = std::min(chunk, n - i);
    launch_eigen_rr_batch(st, r, nullptr, m, EigenBatchMem{m, rec + i, arrive});
  }
  return n * per;
}

void launch_posterior_eigen(hipStream_t st, int r, const double* M, const double* sqrt_lambda, const double* Vwarm, double* V,
                            double* Vt, double* S, double* work, int* status, const EigenSpec* spec, int* host_status, int part) {
  if (tridiag_route(r) && spec == nullptr) {
    ProfScope _ps(st, KID_EIGEN);
    launch_eigen_tridiag(st, r, M, sqrt_lambda, V, Vt, S, work, status, host_status, nullptr, 0, part);
    return;
  }
  if (part == 2) return;  // (the other routes are not split: part 1 has issued all of them)
  {
    const EigenRequest rq{M, Vwarm, V, Vt, S, work, status, spec, host_status, nullptr, 0};
    if (launch_posterior_eigen_pair(st, r, sqrt_lambda, 1, &rq)) return;
  }
  if (tridiag_route(r)) {
    ProfScope _ps(st, KID_EIGEN);
    launch_eigen_tridiag(st, r, M, sqrt_lambda, V, Vt, S, work, status, host_status, nullptr, 0);
    return;
  }
  if (r > 64 && r <= kBigMaxRank) {  // in-place parallel Jacobi, packed triangle in one CU's LDS + replay workgroups
    ProfScope _ps(st, KID_EIGEN);
    launch_eigen_big(st, r, M, sqrt_lambda, Vwarm, V, Vt, S, work, status, host_status, nullptr);
    return;
  }
  // ranks above 200: the generic single-workgroup kernel (matrix behind L2)
  const int ld = r | 1;
  const size_t budget = (size_t)kLdsDoubles - 1800;  // static LDS of the kernel
  const int a_in_lds = (size_t)r * ld <= budget;
  const int v_in_lds = 2 * (size_t)r * ld <= budget;
  const size_t shmem = sizeof(double) * ((a_in_lds ? (size_t)r * ld : 0) + (v_in_lds ? (size_t)r * ld : 0));
  if (!a_in_lds) Vwarm = nullptr;  // the warm-start transform needs `work` as scratch
  set_dyn_lds((const void*)k_posterior_eigen, shmem);
  { ProfScope _ps(st, KID_EIGEN);
    hipLaunchKernelGGL(k_posterior_eigen, dim3(1), dim3(1024), shmem, st, r, M, sqrt_lambda, Vwarm, V, Vt, S, work, status, a_in_lds,
                       v_in_lds); }
}

void launch_propose(hipStream_t st, int r, const double* alpha, const double* V, const double* S,
                    const double* inv_sqrt_lambda, const double* P, double sigma2, const double* c,
                    const double* z, double step, double* c_out, int root, const int* relay_in, int* relay_out) {
  { ProfScope _ps(st, KID_PROPOSE);
    ProposeIn in{alpha, V, S, inv_sqrt_lambda, P, c, z, sigma2, step, root};
    // (ranks <= 64: 256 threads, the arithmetic of the merged step's own copy of the proposal; above, only this kernel proposes)
    // (ranks above 134 — no matrix of the proposal fits LDS and no merged step exists —: 1,024 threads, 16 lanes per row of the two
    // products with V and P, whose rows come from L2: 30 -> 14 µs at rank 200)
    if (root && r > 64) hipLaunchKernelGGL(k_propose<1024>, dim3(1), dim3(1024), 0, st, r, in, c_out, matvec_tpr_log2(r, 1024), relay_in, relay_out);
    else if (r > 134) hipLaunchKernelGGL(k_propose<1024>, dim3(1), dim3(1024), 0, st, r, in, c_out, 4, relay_in, relay_out);
    else hipLaunchKernelGGL(k_propose<256>, dim3(1), dim3(256), 0, st, r, in, c_out, matvec_tpr_log2(r, 256), relay_in, relay_out); }
}

void launch_gather_points(hipStream_t st, int K, const double* x, const int* ids, double* P) {
  if (K > 0) hipLaunchKernelGGL(k_gather_points, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, K, x, ids, P);
}
void launch_correspond_plain(hipStream_t st, int K, const int* ids, const double* pts, const double* ref, const double* mean,
                             const CorrBuffers& cb) {
  if (K > 0) hipLaunchKernelGGL(k_correspond_plain, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, K, ids, pts, ref, mean, cb);
}
void launch_mean_step(hipStream_t st, int r, const double* alpha, const double* P, double sigma2, double step, double* c) {
  hipLaunchKernelGGL(k_mean_step, dim3(1), dim3(256), 0, st, r, alpha, P, sigma2, step, c, matvec_tpr_log2(r, 256));
}

void launch_accumulate(hipStream_t st, int n, const double* src, double scale_after, double* acc) {
  hipLaunchKernelGGL(k_accumulate, dim3(cdiv(n, kBlock)), dim3(kBlock), 0, st, n, src, scale_after, acc);
}
void launch_variability(hipStream_t st, int N, int S, const double* X, int mode, const double* normals, double* out) {
  hipLaunchKernelGGL(k_variability, dim3(cdiv(N, kBlock)), dim3(kBlock), 0, st, N, S, X, mode, normals, out);
}

void launch_sum_gauss_logpdf(hipStream_t st, int K, const double* d2, double mean, double sigma, double* out) {
  { ProfScope _ps(st, KID_REDUCE);
    hipLaunchKernelGGL(k_sum_gauss_logpdf, dim3(1), dim3(kBlock), 0, st, K, d2, mean, sigma, out); }
}

void launch_dist_max(hipStream_t st, int K, const double* d2, double* out_max) {
  if (K <= 0) return;  // (an empty list leaves the zero in place: distances are non-negative)
  ProfScope _ps(st, KID_REDUCE);
  hipLaunchKernelGGL(k_dist_max, dim3((K + 1023) / 1024), dim3(1024), 0, st, K, d2, out_max);
}

void launch_dist_stats(hipStream_t st, int K, const double* d2, const unsigned char* flags, const int* idx,
                       int n_flags, double* out) {
  { ProfScope _ps(st, KID_REDUCE);
    // (one workgroup: the sum has one fixed order; four times the threads where the list is a whole mesh — 40 us per call at 28k points)
    hipLaunchKernelGGL(k_dist_stats, dim3(1), dim3(K > 4096 ? 1024 : kBlock), 0, st, K, d2, flags, idx, n_flags, out); }
}

}  // namespace icp
