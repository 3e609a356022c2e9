// kernels_posterior.hip — correspondences -> GP regression -> r-space closed forms, and evaluator reductions.
//
// Math: SURVEY.md App. A.  With Q = Φ·diag(√λ), per-correspondence noise Σ_i = σ_t² I + (σ_n² − σ_t²) n̂n̂ᵀ
// (SurfaceNoiseHelpers.scala:32-60 for an orthonormal frame), so Σ_i⁻¹ = w_t I + κ n̂n̂ᵀ, w_t = 1/σ_t², κ = 1/σ_n² − 1/σ_t²:
//   M = I + Σ_i Q_iᵀ Σ_i⁻¹ Q_i,  b = Σ_i Q_iᵀ Σ_i⁻¹ (y_i − μ_i),  α = M⁻¹ b            (NonRigidIcpProposal.scala:152)
//   propose:   c_new = (G + σ²I)⁻¹ G (α + D⁻¹ V √S z),  D M⁻¹ D = V S Vᵀ, G = QᵀQ, σ² = 1e-5      (:53-68)
//   transition: log T = −½ γᵀMγ − (r/2) ln 2π,  (G + σ²M) γ = G (c̃ − α)                           (:71-85)
//     (equal to the reference's whitened-coefficient form for ANY square root of D M⁻¹ D; derivation in DESIGN.md)
// These kernels are latency-bound r×r work (r = 51…201): one workgroup per matrix, data in LDS when it fits.
#include "icp_kernels.hpp"

namespace icp {

namespace {

constexpr int kBlock = 256;
constexpr int kLdsDoubles = 18432;  // 144 KiB of the 160 KiB LDS for one matrix-sized buffer

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

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

__global__ void __launch_bounds__(kBlock) k_correspond_model(int K, const double* __restrict__ x, const double* __restrict__ cp,
                                                              const int* __restrict__ nnv, const unsigned char* __restrict__ tgt_boundary,
                                                              int boundary_aware, Pose pose, const double* __restrict__ ref,
                                                              const double* __restrict__ mean, const int* __restrict__ tris,
                                                              const int* __restrict__ adj_off, const int* __restrict__ adj, CorrBuffers cb) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int aux = nnv ? nnv[k] : -1;
  bool on_boundary = (nnv && aux >= 0) ? tgt_boundary[aux] != 0 : false;  // :98-99
  d3 n = vertex_normal(x, tris, adj_off, adj, k);                          // :100
  write_corr(cb, k, k, aux, ld3(cp + 3 * k), boundary_aware ? !on_boundary : true, n, pose, ref, mean);
}

__global__ void __launch_bounds__(kBlock) k_correspond_target(int K, const double* __restrict__ x, const double* __restrict__ tpts,
                                                               const int* __restrict__ nn_id, const unsigned char* __restrict__ model_boundary,
                                                               int boundary_aware, Pose pose, const double* __restrict__ ref,
                                                               const double* __restrict__ mean, const int* __restrict__ tris,
                                                               const int* __restrict__ adj_off, const int* __restrict__ adj, CorrBuffers cb) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k >= K) return;
  int id = nn_id[k];                                   // :118
  bool on_boundary = model_boundary[id] != 0;          // :119
  d3 n = vertex_normal(x, tris, adj_off, adj, id);     // :120
  write_corr(cb, k, id, -1, ld3(tpts + 3 * k), boundary_aware ? !on_boundary : true, n, pose, ref, mean);
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

__global__ void __launch_bounds__(64) k_regression_mfma(int K, int kchunk, int r, const double* __restrict__ Q, CorrBuffers cb,
                                                         double wt, double kappa, double* __restrict__ Mpart) {
  const int n = r + 1, nt = (n + 15) >> 4;
  const int ti = blockIdx.x / nt, tj = blockIdx.x - ti * nt;
  const int l = threadIdx.x, i16 = l & 15, kk = l >> 4;
  const int a = 16 * ti + i16, b = 16 * tj + i16;
  const int ca = a < r ? a : 0, cbi = b < r ? b : 0;
  const double ma = a < r ? 1.0 : 0.0, mb = b < r ? 1.0 : 0.0;      // basis column?
  const double ea = a == r ? 1.0 : 0.0, eb = b == r ? 1.0 : 0.0;    // the appended observation column?
  const int k0 = blockIdx.y * kchunk, k1 = min(K, k0 + kchunk);
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
  double* out = Mpart + (size_t)blockIdx.y * n * n;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const int row = 16 * ti + kk + 4 * g, col = 16 * tj + i16;
    if (row < n && col < n) out[(size_t)row * n + col] = acc[g];
  }
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
      const double* a = A + (size_t)row * lda;
#pragma unroll 8
      for (int j = sub; j < r; j += tpr) acc = fma(a[j], x[j], acc);
    }
    for (int o = tpr >> 1; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (row < r && sub == 0) y[row] = acc;
  }
  __syncthreads();
}

// threads per row for block_matvec: enough rows in flight to occupy the block, at least ~8 terms per thread
inline int matvec_tpr_log2(int r, int block) {
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

extern __shared__ double s_dyn[];

struct FactorArgs {  // up to 4 posteriors per launch (both ICP directions of one or two states)
  const double* Mpart[4];
  int splits[4];
  double* M[4];
  double* alpha[4];
  int* status[4];
  double* scratch[4];  // (r+1)·r doubles, used only when the matrix does not fit in LDS
};

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
    for (int s = 0; s < S; ++s) m += Mpart[(size_t)s * n * n + (size_t)i * n + j];
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

// Fast path (the matrix fits a few elements per thread): every thread OWNS E fixed elements of the lower triangle of
// [M; bᵀ] and keeps them in registers for the whole root-free factorisation; only the current pivot column is
// published through a double-buffered LDS vector, so a column costs one reciprocal, ~E fused multiply-adds per thread
// and ONE barrier.  The finished factor goes to LDS once, for the back substitution.
template <int E, int NT>
__global__ void __launch_bounds__(NT) k_posterior_factor_reg(int r, FactorArgs fa) {
  __shared__ double s_col[2][516], s_dinv[512], s_v[512];
  const int tid = threadIdx.x, n = r + 1, p = blockIdx.x;
  const double* Mpart = fa.Mpart[p];
  const int S = fa.splits[p];
  double* M = fa.M[p];
  const int ld = r | 1;
  double* W = s_dyn;  // (r+1) × ld
  const int tri = r * (r + 1) / 2, total = tri + r;
  double v[E];
  int im[E], km[E];
#pragma unroll
  for (int m = 0; m < E; ++m) {
    const int e = tid + NT * m;
    int i = r, k = e - tri;
    if (e < tri) {
      i = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while ((i + 1) * (i + 2) / 2 <= e) ++i;
      while (i * (i + 1) / 2 > e) --i;
      k = e - i * (i + 1) / 2;
    }
    im[m] = i; km[m] = e < total ? k : -1;  // k = -1: slot unused (never > j, never published)
    double x = 0.0;
    if (e < total) {
      for (int s = 0; s < S; ++s) x += Mpart[(size_t)s * n * n + (size_t)i * n + k];
      if (i < r) {
        if (i == k) x += 1.0;
        M[(size_t)i * r + k] = x;
        M[(size_t)k * r + i] = x;
      }
      if (k == 0) s_col[0][i] = x;
    }
    v[m] = x;
  }
  __syncthreads();
  for (int j = 0; j < r; ++j) {
    const double* cur = s_col[j & 1];
    double* nxt = s_col[(j + 1) & 1];
    const double ajj = cur[j];
    if (!(ajj > 0.0)) {  // same value in every thread: uniform exit
      if (tid == 0) fa.status[p][0] = 1;
      return;
    }
    const double inv = fast_rcp(ajj);
#pragma unroll
    for (int m = 0; m < E; ++m) {
      if (km[m] > j) {
        v[m] = fma(-(cur[im[m]] * inv), cur[km[m]], v[m]);
        if (km[m] == j + 1) nxt[im[m]] = v[m];  // column j+1 is final now: publish it as the next pivot column
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int m = 0; m < E; ++m)
    if (km[m] >= 0) W[(size_t)im[m] * ld + km[m]] = v[m];
  __syncthreads();
  if (tid == 0) fa.status[p][0] = 0;
  // y = L⁻¹ b sits (unscaled) in row r: y_j = W[r][j]·dinv_j,  L[i][j] = W[i][j]·dinv_j
  for (int j = tid; j < r; j += NT) {
    const double d = fast_rsqrt(W[(size_t)j * ld + j]);
    s_dinv[j] = d;
    s_v[j] = W[(size_t)r * ld + j] * d;
  }
  __syncthreads();
  if (r <= 64) {
    if (tid < 64) {  // one wave, registers + cross-lane reads: no barriers on the sequential chain
      const int i = tid;
      double x = i < r ? s_v[i] : 0.0;
      const double di = i < r ? s_dinv[i] : 0.0;
      for (int j = r - 1; j >= 0; --j) {
        const double xj = __shfl(x, j, 64) * s_dinv[j];
        if (i == j) x = xj;
        else if (i < j) x = fma(-(W[(size_t)j * ld + i] * di), xj, x);
      }
      if (i < r) fa.alpha[p][i] = x;
    }
  } else {
    for (int j = r - 1; j >= 0; --j) {
      if (tid == 0) s_v[j] = s_v[j] * s_dinv[j];
      __syncthreads();
      const double xj = s_v[j];
      for (int i = tid; i < j; i += NT) s_v[i] = fma(-(W[(size_t)j * ld + i] * s_dinv[i]), xj, s_v[i]);
      __syncthreads();
    }
    for (int i = tid; i < r; i += NT) fa.alpha[p][i] = s_v[i];
  }
}

// ---------------------------------------------------------------- a9 transition tails (batched, one workgroup each)
// (G + σ²M) γ = G d  ⇔  γ = d − σ² G⁻¹ M γ : fixed-point iteration with contraction factor ρ(σ² G⁻¹ M) (≈ 4e-8 for the
// femur model), run to machine precision; status != 0 if it does not contract, and the host then uses the direct
// (Cholesky) kernel below.  M and G⁻¹ are staged in LDS when they fit.

struct TailArgs {
  int n;
  const double* alpha[8];
  const double* M[8];
  const double* c_from[8];
  const double* c_to[8];
  double step[8];
  double* out[8];
  int* status[8];
};

__global__ void __launch_bounds__(256) k_transition_tails(int r, TailArgs ta, const double* __restrict__ Ginv, double sigma2,
                                                           int n_lds, int tpr_log2) {
  __shared__ double s_d[512], s_g[512], s_t[512], s_u[512], s_red[16];
  const int tid = threadIdx.x, nt = blockDim.x, t = blockIdx.x;
  const int ld = r | 1;
  const double* M = ta.M[t];
  const double* Gi = Ginv;
  int ldm = r, ldg = r;
  if (n_lds >= 1) { stage_matrix(r, ta.M[t], s_dyn, ld); M = s_dyn; ldm = ld; }
  if (n_lds >= 2) { stage_matrix(r, Ginv, s_dyn + (size_t)r * ld, ld); Gi = s_dyn + (size_t)r * ld; ldg = ld; }
  for (int i = tid; i < r; i += nt) {
    const double d = (ta.c_from[t][i] + (ta.c_to[t][i] - ta.c_from[t][i]) / ta.step[t]) - ta.alpha[t][i];  // :79 minus posterior mean
    s_d[i] = d;
    s_g[i] = d;
  }
  __syncthreads();
  int converged = 0;
  for (int it = 0; it < 12 && !converged; ++it) {
    block_matvec(r, M, ldm, s_g, s_t, tpr_log2);
    block_matvec(r, Gi, ldg, s_t, s_u, tpr_log2);
    double delta = 0.0, gmax = 0.0;
    for (int i = tid; i < r; i += nt) {
      const double gn = fma(-sigma2, s_u[i], s_d[i]);
      delta = fmax(delta, fabs(gn - s_g[i]));
      gmax = fmax(gmax, fabs(gn));
      s_g[i] = gn;
    }
    delta = block_max(delta, s_red);
    gmax = block_max(gmax, s_red);
    converged = delta <= 1e-15 * gmax || gmax == 0.0;
    __syncthreads();
  }
  block_matvec(r, M, ldm, s_g, s_t, tpr_log2);
  double part = 0.0;
  for (int i = tid; i < r; i += nt) part = fma(s_g[i], s_t[i], part);
  const double q = block_sum(part, s_red);
  if (tid == 0) {
    ta.out[t][0] = -0.5 * q - 0.5 * (double)r * 1.8378770664093453;  // ln(2π); no log-det term
    ta.status[t][0] = converged ? 0 : 3;
  }
}

// direct version: factor G + σ²M, solve, quadratic form.  Used only if the iteration above reports non-contraction.
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

__device__ __forceinline__ void rr_pair(int n2, int rnd, int slot, int* p, int* q) {
  const int m = n2 - 1;
  int a, b;
  if (slot == 0) { a = m; b = rnd % m; }
  else { a = (rnd + slot) % m; b = (rnd - slot + m) % m; }
  *p = a < b ? a : b;
  *q = a < b ? b : a;
}

__global__ void __launch_bounds__(1024) k_posterior_eigen(int r, const double* __restrict__ M, const double* __restrict__ sqrt_lambda,
                                                           const double* __restrict__ Vwarm, double* __restrict__ Vout,
                                                           double* __restrict__ Vtout, double* __restrict__ Sout,
                                                           double* __restrict__ work, int* __restrict__ status, int a_in_lds, int v_in_lds) {
  __shared__ double s_red[16], s_mu[512], s_sgn[512], s_c[256], s_s[256];
  __shared__ int s_rank[512];
  __shared__ short s_p[256], s_q[256];
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

// ---------------------------------------------------------------- a8 propose
// c_new = (G + σ²I)⁻¹ G w = w − σ² P w with P = (G + σ²I)⁻¹ precomputed;  w = α + D⁻¹ V (√S ∘ z)

__global__ void __launch_bounds__(256) k_propose(int r, const double* __restrict__ alpha, const double* __restrict__ V,
                                                  const double* __restrict__ S, const double* __restrict__ inv_sqrt_lambda,
                                                  const double* __restrict__ P, double sigma2, const double* __restrict__ c,
                                                  const double* __restrict__ z, double step, double* __restrict__ c_out, int tpr_log2) {
  __shared__ double s_x[512], s_y[512], s_w[512];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = tid; j < r; j += nt) s_x[j] = sqrt(S[j]) * z[j];
  __syncthreads();
  block_matvec(r, V, r, s_x, s_y, tpr_log2);
  for (int i = tid; i < r; i += nt) s_w[i] = fma(s_y[i], inv_sqrt_lambda[i], alpha[i]);
  __syncthreads();
  block_matvec(r, P, r, s_w, s_y, tpr_log2);
  for (int i = tid; i < r; i += nt) {
    const double cnew = fma(-sigma2, s_y[i], s_w[i]);  // model.coefficients(...) with σ² = 1e-5 (:59)
    c_out[i] = c[i] + (cnew - c[i]) * step;            // :61-62
  }
}

// ---------------------------------------------------------------- evaluator reductions

__global__ void __launch_bounds__(kBlock) k_sum_gauss_logpdf(int K, const double* __restrict__ d2, double mean, double sigma,
                                                              double* __restrict__ out) {
  __shared__ double s_red[16];
  const double lognorm = log(sqrt(2.0 * 3.14159265358979323846)) + log(sigma);  // Breeze Gaussian.logNormalizer
  double part = 0.0;
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    double d = (sqrt(d2[k]) - mean) / sigma;
    part += -d * d / 2.0 - lognorm;
  }
  double t = block_sum(part, s_red);
  if (threadIdx.x == 0) out[0] = t;
}

__global__ void __launch_bounds__(kBlock) k_dist_stats(int K, const double* __restrict__ d2, const unsigned char* __restrict__ flags,
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

}  // namespace

void launch_correspond_model(hipStream_t st, int K, const double* x, const double* cp, const int* nnv,
                             const unsigned char* tgt_boundary, int boundary_aware, const Pose& pose,
                             const double* ref, const double* mean, const int* tris, const int* adj_off,
                             const int* adj, const CorrBuffers& cb) {
  if (K <= 0) return;
  { ProfScope _ps(st, KID_CORRESPOND);
    hipLaunchKernelGGL(k_correspond_model, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, K, x, cp, nnv, tgt_boundary,
                     boundary_aware, pose, ref, mean, tris, adj_off, adj, cb); }
}

void launch_correspond_target(hipStream_t st, int K, const double* x, const double* tpts, const int* nn_id,
                              const unsigned char* model_boundary, int boundary_aware, const Pose& pose,
                              const double* ref, const double* mean, const int* tris, const int* adj_off,
                              const int* adj, const CorrBuffers& cb) {
  if (K <= 0) return;
  { ProfScope _ps(st, KID_CORRESPOND);
    hipLaunchKernelGGL(k_correspond_target, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, K, x, tpts, nn_id, model_boundary,
                     boundary_aware, pose, ref, mean, tris, adj_off, adj, cb); }
}

int regression_splits(int K) {
  int s = (K + 23) / 24;  // ~24 correspondences per wave: short dependent chains, enough waves to overlap the gathers
  return s < 1 ? 1 : (s > 64 ? 64 : s);
}

void launch_regression(hipStream_t st, int K, int r, const double* Q, const CorrBuffers& cb, double w_tangent,
                       double kappa, double* Mpart, int* splits_out) {
  const int n = r + 1, nt = (n + 15) / 16;
  const int S = regression_splits(K);
  int kchunk = (K + S - 1) / S;
  if (kchunk < 1) kchunk = 1;
  *splits_out = S;
  { ProfScope _ps(st, KID_REGRESSION);
    hipLaunchKernelGGL(k_regression_mfma, dim3(nt * nt, S), dim3(64), 0, st, K, kchunk, r, Q, cb, w_tangent, kappa, Mpart); }
}

static void set_dyn_lds(const void* fn, size_t bytes) {
  if (bytes > 48 * 1024) (void)hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
}

template <int E, int NT>
static void launch_factor_reg(hipStream_t st, int r, int n_post, const FactorArgs& fa) {
  const size_t shmem = sizeof(double) * (size_t)(r + 1) * (r | 1);
  set_dyn_lds((const void*)k_posterior_factor_reg<E, NT>, shmem);
  hipLaunchKernelGGL((k_posterior_factor_reg<E, NT>), dim3(n_post), dim3(NT), shmem, st, r, fa);
}

void launch_posterior_factor(hipStream_t st, int r, int n_post, const PosteriorFactorIO* io) {
  FactorArgs fa{};
  for (int p = 0; p < n_post; ++p) {
    fa.Mpart[p] = io[p].Mpart; fa.splits[p] = io[p].splits; fa.M[p] = io[p].M; fa.alpha[p] = io[p].alpha;
    fa.status[p] = io[p].status; fa.scratch[p] = io[p].scratch;
  }
  const int ld = r | 1;
  const size_t total = (size_t)r * (r + 1) / 2 + r;
  const bool w_fits = (size_t)(r + 1) * ld <= (size_t)kLdsDoubles - 2000;  // + the static LDS of the kernel
  ProfScope _ps(st, KID_FACTOR);
  if (w_fits && total <= 256 * 6) launch_factor_reg<6, 256>(st, r, n_post, fa);
  else if (w_fits && total <= 256 * 12) launch_factor_reg<12, 256>(st, r, n_post, fa);
  else if (w_fits && total <= 1024 * 6) launch_factor_reg<6, 1024>(st, r, n_post, fa);
  else if (w_fits && total <= 1024 * 12) launch_factor_reg<12, 1024>(st, r, n_post, fa);
  else {
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
  }
  const int ld = r | 1;
  const size_t one = (size_t)r * ld;
  const int n_lds = 2 * one <= (size_t)kLdsDoubles - 2560 ? 2 : (one <= (size_t)kLdsDoubles - 2560 ? 1 : 0);
  const size_t shmem = sizeof(double) * one * n_lds;
  set_dyn_lds((const void*)k_transition_tails, shmem);
  { ProfScope _ps(st, KID_TAIL);
    hipLaunchKernelGGL(k_transition_tails, dim3(n), dim3(256), shmem, st, r, ta, Ginv, sigma2, n_lds, matvec_tpr_log2(r, 256)); }
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

void launch_posterior_eigen(hipStream_t st, int r, const double* M, const double* sqrt_lambda, const double* Vwarm, double* V,
                            double* Vt, double* S, double* work, int* status) {
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
                    const double* z, double step, double* c_out) {
  { ProfScope _ps(st, KID_PROPOSE);
    hipLaunchKernelGGL(k_propose, dim3(1), dim3(256), 0, st, r, alpha, V, S, inv_sqrt_lambda, P, sigma2, c, z, step, c_out,
                       matvec_tpr_log2(r, 256)); }
}

void launch_sum_gauss_logpdf(hipStream_t st, int K, const double* d2, double mean, double sigma, double* out) {
  { ProfScope _ps(st, KID_REDUCE);
    hipLaunchKernelGGL(k_sum_gauss_logpdf, dim3(1), dim3(kBlock), 0, st, K, d2, mean, sigma, out); }
}

void launch_dist_stats(hipStream_t st, int K, const double* d2, const unsigned char* flags, const int* idx,
                       int n_flags, double* out) {
  { ProfScope _ps(st, KID_REDUCE);
    hipLaunchKernelGGL(k_dist_stats, dim3(1), dim3(kBlock), 0, st, K, d2, flags, idx, n_flags, out); }
}

}  // namespace icp
