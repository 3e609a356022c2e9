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
#include "icp_dense.hpp"

namespace icp {

namespace {

constexpr int kBlock = 256;

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// ---------------------------------------------------------------- correspondences (one thread per correspondence)

__global__ void __launch_bounds__(kBlock) k_correspond_model(CorrTask c, const double* __restrict__ cp) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < c.K) correspond_model_one(c, k, ld3(cp + 3 * k));
}

__global__ void __launch_bounds__(kBlock) k_correspond_target(CorrTask c, const int* __restrict__ nn_id) {
  int k = blockIdx.x * kBlock + threadIdx.x;
  if (k < c.K) correspond_target_one(c, k, nn_id[k]);
}

__global__ void __launch_bounds__(64) k_regression_mfma(int K, int kchunk, int r, const double* __restrict__ Q, CorrBuffers cb,
                                                         double wt, double kappa, double* __restrict__ Mpart) {
  regression_tile(blockIdx.x, blockIdx.y, K, kchunk, r, Q, cb, wt, kappa, Mpart);
}

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

template <int E, int NT>
__global__ void __launch_bounds__(NT) k_posterior_factor_reg(int r, FactorArgs fa) {
  const int p = blockIdx.x;
  factor_reg_body<E, NT>(r, fa.Mpart[p], fa.splits[p], fa.M[p], fa.alpha[p], fa.status[p]);
}

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
  const int t = blockIdx.x;
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

__global__ void __launch_bounds__(256) k_propose(int r, ProposeIn in, double* __restrict__ c_out, int tpr_log2) {
  propose_body(r, in, c_out, tpr_log2);
}

// ---------------------------------------------------------------- evaluator reductions

__global__ void __launch_bounds__(kBlock) k_sum_gauss_logpdf(int K, const double* __restrict__ d2, double mean, double sigma,
                                                              double* __restrict__ out) {
  sum_gauss_logpdf_body(K, d2, mean, sigma, out);
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
  CorrTask c{K, cb, x, nullptr, tgt_boundary, nnv, boundary_aware, pose, ref, mean, tris, adj_off, adj};
  ProfScope _ps(st, KID_CORRESPOND);
  hipLaunchKernelGGL(k_correspond_model, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, c, cp);
}

void launch_correspond_target(hipStream_t st, int K, const double* x, const double* tpts, const int* nn_id,
                              const unsigned char* model_boundary, int boundary_aware, const Pose& pose,
                              const double* ref, const double* mean, const int* tris, const int* adj_off,
                              const int* adj, const CorrBuffers& cb) {
  if (K <= 0) return;
  CorrTask c{K, cb, x, tpts, model_boundary, nullptr, boundary_aware, pose, ref, mean, tris, adj_off, adj};
  ProfScope _ps(st, KID_CORRESPOND);
  hipLaunchKernelGGL(k_correspond_target, dim3(cdiv(K, kBlock)), dim3(kBlock), 0, st, c, nn_id);
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
    ProposeIn in{alpha, V, S, inv_sqrt_lambda, P, c, z, sigma2, step};
    hipLaunchKernelGGL(k_propose, dim3(1), dim3(256), 0, st, r, in, c_out, matvec_tpr_log2(r, 256)); }
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
