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

// ---------------------------------------------------------------- K5a regression assembly
// One thread per entry (a,b) of the augmented (r+1)x(r+1) normal matrix, 16x16 entries per workgroup; the
// correspondence loop is wave-uniform (ids, normals, keep flags via the scalar unit).

__global__ void __launch_bounds__(kBlock) k_regression(int K, int r, const double* __restrict__ Q, CorrBuffers cb,
                                                        double wt, double kappa, double* __restrict__ Maug) {
  const int a = blockIdx.y * 16 + (threadIdx.x >> 4);
  const int b = blockIdx.x * 16 + (threadIdx.x & 15);
  const int n = r + 1;
  const bool live = a < n && b < n;
  const int ca = a < r ? a : 0, cb_ = b < r ? b : 0;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    if (!cb.keep[k]) continue;
    const double* q = Q + (size_t)3 * cb.id[k] * r;
    const double n0 = cb.nhat[3 * k], n1 = cb.nhat[3 * k + 1], n2 = cb.nhat[3 * k + 2];
    double a0, a1, a2, b0, b1, b2;
    if (a < r) { a0 = q[ca]; a1 = q[r + ca]; a2 = q[2 * r + ca]; }
    else { a0 = cb.e[3 * k]; a1 = cb.e[3 * k + 1]; a2 = cb.e[3 * k + 2]; }
    if (b < r) { b0 = q[cb_]; b1 = q[r + cb_]; b2 = q[2 * r + cb_]; }
    else { b0 = cb.e[3 * k]; b1 = cb.e[3 * k + 1]; b2 = cb.e[3 * k + 2]; }
    double va = fma(a2, n2, fma(a1, n1, a0 * n0));
    double vb = fma(b2, n2, fma(b1, n1, b0 * n0));
    double dab = fma(a2, b2, fma(a1, b1, a0 * b0));
    acc = fma(wt, dab, fma(kappa * va, vb, acc));
  }
  if (live) Maug[(size_t)a * n + b] = acc;
}

// ---------------------------------------------------------------- dense helpers (one workgroup, matrix behind a generic pointer)

// in-place lower Cholesky of the n×n matrix A (leading dimension n); returns false if not SPD
__device__ bool block_cholesky(double* A, int n, int* s_flag) {
  const int tid = threadIdx.x, nt = blockDim.x;
  if (tid == 0) *s_flag = 0;
  __syncthreads();
  for (int j = 0; j < n; ++j) {
    // column j: subtract the contributions of columns < j was already applied (right-looking)
    double ajj = A[(size_t)j * n + j];
    if (!(ajj > 0.0)) {
      if (tid == 0) *s_flag = 1;
    }
    __syncthreads();
    if (*s_flag) return false;
    double l = sqrt(ajj);
    for (int i = j + 1 + tid; i < n; i += nt) A[(size_t)i * n + j] = A[(size_t)i * n + j] / l;
    __syncthreads();
    if (tid == 0) A[(size_t)j * n + j] = l;
    // trailing update of the lower triangle: A[i][k] -= L[i][j]·L[k][j], j < k <= i < n
    const int m = n - j - 1;
    const int total = m * (m + 1) / 2;
    for (int e = tid; e < total; e += nt) {
      // unrank e -> (ii >= kk) within the m×m lower triangle
      int ii = (int)((sqrt(8.0 * (double)e + 1.0) - 1.0) * 0.5);
      while ((ii + 1) * (ii + 2) / 2 <= e) ++ii;
      while (ii * (ii + 1) / 2 > e) --ii;
      int kk = e - ii * (ii + 1) / 2;
      int i = j + 1 + ii, k = j + 1 + kk;
      A[(size_t)i * n + k] = fma(-A[(size_t)i * n + j], A[(size_t)k * n + j], A[(size_t)i * n + k]);
    }
    __syncthreads();
  }
  return true;
}

// solve L Lᵀ x = b in place (x in shared/global vector v of length n); L lower (leading dimension n)
__device__ void block_chol_solve(const double* L, int n, double* v) {
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = 0; j < n; ++j) {  // forward, column oriented
    if (tid == 0) v[j] = v[j] / L[(size_t)j * n + j];
    __syncthreads();
    double vj = v[j];
    for (int i = j + 1 + tid; i < n; i += nt) v[i] = fma(-L[(size_t)i * n + j], vj, v[i]);
    __syncthreads();
  }
  for (int j = n - 1; j >= 0; --j) {  // backward: Lᵀ x = y, row j of Lᵀ = column j of L
    if (tid == 0) v[j] = v[j] / L[(size_t)j * n + j];
    __syncthreads();
    double vj = v[j];
    for (int i = tid; i < j; i += nt) v[i] = fma(-L[(size_t)j * n + i], vj, v[i]);
    __syncthreads();
  }
}

__device__ double block_sum(double v, double* s_red) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o, 64);
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = 0.0;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += s_red[w];
  return t;
}
__device__ double block_max(double v, double* s_red) {
  const int tid = threadIdx.x;
  for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
  __syncthreads();
  if ((tid & 63) == 0) s_red[tid >> 6] = v;
  __syncthreads();
  double t = s_red[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) t = fmax(t, s_red[w]);
  return t;
}

// ---------------------------------------------------------------- K5b factorisations

extern __shared__ double s_dyn[];

__global__ void __launch_bounds__(kBlock) k_posterior_factor(int r, const double* __restrict__ Maug, const double* __restrict__ G,
                                                              double sigma2, double* __restrict__ M, double* __restrict__ L,
                                                              double* __restrict__ alpha, double* __restrict__ L2,
                                                              int* __restrict__ status, int use_lds) {
  __shared__ int s_flag;
  const int tid = threadIdx.x, nt = blockDim.x, n = r + 1;
  const bool second = blockIdx.x == 1;
  double* out = second ? L2 : L;
  double* W = use_lds ? s_dyn : out;  // factor in LDS when it fits, else in place in global memory
  for (int e = tid; e < r * r; e += nt) {
    int i = e / r, j = e - i * r;
    double m = Maug[(size_t)i * n + j] + (i == j ? 1.0 : 0.0);
    if (!second) M[e] = m;
    W[e] = second ? fma(sigma2, m, G[e]) : m;
  }
  __syncthreads();
  bool ok = block_cholesky(W, r, &s_flag);
  if (tid == 0) status[blockIdx.x] = ok ? 0 : 1;
  if (!ok) return;
  if (!second) {
    __shared__ double s_v[256];
    double* v = r <= 256 ? s_v : alpha;
    for (int i = tid; i < r; i += nt) v[i] = Maug[(size_t)i * n + r];
    __syncthreads();
    block_chol_solve(W, r, v);
    if (v != alpha)
      for (int i = tid; i < r; i += nt) alpha[i] = v[i];
  }
  if (use_lds) {
    __syncthreads();
    for (int e = tid; e < r * r; e += nt) out[e] = W[e];
  }
}

// ---------------------------------------------------------------- a9 transition tail

__global__ void __launch_bounds__(kBlock) k_transition_tail(int r, const double* __restrict__ alpha, const double* __restrict__ M,
                                                             const double* __restrict__ L2, const double* __restrict__ G,
                                                             const double* __restrict__ c_from, const double* __restrict__ c_to,
                                                             double step, double* __restrict__ out) {
  __shared__ double s_d[512], s_g[512], s_red[8];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = tid; j < r; j += nt) s_d[j] = (c_from[j] + (c_to[j] - c_from[j]) / step) - alpha[j];  // :79 minus posterior mean
  __syncthreads();
  for (int i = tid; i < r; i += nt) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(G[(size_t)i * r + j], s_d[j], s);
    s_g[i] = s;
  }
  __syncthreads();
  block_chol_solve(L2, r, s_g);  // γ
  double part = 0.0;
  for (int i = tid; i < r; i += nt) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(M[(size_t)i * r + j], s_g[j], s);
    part = fma(s_g[i], s, part);
  }
  double q = block_sum(part, s_red);
  if (tid == 0) out[0] = -0.5 * q - 0.5 * (double)r * 1.8378770664093453;  // ln(2π)
}

// ---------------------------------------------------------------- posterior KL basis: parallel two-sided Jacobi

// round-robin (circle method) pairing of n2 (even) players in round `rnd`; slot 0..n2/2-1
__device__ __forceinline__ void rr_pair(int n2, int rnd, int slot, int* p, int* q) {
  const int m = n2 - 1;
  int a, b;
  if (slot == 0) { a = m; b = rnd % m; }
  else { a = (rnd + slot) % m; b = (rnd - slot + m) % m; }
  *p = a < b ? a : b;
  *q = a < b ? b : a;
}

__global__ void __launch_bounds__(kBlock) k_posterior_eigen(int r, const double* __restrict__ M, const double* __restrict__ sqrt_lambda,
                                                             double* __restrict__ Vout, double* __restrict__ Sout,
                                                             double* __restrict__ work, int* __restrict__ status, int a_in_lds, int v_in_lds) {
  __shared__ double s_c[256], s_s[256], s_red[8], s_mu[512], s_sgn[512];
  __shared__ int s_p[256], s_q[256], s_rank[512];
  const int tid = threadIdx.x, nt = blockDim.x;
  double* A = a_in_lds ? s_dyn : work;
  double* V = v_in_lds ? (s_dyn + (a_in_lds ? r * r : 0)) : Vout;
  // A = D⁻¹ M D⁻¹ (same eigenvectors as D M⁻¹ D, reciprocal eigenvalues); V = I
  for (int e = tid; e < r * r; e += nt) {
    int i = e / r, j = e - i * r;
    double mij = 0.5 * (M[e] + M[(size_t)j * r + i]);
    A[e] = mij / (sqrt_lambda[i] * sqrt_lambda[j]);
    V[e] = i == j ? 1.0 : 0.0;
  }
  __syncthreads();
  const int n2 = (r + 1) & ~1, half = n2 / 2;
  int converged = 0;
  for (int sweep = 0; sweep < 40 && !converged; ++sweep) {
    for (int rnd = 0; rnd < n2 - 1; ++rnd) {
      if (tid < half) {
        int p, q;
        rr_pair(n2, rnd, tid, &p, &q);
        double c = 1.0, s = 0.0;
        if (q < r) {
          double apq = A[(size_t)p * r + q], app = A[(size_t)p * r + p], aqq = A[(size_t)q * r + q];
          if (fabs(apq) > 1e-300 && fabs(apq) > 1e-18 * sqrt(fabs(app * aqq))) {
            double tau = (aqq - app) / (2.0 * apq);
            double t = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(1.0 + tau * tau));
            c = 1.0 / sqrt(1.0 + t * t);
            s = t * c;
          }
        } else {
          p = -1;
        }
        s_p[tid] = p; s_q[tid] = q; s_c[tid] = c; s_s[tid] = s;
      }
      __syncthreads();
      // rows p,q of A:  A' = Jᵀ A
      for (int e = tid; e < half * r; e += nt) {
        int pr = e / r, k = e - pr * r, p = s_p[pr], q = s_q[pr];
        if (p < 0) continue;
        double c = s_c[pr], s = s_s[pr];
        double apk = A[(size_t)p * r + k], aqk = A[(size_t)q * r + k];
        A[(size_t)p * r + k] = c * apk - s * aqk;
        A[(size_t)q * r + k] = s * apk + c * aqk;
      }
      __syncthreads();
      // columns p,q of A and V:  A'' = A' J,  V' = V J
      for (int e = tid; e < half * r; e += nt) {
        int pr = e / r, k = e - pr * r, p = s_p[pr], q = s_q[pr];
        if (p < 0) continue;
        double c = s_c[pr], s = s_s[pr];
        double akp = A[(size_t)k * r + p], akq = A[(size_t)k * r + q];
        A[(size_t)k * r + p] = c * akp - s * akq;
        A[(size_t)k * r + q] = s * akp + c * akq;
        double vkp = V[(size_t)k * r + p], vkq = V[(size_t)k * r + q];
        V[(size_t)k * r + p] = c * vkp - s * vkq;
        V[(size_t)k * r + q] = s * vkp + c * vkq;
      }
      __syncthreads();
    }
    double off = 0.0, dg = 0.0;
    for (int e = tid; e < r * r; e += nt) {
      int i = e / r, j = e - i * r;
      double v = A[e];
      if (i == j) dg = fma(v, v, dg);
      else off = fma(v, v, off);
    }
    off = block_sum(off, s_red);
    dg = block_sum(dg, s_red);
    converged = off <= 1e-30 * dg;
  }
  if (tid == 0) status[0] = converged ? 0 : 2;
  // eigenvalues of D M⁻¹ D are 1/μ; order S descending = μ ascending (ties: lower original index first)
  for (int i = tid; i < r; i += nt) s_mu[i] = A[(size_t)i * r + i];
  __syncthreads();
  for (int i = tid; i < r; i += nt) {
    int rank = 0;
    double mi = s_mu[i];
    for (int j = 0; j < r; ++j) rank += (s_mu[j] < mi) || (s_mu[j] == mi && j < i);
    s_rank[i] = rank;
  }
  __syncthreads();
  // column i of V -> column rank[i] of Vout, with the largest-|.| component made positive
  for (int i = tid; i < r; i += nt) {
    int best = 0;
    double bv = fabs(V[i]);
    for (int k = 1; k < r; ++k) {
      double a = fabs(V[(size_t)k * r + i]);
      if (a > bv) { bv = a; best = k; }
    }
    s_sgn[i] = V[(size_t)best * r + i] < 0.0 ? -1.0 : 1.0;
    Sout[s_rank[i]] = 1.0 / s_mu[i];
  }
  __syncthreads();
  if (v_in_lds) {
    for (int e = tid; e < r * r; e += nt) {
      int k = e / r, i = e - k * r;
      Vout[(size_t)k * r + s_rank[i]] = V[e] * s_sgn[i];
    }
  } else {
    // V aliases Vout: permute through `work` (free if A sat in LDS; otherwise A lived there and is dead now)
    for (int e = tid; e < r * r; e += nt) work[e] = V[e];
    __syncthreads();
    for (int e = tid; e < r * r; e += nt) {
      int k = e / r, i = e - k * r;
      Vout[(size_t)k * r + s_rank[i]] = work[e] * s_sgn[i];
    }
  }
}

// ---------------------------------------------------------------- a8 propose

__global__ void __launch_bounds__(kBlock) k_propose(int r, const double* __restrict__ alpha, const double* __restrict__ V,
                                                     const double* __restrict__ S, const double* __restrict__ inv_sqrt_lambda,
                                                     const double* __restrict__ G, const double* __restrict__ Lg,
                                                     const double* __restrict__ c, const double* __restrict__ z, double step,
                                                     double* __restrict__ c_out) {
  __shared__ double s_w[512], s_g[512], s_z[512];
  const int tid = threadIdx.x, nt = blockDim.x;
  for (int j = tid; j < r; j += nt) s_z[j] = sqrt(S[j]) * z[j];
  __syncthreads();
  for (int i = tid; i < r; i += nt) {  // w = α + D⁻¹ V (√S ∘ z): coefficients of the sampled field in the scaled basis
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(V[(size_t)i * r + j], s_z[j], s);
    s_w[i] = fma(s, inv_sqrt_lambda[i], alpha[i]);
  }
  __syncthreads();
  for (int i = tid; i < r; i += nt) {
    double s = 0.0;
    for (int j = 0; j < r; ++j) s = fma(G[(size_t)i * r + j], s_w[j], s);
    s_g[i] = s;
  }
  __syncthreads();
  block_chol_solve(Lg, r, s_g);  // model.coefficients(...) with σ² = 1e-5 (:59)
  for (int j = tid; j < r; j += nt) c_out[j] = c[j] + (s_g[j] - c[j]) * step;  // :61-62
}

// ---------------------------------------------------------------- evaluator reductions

__global__ void __launch_bounds__(kBlock) k_sum_gauss_logpdf(int K, const double* __restrict__ d2, double mean, double sigma,
                                                              double* __restrict__ out) {
  __shared__ double s_red[8];
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
  __shared__ double s_red[8];
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

void launch_regression(hipStream_t st, int K, int r, const double* Q, const CorrBuffers& cb, double w_tangent,
                       double kappa, double* Maug) {
  int nb = cdiv(r + 1, 16);
  { ProfScope _ps(st, KID_REGRESSION);
    hipLaunchKernelGGL(k_regression, dim3(nb, nb), dim3(kBlock), 0, st, K, r, Q, cb, w_tangent, kappa, Maug); }
}

void launch_posterior_factor(hipStream_t st, int r, const double* Maug, const double* G, double sigma2, double* M,
                             double* L, double* alpha, double* L2, int* status) {
  int use_lds = r * r <= kLdsDoubles;
  size_t shmem = use_lds ? sizeof(double) * r * r : 0;
  { ProfScope _ps(st, KID_FACTOR);
    hipLaunchKernelGGL(k_posterior_factor, dim3(2), dim3(kBlock), shmem, st, r, Maug, G, sigma2, M, L, alpha, L2, status, use_lds); }
}

void launch_transition_tail(hipStream_t st, int r, const double* alpha, const double* M, const double* L2,
                            const double* G, const double* c_from, const double* c_to, double step, double* out) {
  { ProfScope _ps(st, KID_TAIL);
    hipLaunchKernelGGL(k_transition_tail, dim3(1), dim3(kBlock), 0, st, r, alpha, M, L2, G, c_from, c_to, step, out); }
}

void launch_posterior_eigen(hipStream_t st, int r, const double* M, const double* sqrt_lambda, double* V, double* S,
                            double* work, int* status) {
  int a_in_lds = r * r <= kLdsDoubles;
  int v_in_lds = 2 * r * r <= kLdsDoubles;
  size_t shmem = sizeof(double) * ((a_in_lds ? r * r : 0) + (v_in_lds ? r * r : 0));
  { ProfScope _ps(st, KID_EIGEN);
    hipLaunchKernelGGL(k_posterior_eigen, dim3(1), dim3(kBlock), shmem, st, r, M, sqrt_lambda, V, S, work, status, a_in_lds, v_in_lds); }
}

void launch_propose(hipStream_t st, int r, const double* alpha, const double* V, const double* S,
                    const double* inv_sqrt_lambda, const double* G, const double* Lg, const double* c,
                    const double* z, double step, double* c_out) {
  { ProfScope _ps(st, KID_PROPOSE);
    hipLaunchKernelGGL(k_propose, dim3(1), dim3(kBlock), 0, st, r, alpha, V, S, inv_sqrt_lambda, G, Lg, c, z, step, c_out); }
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
