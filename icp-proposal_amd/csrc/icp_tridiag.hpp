// icp_tridiag.hpp — the posterior KL basis by the direct route: Householder tridiagonalisation of N = D⁻¹MD⁻¹ on ONE workgroup
// (the matrix lives in registers), then one wave per eigenpair on many CUs: multisection on the Sturm sequence, the
// eigenvector of the tridiagonal matrix by twisted factorisation with a Rayleigh-quotient correction, the Householder
// reflectors applied to it, canonical sign, output.  (NonRigidIcpProposal.scala:53-56 samples from this basis.)
//
// Why not the Jacobi iteration for every rank: a sweep is r−1 dependent rounds and a warm start needs 2-4 of them (a cold one
// 7-9); the reduction to tridiagonal form is r−2 dependent steps ONCE, and everything behind it is parallel over the
// eigenpairs.  No warm start, no state carried from one decomposition to the next.
//
// Included by kernels_posterior.hip only.
#pragma once
#include <hip/hip_runtime.h>

namespace icp {
namespace tri {

#ifdef ICP_EIGEN_TIMING
#define TRI_STAMP(i) do { if (threadIdx.x == 0 && blockIdx.x == 0) g_eigen_stamps[i] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define TRI_STAMP(i)
#endif

template <int N> struct Tag { static constexpr int value = N; };

template <int CTRL> __device__ __forceinline__ double dpp_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over the 64 lanes, the same value (bit for bit) in every lane and in every wave that sums the same numbers: four DPP
// exchanges inside the rows of 16 (xor 1, xor 2, half mirror, mirror), then the four row totals through scalar registers.
// Call with all lanes active.
// LDS written by some lanes of a wave and read by others of the same wave (the LDS queue keeps a wave's accesses in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);   // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);   // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);  // row_half_mirror
  v += dpp_f64<0x140>(v);  // row_mirror
  return (readlane_f64(v, 0) + readlane_f64(v, 16)) + (readlane_f64(v, 32) + readlane_f64(v, 48));
}

// ---------------------------------------------------------------------------------------------------------------------
// Tridiagonalisation.  NW waves; lane l of wave w holds A[i][j] for i = l + 64·s (s < SI), j = w + NW·t (t < NT), the FULL
// symmetric matrix: the product A·v then accumulates inside a thread over its wave's columns, and the NW partial sums per row
// meet in LDS.  Every wave carries the Householder vector v and w = β(Av) − K·v redundantly, one row slot per lane, so a
// wave's own columns' entries v_j, w_j come out of its own registers (v_readlane) — no second exchange.  The column that is
// eliminated next travels through LDS one step ahead, before this step's update, and every wave applies the update to it
// itself.  One barrier per step (two where the partial sums are reduced in two stages: NW·SI > 8).
// Finished rows and columns are skipped by whole slots (compile-time bounds per phase of NW steps).
struct TridiagIO {
  int n;
  const double* M;            // n×n, row-major; symmetrised on load
  const double* sqrt_lambda;  // N_ij = M_ij / (sqrt_lambda_i · sqrt_lambda_j)
  double* d;                  // [n] diagonal of T
  double* e;                  // [n−1] sub-diagonal
  double* beta;               // [n] H_k = I − beta_k v_k v_kᵀ, k = 0..n−3
  double* Hv;                 // [n][64·SI] v_k, zero outside rows k+1..n−1
};

template <int NW, int SI, int NT>
__device__ __forceinline__ void tridiagonalise(const TridiagIO& a, double (&A)[SI][NT], double* part, double* psum, double* colbuf) {
  constexpr int LD = 64 * SI;
  constexpr bool TWO = NW * SI > 8;
  static_assert(64 % NW == 0, "a wave's columns must map to fixed lanes");
  const int n = a.n;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double xs[SI];
  // column 0 to everyone
  if (w == 0) {
#pragma unroll
    for (int s = 0; s < SI; ++s) colbuf[LD + l + 64 * s] = A[s][0];
  }
  __syncthreads();
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s;
    const double cb = colbuf[LD + i];
    xs[s] = i > 0 ? cb : 0.0;
    if (w == 0 && i == 0) a.d[0] = cb;
  }

  auto phase = [&](auto t0tag) {
    constexpr int T0 = decltype(t0tag)::value;
    constexpr int S0 = (NW * T0) / 64;
    constexpr int T1 = T0 + 1 < NT ? T0 + 1 : NT - 1;
    const int kend = min(NW * (T0 + 1), n - 2);
    for (int k = NW * T0; k < kend; ++k) {
      const int par = k & 1;
      const int k1 = k + 1;
      // ---- the reflector of column k
      double loc = 0.0;
#pragma unroll
      for (int s = S0; s < SI; ++s) loc = (l + 64 * s > k1) ? fma(xs[s], xs[s], loc) : loc;
      const double sig2 = wave_sum(loc);
      double xk1 = xs[S0];
#pragma unroll
      for (int s = S0 + 1; s < SI; ++s) xk1 = ((k1 >> 6) == s) ? xs[s] : xk1;
      const double x0 = readlane_f64(xk1, k1 & 63);
      double alpha = x0, beta = 0.0;
      if (sig2 != 0.0) {
        const double nrm = sqrt(fma(x0, x0, sig2));
        alpha = x0 >= 0.0 ? -nrm : nrm;
        beta = 1.0 / (alpha * (alpha - x0));
      }
      double vs[SI], ws[SI];
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        const int i = l + 64 * s;
        vs[s] = s < S0 ? 0.0 : (i > k1 ? xs[s] : (i == k1 ? x0 - alpha : 0.0));
      }
      if (w == (k & (NW - 1))) {
#pragma unroll
        for (int s = 0; s < SI; ++s) a.Hv[(size_t)k * LD + l + 64 * s] = vs[s];
        if (l == 0) { a.beta[k] = beta; a.e[k] = alpha; }
      }
      // ---- partial sums of A·v over this wave's columns; the next column to everyone
      double vj[NT];
#pragma unroll
      for (int t = T0; t < NT; ++t) vj[t] = readlane_f64(vs[(NW * t) >> 6], ((NW * t) & 63) + w);
      double* mypart = part + (TWO ? 0 : par * NW * LD) + w * LD;
#pragma unroll
      for (int s = S0; s < SI; ++s) {
        double acc = 0.0;
#pragma unroll
        for (int t = T0; t < NT; ++t) acc = fma(A[s][t], vj[t], acc);
        mypart[l + 64 * s] = acc;
      }
      if (w == (k1 & (NW - 1))) {
        const bool first = (k1 / NW) == T0;
#pragma unroll
        for (int s = S0; s < SI; ++s) colbuf[par * LD + l + 64 * s] = first ? A[s][T0] : A[s][T1];
      }
      __syncthreads();
      double p[SI];
      if constexpr (TWO) {
        constexpr int CH = LD / NW;  // rows summed by one wave
        if (l < CH) {
          const int i = w * CH + l;
          double sum = part[i];
#pragma unroll
          for (int ww = 1; ww < NW; ++ww) sum += part[ww * LD + i];
          psum[i] = sum;
        }
        __syncthreads();
#pragma unroll
        for (int s = S0; s < SI; ++s) p[s] = beta * psum[l + 64 * s];
      } else {
        const double* pp = part + par * NW * LD;
#pragma unroll
        for (int s = S0; s < SI; ++s) {
          double sum = pp[l + 64 * s];
#pragma unroll
          for (int ww = 1; ww < NW; ++ww) sum += pp[ww * LD + l + 64 * s];
          p[s] = beta * sum;
        }
      }
      // ---- w = p − K v, rows above the active block frozen
      double pv = 0.0;
#pragma unroll
      for (int s = S0; s < SI; ++s) pv = fma(p[s], vs[s], pv);
      const double K = 0.5 * beta * wave_sum(pv);
#pragma unroll
      for (int s = 0; s < SI; ++s) ws[s] = (s >= S0 && l + 64 * s > k) ? fma(-K, vs[s], p[s]) : 0.0;
      // ---- A ← A − v wᵀ − w vᵀ on the slots still alive
#pragma unroll
      for (int t = T0; t < NT; ++t) {
        const double wj = readlane_f64(ws[(NW * t) >> 6], ((NW * t) & 63) + w);
#pragma unroll
        for (int s = S0; s < SI; ++s) A[s][t] = fma(-vs[s], wj, fma(-ws[s], vj[t], A[s][t]));
      }
      // ---- column k+1 after the update: the next x, and d[k+1]
      double vk1 = vs[S0], wk1 = ws[S0];
#pragma unroll
      for (int s = S0 + 1; s < SI; ++s) { const bool here = (k1 >> 6) == s; vk1 = here ? vs[s] : vk1; wk1 = here ? ws[s] : wk1; }
      vk1 = readlane_f64(vk1, k1 & 63);
      wk1 = readlane_f64(wk1, k1 & 63);
#pragma unroll
      for (int s = S0; s < SI; ++s) {
        const int i = l + 64 * s;
        const double cn = fma(-vs[s], wk1, fma(-ws[s], vk1, colbuf[par * LD + i]));
        xs[s] = i > k1 ? cn : 0.0;
        if (w == 0 && i == k1) a.d[k1] = cn;
      }
    }
  };
  // phases T0 = 0 .. NT−1, each with its own compile-time bounds (stops where the matrix ends)
  auto run = [&](auto self, auto t0tag) -> void {
    constexpr int T0 = decltype(t0tag)::value;
    if (NW * T0 < n - 2) {
      phase(t0tag);
      if constexpr (T0 + 1 < NT) self(self, Tag<T0 + 1>{});
    }
  };
  run(run, Tag<0>{});
  // what is left: e[n−2] = A[n−1][n−2] (the last x), d[n−1]
  if (w == 0) {
#pragma unroll
    for (int s = 0; s < SI; ++s)
      if (l + 64 * s == n - 1) a.e[n - 2] = xs[s];
  }
#pragma unroll
  for (int s = 0; s < SI; ++s)
#pragma unroll
    for (int t = 0; t < NT; ++t)
      if (l + 64 * s == n - 1 && w + NW * t == n - 1) a.d[n - 1] = A[s][t];
}

template <int NW, int SI, int NT>
__global__ void __launch_bounds__(NW * 64) k_tridiag(TridiagIO a) {
  constexpr int LD = 64 * SI;
  constexpr bool TWO = NW * SI > 8;
  __shared__ double part[(TWO ? 1 : 2) * NW * LD];
  __shared__ double psum[LD];
  __shared__ double colbuf[2 * LD];
  const int n = a.n;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  double A[SI][NT];
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s;
    const double sli = i < n ? a.sqrt_lambda[i] : 1.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = w + NW * t;
      double v = 0.0;
      if (i < n && j < n) v = 0.5 * (a.M[(size_t)j * n + i] + a.M[(size_t)i * n + j]) / (sli * a.sqrt_lambda[j]);
      A[s][t] = v;
    }
  }
  EIG_STAMP(0);
  tridiagonalise<NW, SI, NT>(a, A, part, psum, colbuf);
  EIG_STAMP(1);
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per eigenpair.
struct TriSolveIO {
  int n;
  const double* d;
  const double* e;
  const double* beta;
  const double* Hv;   // [n][64·SI]
  double* V;          // n×n: column `rank` = eigenvector
  double* Vt;         // n×n: row `rank`
  double* S;          // [n] = 1/mu, descending
  double* mu;         // [n] scratch: the eigenvalues, ascending (the last wave checks the gaps)
  int* sync;          // [2] {waves finished, trouble flags}: zero between launches
  int* status;        // status[0]: 0 ok, 2 = not trustworthy (gaps below resolution / non-finite); status[-1]: 0
  int* host_status;   // optional pinned copy
  int* done_word;     // optional completion word (agent scope)
  int done_value;
};

constexpr int kTriPasses = 5;       // multisection passes of 64 points: the bracket shrinks 65× per pass
constexpr int kTriRounds = 2;       // twisted factorisations (each followed by a Rayleigh-quotient correction)
constexpr int kTriMaxN = 256;

// number of eigenvalues of the (scaled) tridiagonal matrix below x: sign changes of the leading principal minors, by the
// three-term recurrence (one dependent fma per row), rescaled every eighth row
__device__ __forceinline__ int sturm_count(const double* __restrict__ ds, const double* __restrict__ e2, int n, double x) {
  double p0 = 1.0, p1 = ds[0] - x;
  if (p1 == 0.0) p1 = -1e-300;
  int cnt = (unsigned)__double2hiint(p1) >> 31;
  for (int i = 1; i < n; ++i) {
    double p2 = fma(ds[i] - x, p1, -(e2[i - 1] * p0));
    if (p2 == 0.0) p2 = p1 < 0.0 ? 1e-300 : -1e-300;
    cnt += (unsigned)(__double2hiint(p2) ^ __double2hiint(p1)) >> 31;
    p0 = p1;
    p1 = p2;
    if ((i & 7) == 0) {
      const int ex = max(__builtin_amdgcn_frexp_exp(p0), __builtin_amdgcn_frexp_exp(p1));
      p0 = ldexp(p0, -ex);
      p1 = ldexp(p1, -ex);
    }
  }
  return cnt;
}

template <int SI>
__global__ void __launch_bounds__(256) k_tri_solve(TriSolveIO a) {
  constexpr int LD = 64 * SI;
  __shared__ double ds[kTriMaxN], es[kTriMaxN], e2[kTriMaxN], bet[kTriMaxN];
  __shared__ double scr[4][4 * kTriMaxN];
  __shared__ double red[8];
  const int n = a.n;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x * 4 + w;  // this wave's eigenvalue (ascending), rank j of the output
  TRI_STAMP(8);
  // ---- the matrix, scaled to norm <= 1 (Gershgorin)
  double glo = 1e300, ghi = -1e300;
  for (int i = tid; i < n; i += 256) {
    const double di = a.d[i], el = i > 0 ? fabs(a.e[i - 1]) : 0.0, er = i < n - 1 ? fabs(a.e[i]) : 0.0;
    glo = fmin(glo, di - el - er);
    ghi = fmax(ghi, di + el + er);
  }
  for (int o = 32; o > 0; o >>= 1) { glo = fmin(glo, __shfl_xor(glo, o, 64)); ghi = fmax(ghi, __shfl_xor(ghi, o, 64)); }
  if (l == 0) { red[w] = glo; red[4 + w] = ghi; }
  __syncthreads();
  glo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3]));
  ghi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
  const double anorm = fmax(fabs(glo), fabs(ghi));
  const double inv = anorm > 0.0 ? 1.0 / anorm : 1.0;
  for (int i = tid; i < n; i += 256) {
    ds[i] = a.d[i] * inv;
    const double ee = i < n - 1 ? a.e[i] * inv : 0.0;
    es[i] = ee;
    e2[i] = fmax(ee * ee, 1e-280);
    bet[i] = i < n - 2 ? a.beta[i] : 0.0;
  }
  __syncthreads();
  if (j >= n) return;  // (no barrier below)
  TRI_STAMP(9);
  // ---- multisection: 64 points per pass inside the bracket, the count tells on which side of each point eigenvalue j lies
  double lo = glo * inv - 1e-9, hi = ghi * inv + 1e-9;
  for (int pass = 0; pass < kTriPasses; ++pass) {
    const double h = (hi - lo) * (1.0 / 65.0);
    const double x = fma((double)(l + 1), h, lo);
    const int c = sturm_count(ds, e2, n, x);
    const unsigned long long above = __ballot(c >= j + 1);
    if (above == 0ull) {
      lo = readlane_f64(x, 63);
    } else {
      const int f = __ffsll((long long)above) - 1;
      hi = readlane_f64(x, f);
      if (f > 0) lo = readlane_f64(x, f - 1);
    }
  }
  TRI_STAMP(10);
  // ---- eigenvector of T by twisted factorisation: lane 0 factors T − λI from the top, lane 1 from the bottom (the same
  // instruction stream), the twist goes where |γ| is smallest; (T − λI) z = γ_r e_r, and γ_r/‖z‖² corrects λ
  double* Dq = scr[w];               // [2][n]: D⁺ | D⁻
  double* Lq = scr[w] + 2 * n;       // [2][n]: L⁺ | U⁻
  double* zb = Dq;                   // z overwrites D⁺ once γ is known
  double lam = 0.5 * (lo + hi);
  double znorm2 = 1.0;
  bool trouble = false;
  for (int round = 0; round < kTriRounds; ++round) {
    if (l < 2) {
      const int dir = l;
      int i = dir ? n - 1 : 0;
      const int st = dir ? -1 : 1;
      double D = ds[i] - lam;
      Dq[dir * n + i] = D;
      for (int k = 0; k < n - 1; ++k) {
        const int ie = dir ? i - 1 : i;
        if (fabs(D) < 1e-150) D = D < 0.0 ? -1e-150 : 1e-150;
        const double ee = es[ie];
        const double L = ee / D;
        Lq[dir * n + ie] = L;
        i += st;
        D = (ds[i] - lam) - L * ee;
        Dq[dir * n + i] = D;
      }
    }
    wave_lds_sync();
    double gbest = 1e300;
    int ibest = 0;
    for (int i = l; i < n; i += 64) {
      const double g = fabs((Dq[i] + Dq[n + i]) - (ds[i] - lam));
      if (g < gbest) { gbest = g; ibest = i; }
    }
    for (int o = 32; o > 0; o >>= 1) {
      const double og = __shfl_xor(gbest, o, 64);
      const int oi = __shfl_xor(ibest, o, 64);
      if (og < gbest || (og == gbest && oi < ibest)) { gbest = og; ibest = oi; }
    }
    const int rt = ibest;
    const double gamma = (Dq[rt] + Dq[n + rt]) - (ds[rt] - lam);
    wave_lds_sync();
    if (l < 2) {
      const int dir = l;
      const int cnt = dir ? n - 1 - rt : rt;
      double zc = 1.0;
      int idx = rt;
      for (int k = 0; k < cnt; ++k) {
        const int ie = dir ? idx : idx - 1;
        zc = -(Lq[dir * n + ie] * zc);
        idx += dir ? 1 : -1;
        zb[idx] = zc;
      }
      if (dir == 0) zb[rt] = 1.0;
    }
    wave_lds_sync();
    double zz = 0.0;
    for (int i = l; i < n; i += 64) zz = fma(zb[i], zb[i], zz);
    znorm2 = wave_sum(zz);
    const double lam_new = lam + gamma / znorm2;
    if (lam_new >= lo && lam_new <= hi) lam = lam_new;
    if (!(znorm2 > 0.0) || !(znorm2 < 1e300)) trouble = true;
    if (round + 1 < kTriRounds) wave_lds_sync();
  }
  TRI_STAMP(11);
  // ---- back-transformation: z ← H_0 H_1 ··· H_{n−3} z, one row slot per lane, reflectors fetched four ahead
  double zs[SI];
  {
    const double sc = 1.0 / sqrt(znorm2);
#pragma unroll
    for (int s = 0; s < SI; ++s) zs[s] = l + 64 * s < n ? zb[l + 64 * s] * sc : 0.0;
  }
  {
    constexpr int PF = 4;
    double vq[PF][SI];
#pragma unroll
    for (int q = 0; q < PF; ++q) {
      const int k = n - 3 - q;
#pragma unroll
      for (int s = 0; s < SI; ++s) vq[q][s] = k >= 0 ? a.Hv[(size_t)k * LD + l + 64 * s] : 0.0;
    }
    for (int kb = n - 3; kb >= 0; kb -= PF) {
#pragma unroll
      for (int q = 0; q < PF; ++q) {
        const int k = kb - q;
        double v[SI];
#pragma unroll
        for (int s = 0; s < SI; ++s) v[s] = vq[q][s];
        const int kn = k - PF;
#pragma unroll
        for (int s = 0; s < SI; ++s) vq[q][s] = kn >= 0 ? a.Hv[(size_t)kn * LD + l + 64 * s] : 0.0;
        if (k >= 0) {
          double dp = 0.0;
#pragma unroll
          for (int s = 0; s < SI; ++s) dp = fma(v[s], zs[s], dp);
          const double f = bet[k] * wave_sum(dp);
#pragma unroll
          for (int s = 0; s < SI; ++s) zs[s] = fma(-f, v[s], zs[s]);
        }
      }
    }
  }
  TRI_STAMP(12);
  // ---- canonical sign (largest-|.| component positive, the first among equals), output
  double bv = -1.0;
  int bi = 0x7fffffff;
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s;
    const double av = fabs(zs[s]);
    if (i < n && av > bv) { bv = av; bi = i; }
  }
  for (int o = 32; o > 0; o >>= 1) {
    const double ov = __shfl_xor(bv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  double lead = zs[0];
#pragma unroll
  for (int s = 1; s < SI; ++s) lead = (bi >> 6) == s ? zs[s] : lead;
  lead = readlane_f64(lead, bi & 63);
  const double sgn = lead < 0.0 ? -1.0 : 1.0;
  const double muj = lam * anorm;
  if (!(bv >= 0.0) || !(muj > 0.0)) trouble = true;
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s;
    if (i < n) {
      const double v = zs[s] * sgn;
      a.V[(size_t)i * n + j] = v;
      a.Vt[(size_t)j * n + i] = v;
    }
  }
  if (l == 0) { a.S[j] = 1.0 / muj; a.mu[j] = muj; }
  TRI_STAMP(13);
  // ---- the last wave to finish checks that the eigenvalues are told apart and publishes the status
  __threadfence();
  int last = 0;
  if (l == 0) {
    if (trouble) atomicOr(a.sync + 1, 1);
    last = atomicAdd(a.sync, 1) == n - 1;
  }
  last = __builtin_amdgcn_readfirstlane(last);
  if (!last) return;
  __threadfence();
  int bad = 0;
  for (int i = l; i < n - 1; i += 64) {
    const double m0 = __hip_atomic_load(a.mu + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m1 = __hip_atomic_load(a.mu + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(m1 - m0 > 1e-10 * anorm)) bad = 1;
  }
  bad = __any(bad);
  if (l == 0) {
    if (__hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bad = 1;
    const int st = bad ? 2 : 0;
    a.status[0] = st;
    a.status[-1] = 0;
    a.sync[0] = 0;
    a.sync[1] = 0;
    if (a.host_status) __hip_atomic_store(a.host_status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.done_word) __hip_atomic_store(a.done_word, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}

}  // namespace tri
}  // namespace icp
