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
#ifdef ICP_TRI_CYCLES  // developer aid: shader-clock cycles per part of a reduction step, summed over the steps (wave 0)
#define TRI_CYC(i) do { const long long t_ = (long long)__builtin_amdgcn_s_memtime(); cyc[i] += t_ - tprev; tprev = t_; } while (0)
#else
#define TRI_CYC(i)
#endif

template <int N> struct Tag { static constexpr int value = N; };

template <int CTRL, int ROWMASK = 0xf> __device__ __forceinline__ double dpp_f64(double v) {
  if constexpr (ROWMASK == 0xf) {  // every lane receives a value: no "old" operand to set up
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
  } else {  // lanes of the masked rows receive zero
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROWMASK, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROWMASK, 0xf, true);
    return __hiloint2double(hi, lo);
  }
}
__device__ __forceinline__ double readlane_f64(double v, int lane) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), lane), __builtin_amdgcn_readlane(__double2loint(v), lane));
}
// sum over the 64 lanes, the same value (bit for bit) in every lane and in every wave that sums the same numbers: four DPP
// exchanges inside the rows of 16 (xor 1, xor 2, half mirror, mirror), then the four row totals through scalar registers.
// Call with all lanes active.
// square root and reciprocal to a few ulp from the hardware seeds (v_rsq_f64 / v_rcp_f64) and Newton steps: the reflector's
// norm and 2/vᵀv need no correct rounding, and the library sequences are three times as long.  Arguments well inside the range.
__device__ __forceinline__ double fast_sqrt(double x) {
  double y = __builtin_amdgcn_rsq(x);
  const double h = 0.5 * x;
  y = fma(y, fma(-h * y, y, 0.5), y);
  y = fma(y, fma(-h * y, y, 0.5), y);
  const double g = x * y;
  return fma(0.5 * y, fma(-g, g, x), g);
}
__device__ __forceinline__ double fast_rcp(double x) {
  double y = __builtin_amdgcn_rcp(x);
  y = fma(y, fma(-x, y, 1.0), y);
  y = fma(y, fma(-x, y, 1.0), y);
  return y;
}
// LDS written by some lanes of a wave and read by others of the same wave (the LDS queue keeps a wave's accesses in order)
__device__ __forceinline__ void wave_lds_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
// workgroup barrier for data exchanged through LDS only: does not wait for the global stores in flight (the reflectors on their
// way to memory are read by the next launch)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_f64<0xB1>(v);        // quad_perm [1,0,3,2]
  v += dpp_f64<0x4E>(v);        // quad_perm [2,3,0,1]
  v += dpp_f64<0x141>(v);       // row_half_mirror
  v += dpp_f64<0x140>(v);       // row_mirror: every lane of a row of 16 holds the row's total
  v += dpp_f64<0x142, 0xa>(v);  // row_bcast15 into rows 1 and 3 (bound_ctrl: the other rows add zero)
  v += dpp_f64<0x143, 0xc>(v);  // row_bcast31 into rows 2 and 3: lane 63 holds the total
  return readlane_f64(v, 63);
}

// NV sums over the 64 lanes at once, the same totals in every lane: the four in-row DPP exchanges stage by stage for all values (NV
// independent chains in flight instead of one: a lone wave_sum is ≈ 250 cycles of dependent DPP steps and their hazards), then
// every value's four row totals meet in ONE register — lane 16g + q takes value q's total of row g —, two cross-row steps on that
// register (lane-wise exchanges with the rows 16 and 32 lanes away), and lane q holds value q's sum.  Call with all lanes active; NV <= 16.
template <int NV> __device__ __forceinline__ void wave_sum_many(double (&v)[NV]) {
  static_assert(NV >= 1 && NV <= 16, "one lane of a row of 16 per value");
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0xB1>(v[q]);
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x4E>(v[q]);
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x141>(v[q]);
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] += dpp_f64<0x140>(v[q]);
  const int lq = threadIdx.x & 15;
  double t = 0.0;
#pragma unroll
  for (int q = 0; q < NV; ++q) t = lq == q ? v[q] : t;
  t += __shfl_xor(t, 16, 64);  // (lane by lane across the rows: the row-broadcast DPP forms hand on ONE lane's value)
  t += __shfl_xor(t, 32, 64);  // every lane 16g + q now holds value q's sum
#pragma unroll
  for (int q = 0; q < NV; ++q) v[q] = readlane_f64(t, q);
}

// a number as mantissa × 2^exponent (products of hundreds of factors — the eigenvector entries of the twisted factorisation — without
// overflow or underflow along the way)
struct ScaledF64 {
  double m;
  int e;
};
__device__ __forceinline__ ScaledF64 scaled_of(double x) {
  const int ex = __builtin_amdgcn_frexp_exp(x);
  return ScaledF64{ldexp(x, -ex), ex};
}
__device__ __forceinline__ ScaledF64 scaled_mul(const ScaledF64& a, const ScaledF64& b) {
  const double m = a.m * b.m;
  const int ex = __builtin_amdgcn_frexp_exp(m);
  return ScaledF64{ldexp(m, -ex), a.e + b.e + ex};
}
__device__ __forceinline__ ScaledF64 scaled_shfl_up(const ScaledF64& a, int d) { return ScaledF64{__shfl_up(a.m, d, 64), __shfl_up(a.e, d, 64)}; }
__device__ __forceinline__ ScaledF64 scaled_shfl_down(const ScaledF64& a, int d) { return ScaledF64{__shfl_down(a.m, d, 64), __shfl_down(a.e, d, 64)}; }

// ---------------------------------------------------------------------------------------------------------------------
// Tridiagonalisation.  NW waves; lane l of wave w holds A[i][j] for i = l + 64·s (s < SI), j = w + NW·t (t < NT), the FULL
// symmetric matrix: a product A·x accumulates inside a thread over its wave's columns, and the NW partial sums per row meet
// in LDS.  Every wave carries the Householder vector v, w = β(Av) − K·v and the next column x redundantly, one row slot per
// lane; the entries at its own columns (v_j, w_j, x_j) come back as LDS broadcast reads from a wave-private copy — the VALU
// does the multiply-adds only.
//
// One pass over the registers per step: A ← A − v wᵀ − w vᵀ and, entry by entry, the NEXT step's product A·x.  That works because
// (i) the column eliminated next travels through LDS one step ahead, before the update, and every wave updates it itself
// (x' = c − v w_{k+1} − w v_{k+1}); (ii) A·v = A·x − α·A e_{k+1} and vᵀAv = xᵀAx − 2α(Ax)_{k+1} + α²A_{k+1,k+1}: the products are
// formed with x, before the reflector (norm, square root, reciprocal — computed beside the pass) is known.
// Per step: one barrier (two where the partial sums are reduced in two stages, SI >= 2), one wave-private LDS round trip, one
// wave reduction (xᵀAx) on the critical path.  Finished rows and columns are skipped by whole slots (compile-time bounds per
// phase of NW steps).
struct TridiagIO {
  int n;
  const double* M;            // n×n, row-major; symmetrised on load
  const double* sqrt_lambda;  // N_ij = M_ij / (sqrt_lambda_i · sqrt_lambda_j)
  double* d;                  // [n] diagonal of T
  double* e;                  // [n−1] sub-diagonal
  double* beta;               // [n] H_k = I − beta_k v_k v_kᵀ, k = 0..n−3
  double* Hv;                 // [n][64·SI] v_k, zero outside rows k+1..n−1
  double* Nout;               // [n][n] (optional) the matrix itself, for the refinement step
};

template <int NW, int SI> struct TridiagLds {
  static constexpr int LD = 64 * SI;
  static constexpr bool TWO = SI >= 2;
  static constexpr int oPart = 0;                                 // [TWO ? 1 : 2][NW][LD] partial sums of A·x
  static constexpr int oPsum = oPart + (TWO ? 1 : 2) * NW * LD;   // [LD] their total (two-stage form)
  static constexpr int oCol = oPsum + LD;                         // [2][LD] the column eliminated next, before the update
  static constexpr int oScal = oCol + 2 * LD;                     // [2][NW] per-wave parts of xᵀAx
  static constexpr int oPriv = oScal + 2 * NW;                    // [NW][3][LD] wave-private: x/v (two parities), w
  static constexpr int doubles = oPriv + NW * 3 * LD;
};

// Index space: the matrix occupies the LAST n of the LD = 64·SI row/column positions (position = index + off, off = LD − n),
// so that the partly filled row slot is the one eliminated — and dropped from the passes — first.  TOFF = column slots that
// are empty for every n of the configuration (they get no registers).
template <int NW, int SI, int NT, int TOFF>
__device__ __forceinline__ void tridiagonalise(const TridiagIO& a, double (&A)[SI][NT], const double (&col0)[SI], double d0, double* lds) {
  using L = TridiagLds<NW, SI>;
  constexpr int LD = L::LD;
  constexpr bool TWO = L::TWO;
  static_assert(NW * (NT + TOFF) == LD, "column slots must cover the index space");
  double* part = lds + L::oPart;
  double* psum = lds + L::oPsum;
  double* colbuf = lds + L::oCol;
  double* scal = lds + L::oScal;
  const int n = a.n, off = LD - n;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // With eight waves (two per SIMD) the chain of scalars and lane vectors of a step — the same in every wave — would be paid twice per
  // SIMD: there ONE wave carries it (LEAD), the others wait at a barrier and take v, w, x' from one shared copy in LDS.
  constexpr bool LEAD = NW == 8;
  const bool lead = !LEAD || w == 0;
  double* vb = lds + L::oPriv + (LEAD ? 0 : w * 3 * LD);  // vb[par·LD + i]: x of step k (v once its entry k+1 is fixed), k & 1 = par
  double* wb = vb + 2 * LD;

#ifdef ICP_TRI_CYCLES
  long long cyc[10] = {}, tprev = 0;
#endif
  // Per step the lane-form vectors (one row slot per lane) are: x, the column eliminated now (rows > k); v; w; x', the next column.
  // They stay in registers from one step to the next — except in the largest configuration (SI = 4: no registers to spare),
  // where they live in the wave-private LDS arrays vb (x / x' by step parity) and wb.
  constexpr bool VLDS = LEAD || SI >= 4;
  double xs[SI];    // x (registers form)
  double x0, sig2;  // its first entry x_{k+1}; Σ_{i>k+1} x_i²
  // ---- the pass over the registers: optional rank-2 update, then acc = A·x'; publishes the partial sums, this wave's part of
  // x'ᵀAx' and column `cnext` of the updated matrix.  Row slots below S0 are finished (compile time); columns are taken four at a
  // time — their entries of v, w, x' come from the lanes that hold those positions as rows (v_readlane), fetched together ahead
  // of the multiply-adds — and a group whose columns are all <= kdone is skipped (uniform branch).
  auto pass = [&](auto s0tag, auto updtag, int vpar, int xpar, int cnext, int kdone, const double (&vr)[SI], const double (&wr)[SI],
                  const double (&xr)[SI]) {
    constexpr int S0 = decltype(s0tag)::value;
    constexpr bool UPD = decltype(updtag)::value != 0;
    constexpr int CH = LEAD ? 2 : 4;  // columns per group (eight waves: two — 413 -> 403 µs at rank 200, 227 -> 218 at 150; one or three: slower)
    double acc[SI], vs[SI], ws[SI], xl[SI];
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      acc[s] = 0.0;
      if constexpr (VLDS) {
        vs[s] = UPD && s >= S0 ? vb[vpar * LD + l + 64 * s] : 0.0;
        ws[s] = UPD && s >= S0 ? wb[l + 64 * s] : 0.0;
        xl[s] = s >= S0 ? vb[xpar * LD + l + 64 * s] : 0.0;
      } else {
        vs[s] = vr[s]; ws[s] = wr[s]; xl[s] = xr[s];
      }
    }
    const bool owner = w == (cnext & (NW - 1));
    const int tc = cnext / NW - TOFF;
    // Eight waves (ranks above 128): the entries of v, w, x' at the wave's columns are LDS broadcast reads, requested ONE GROUP AHEAD of
    // the multiply-adds that use them (two buffers by group parity) — with two waves per SIMD nobody else covers their latency.  Read
    // where they are used (ranks up to 192, round 2) the reduction took 304 µs at rank 150, as v_readlane pairs into SGPRs — which they
    // overflow: the four-slot configuration's form until round 3 — 300 and 477-489 µs at rank 200; one group ahead: 234 µs at rank 150.
    // The four-slot configuration (ranks 193..200) has no registers for two groups' values while three or four row slots are alive: it
    // takes this path once two are finished (rank 200: 411 µs; from one finished slot on 451, always 554, only with one live slot 426)
    // and v_readlane before.  (Measured WITHOUT the cycle counters of ICP_TRI_CYCLES: their waits for the LDS queue penalise this.)
#ifndef ICP_TRI_LDS_LIVE4
#define ICP_TRI_LDS_LIVE4 2
#endif
    constexpr int kLdsLive = SI <= 3 ? SI : ICP_TRI_LDS_LIVE4;  // (developer sweep: -DICP_TRI_LDS_LIVE4=1..4)
    if constexpr (LEAD && VLDS && SI - S0 <= kLdsLive) {
      double vj[2][CH], wj[2][CH], xj[2][CH];
      bool prev_live = false;
#pragma unroll
      for (int c0 = 0; c0 < NT; c0 += CH) {
        const int b = (c0 / CH) & 1;
        const bool live = NW * (c0 + CH + TOFF) > kdone + 1;  // (uniform; once a group is live every later one is)
        if (live) {
          if (!prev_live) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
              const int j = w + NW * ((c0 + q < NT ? c0 + q : NT - 1) + TOFF);
              xj[b][q] = vb[xpar * LD + j];
              if constexpr (UPD) { vj[b][q] = vb[vpar * LD + j]; wj[b][q] = wb[j]; }
            }
          }
          if (c0 + CH < NT) {
#pragma unroll
            for (int q = 0; q < CH; ++q) {
              const int j = w + NW * ((c0 + CH + q < NT ? c0 + CH + q : NT - 1) + TOFF);
              xj[b ^ 1][q] = vb[xpar * LD + j];
              if constexpr (UPD) { vj[b ^ 1][q] = vb[vpar * LD + j]; wj[b ^ 1][q] = wb[j]; }
            }
          }
#pragma unroll
          for (int q = 0; q < CH; ++q) {
            const int t = c0 + q;
            if (t < NT) {
              if constexpr (UPD) {
#pragma unroll
                for (int s = S0; s < SI; ++s) A[s][t] = fma(-vs[s], wj[b][q], fma(-ws[s], vj[b][q], A[s][t]));
              }
#pragma unroll
              for (int s = S0; s < SI; ++s) acc[s] = fma(A[s][t], xj[b][q], acc[s]);
            }
          }
        }
        prev_live = live;
      }
    } else
#pragma unroll
    for (int c0 = 0; c0 < NT; c0 += CH) {
      if (SI == 1 || NW * (c0 + CH + TOFF) > kdone + 1) {
        double vj[CH], wj[CH], xj[CH];
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int t = c0 + q < NT ? c0 + q : NT - 1;
          const int sj = (NW * (t + TOFF)) >> 6;  // where column j = w + NW·(t + TOFF) sits as a row (constants once unrolled)
          const int lj = ((NW * (t + TOFF)) & 63) + w;
          if constexpr (VLDS && SI <= 3) {  // (SI = 4: no registers for the fetched values — the v_readlane results live in SGPRs)
            // (the vectors are in LDS anyway: broadcast reads — 4 cycles of the CU's LDS pipe each — beside the multiply-adds,
            // which are what the SIMDs are short of with two waves each; the other wave covers the latency)
            const int j = w + NW * (t + TOFF);
            xj[q] = vb[xpar * LD + j];
            if constexpr (UPD) { vj[q] = vb[vpar * LD + j]; wj[q] = wb[j]; }
          } else {
            xj[q] = readlane_f64(xl[sj], lj);
            if constexpr (UPD) { vj[q] = readlane_f64(vs[sj], lj); wj[q] = readlane_f64(ws[sj], lj); }
          }
        }
#pragma unroll
        for (int q = 0; q < CH; ++q) {
          const int t = c0 + q;
          if (t < NT) {
            if constexpr (UPD) {
#pragma unroll
              for (int s = S0; s < SI; ++s) A[s][t] = fma(-vs[s], wj[q], fma(-ws[s], vj[q], A[s][t]));
            }
#pragma unroll
            for (int s = S0; s < SI; ++s) acc[s] = fma(A[s][t], xj[q], acc[s]);
          }
        }
      }
    }
    if constexpr (UPD) TRI_CYC(5);
    // the column eliminated after this one leaves the registers of the wave that holds it: a binary search over the column
    // slots (uniform branches), so that every store names its register
    if (owner) {
      auto publish = [&](auto self, auto lotag, auto hitag) -> void {
        constexpr int lo = decltype(lotag)::value, hi = decltype(hitag)::value;
        if constexpr (hi - lo == 1) {
#pragma unroll
          for (int s = S0; s < SI; ++s) {
            double val = A[s][lo];
            asm volatile("" : "+v"(val));  // (keeps the leaves apart: merged, they become an indexed copy of A in scratch memory)
            colbuf[xpar * LD + l + 64 * s] = val;
          }
        } else {
          constexpr int mid = (lo + hi) / 2;
          if (tc < mid) self(self, lotag, Tag<mid>{}); else self(self, Tag<mid>{}, hitag);
        }
      };
      publish(publish, Tag<0>{}, Tag<NT>{});
    }
    double* mypart = part + (TWO ? 0 : xpar * NW * LD) + w * LD;
    double q = 0.0;
#pragma unroll
    for (int s = S0; s < SI; ++s) {
      mypart[l + 64 * s] = acc[s];
      q = fma(xl[s], acc[s], q);
    }
    q = wave_sum(q);
    if (l == 0) scal[xpar * NW + w] = q;
    if constexpr (UPD) TRI_CYC(6);
  };

  // ---- the first column (position off < 64; handed in by the caller, one row slot per lane, with its diagonal entry) and its product
  {
    double loc = 0.0;
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      const int i = l + 64 * s;
      const double cb = col0[s];
      xs[s] = i > off ? cb : 0.0;
      if constexpr (VLDS) { if (lead) vb[(off & 1) * LD + i] = xs[s]; }
      loc = i > off + 1 ? fma(cb, cb, loc) : loc;
    }
    if constexpr (LEAD) lds_barrier(); else if constexpr (VLDS) wave_lds_sync();
    if (w == 0 && l == 0) a.d[0] = d0;
    {
      double t = xs[0];
#pragma unroll
      for (int s = 1; s < SI; ++s) t = ((off + 1) >> 6) == s ? xs[s] : t;
      x0 = readlane_f64(t, (off + 1) & 63);
    }
    sig2 = wave_sum(loc);
    const double zero[SI] = {};
    pass(Tag<0>{}, Tag<0>{}, 0, off & 1, off + 1, off - 1, zero, zero, xs);
  }

  auto phase = [&](auto s0tag) {
    constexpr int S0 = decltype(s0tag)::value;
    const int kbeg = max(64 * S0, off), kend = min(64 * (S0 + 1), off + n - 2);
    for (int k = kbeg; k < kend; ++k) {  // (positions: the step eliminates column k, index k − off)
      const int par = k & 1, k1 = k + 1;
#ifdef ICP_TRI_CYCLES
      tprev = (long long)__builtin_amdgcn_s_memtime();
#endif
      // ---- the reflector of column k: alpha = ∓‖x‖, v = x − alpha e_{k+1}, beta = 2/vᵀv
      double alpha = x0, beta = 0.0;
      if (lead && sig2 != 0.0) {
        const double s2 = fma(x0, x0, sig2);
        const double nrm = fast_sqrt(s2);
        alpha = x0 >= 0.0 ? -nrm : nrm;
        beta = fast_rcp(fma(nrm, fabs(x0), s2));
      }
      const double vk1 = x0 - alpha;
      TRI_CYC(0);
      lds_barrier();
      TRI_CYC(1);
      if constexpr (TWO) {
        constexpr int CHR = LD / NW;  // rows summed by one wave
        if (l < CHR) {
          const int i = w * CHR + l;
          double tq[NW];
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) tq[ww] = part[ww * LD + i];
#pragma unroll
          for (int h = NW / 2; h > 0; h >>= 1)
#pragma unroll
            for (int ww = 0; ww < h; ++ww) tq[ww] += tq[ww + h];
          psum[i] = tq[0];
        }
        lds_barrier();
      }
      double vs[SI], ws[SI], xnew[SI];
      double lead_loc = 0.0;
#pragma unroll
      for (int s = 0; s < SI; ++s) { vs[s] = 0.0; ws[s] = 0.0; xnew[s] = 0.0; }
      if (lead) {
      // every LDS value of this section is requested before the first one is used (−2…4 % of the reduction at ranks 100 and 150) —
      // where the registers allow: not with four live row slots (ranks 193..200, first eight steps: the matrix alone takes 200 VGPRs)
      constexpr bool kHoist = SI - S0 <= 3;
      double sm_[SI], c_[SI], xi_[SI];
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        const int i = l + 64 * s;
        sm_[s] = 0.0; c_[s] = 0.0; xi_[s] = 0.0;
        if (!kHoist || s < S0) continue;
        if constexpr (TWO) {
          sm_[s] = psum[i];
        } else {
          const double* pp = part + par * NW * LD + i;
          double tq[NW];
#pragma unroll
          for (int ww = 0; ww < NW; ++ww) tq[ww] = pp[ww * LD];
#pragma unroll
          for (int h = NW / 2; h > 0; h >>= 1)
#pragma unroll
            for (int ww = 0; ww < h; ++ww) tq[ww] += tq[ww + h];
          sm_[s] = tq[0];
        }
        c_[s] = colbuf[par * LD + i];
        if constexpr (VLDS) xi_[s] = vb[par * LD + i]; else xi_[s] = xs[s];
      }
      double Axk1;
      if constexpr (TWO) {
        Axk1 = psum[k1];
      } else {
        const double* pp = part + par * NW * LD + k1;
        double tq[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tq[ww] = pp[ww * LD];
#pragma unroll
        for (int h = NW / 2; h > 0; h >>= 1)
#pragma unroll
          for (int ww = 0; ww < h; ++ww) tq[ww] += tq[ww + h];
        Axk1 = tq[0];
      }
      double xAx;
      {
        double tq[NW];
#pragma unroll
        for (int ww = 0; ww < NW; ++ww) tq[ww] = scal[par * NW + ww];
#pragma unroll
        for (int h = NW / 2; h > 0; h >>= 1)
#pragma unroll
          for (int ww = 0; ww < h; ++ww) tq[ww] += tq[ww + h];
        xAx = tq[0];
      }
      const double ck1 = colbuf[par * LD + k1];
      // ---- w = beta·A v − K v with A v = A x − alpha·c, K = beta²·vᵀA v / 2; the next column x' = c − v w_{k+1} − w v_{k+1}
      const double vAv = fma(alpha * alpha, ck1, fma(-2.0 * alpha, Axk1, xAx));
      const double K = 0.5 * beta * beta * vAv;
      const double wk1 = fma(-K, vk1, beta * fma(-alpha, ck1, Axk1));
      TRI_CYC(2);
      const bool keeper = LEAD || w == (k & (NW - 1));  // the wave that files this step's reflector
      double* hv = a.Hv + (size_t)(k - off) * LD;
      double loc = 0.0;
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        const int i = l + 64 * s;
        if (s < S0) {
          if (keeper) hv[i] = 0.0;
          continue;
        }
        double sm = sm_[s], c = c_[s], xi = xi_[s];
        if constexpr (!kHoist) {
          if constexpr (TWO) {
            sm = psum[i];
          } else {
            const double* pp = part + par * NW * LD + i;
            double tq[NW];
#pragma unroll
            for (int ww = 0; ww < NW; ++ww) tq[ww] = pp[ww * LD];
#pragma unroll
            for (int h = NW / 2; h > 0; h >>= 1)
#pragma unroll
              for (int ww = 0; ww < h; ++ww) tq[ww] += tq[ww + h];
            sm = tq[0];
          }
          c = colbuf[par * LD + i];
          if constexpr (VLDS) xi = vb[par * LD + i]; else xi = xs[s];
        }
        const double vi = i == k1 ? vk1 : xi;
        const double wi = i > k ? fma(-K, vi, beta * fma(-alpha, c, sm)) : 0.0;
        const double xw = i > k1 ? fma(-vi, wk1, fma(-wi, vk1, c)) : 0.0;
        loc = i > k1 + 1 ? fma(xw, xw, loc) : loc;
        if constexpr (VLDS) {
          wb[i] = wi;
          vb[(par ^ 1) * LD + i] = xw;
          if (i == k1) vb[par * LD + i] = vk1;
        }
        vs[s] = vi; ws[s] = wi; xnew[s] = xw;
        if (keeper) hv[i] = vi;
      }
      if (keeper && l == 0) {
        a.beta[k - off] = beta;
        a.e[k - off] = alpha;
        a.d[k1 - off] = fma(-vk1, wk1, fma(-wk1, vk1, ck1));
      }
      if constexpr (VLDS && !LEAD) wave_lds_sync();
      TRI_CYC(3);
      // The next column's first entry and norm — what the NEXT step's reflector starts from.  Where one wave carries the scalar chain
      // (LEAD) they are taken BEHIND the barrier that releases the other waves into the pass (round 5): the lead reaches the next
      // step's first barrier ≈ 1,200 cycles ahead of the others anyway (tri_bench -DICP_TRI_CYCLES: "barrier 1,194"), the wave
      // reduction (≈ 300 cycles) was on every step's critical path for nothing.
      if constexpr (!LEAD) {
        double t = xnew[S0];
#pragma unroll
        for (int s = S0 + 1; s < SI; ++s) t = ((k1 + 1) >> 6) == s ? xnew[s] : t;
        x0 = readlane_f64(t, (k1 + 1) & 63);
        sig2 = wave_sum(loc);
      } else {
        lead_loc = loc;
      }
      }  // lead
      if constexpr (LEAD) {
        lds_barrier();  // v, w, x' of this step are in LDS
        if (lead) {
          double t = xnew[S0];
#pragma unroll
          for (int s = S0 + 1; s < SI; ++s) t = ((k1 + 1) >> 6) == s ? xnew[s] : t;
          x0 = readlane_f64(t, (k1 + 1) & 63);
          sig2 = wave_sum(lead_loc);
        }
      }
      TRI_CYC(4);
      pass(s0tag, Tag<1>{}, par, par ^ 1, k1 + 1, k, vs, ws, xnew);
#pragma unroll
      for (int s = 0; s < SI; ++s) xs[s] = xnew[s];
      TRI_CYC(7);
    }
  };
  // one phase per row slot: rows of the slots before it are finished
  auto run = [&](auto self, auto s0tag) -> void {
    constexpr int S0 = decltype(s0tag)::value;
    phase(s0tag);
    if constexpr (S0 + 1 < SI) self(self, Tag<S0 + 1>{});
  };
  run(run, Tag<0>{});
#ifdef ICP_TRI_CYCLES
  if (threadIdx.x == 0)
    for (int i = 0; i < 8; ++i) g_eigen_stamps[16 + i] = cyc[i];
#endif
  // what is left: e[n−2] = A[n−1][n−2] (the last x), d[n−1] (the last column published)
  lds_barrier();
  if (w == 0 && l == 0) {
    a.e[n - 2] = x0;
    a.d[n - 1] = colbuf[(((off + n - 3) & 1) ^ 1) * LD + LD - 1];
  }
}

template <int NW, int SI, int NT, int TOFF>
__device__ __forceinline__ void tridiag_kernel_body(const TridiagIO& a) {
  constexpr int LD = 64 * SI;
  __shared__ double lds[TridiagLds<NW, SI>::doubles];
  const int n = a.n, off = LD - n;
  const int l = threadIdx.x & 63;
  const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // N_ij = ½(M_ij + M_ji) / (sqrt_lambda_i · sqrt_lambda_j).  M is EXACTLY symmetric — every kernel that assembles it writes both
  // triangles from the lower triangle of the summed partials (k_assemble_many, k_assemble_posterior_matrix, the factor kernels) — so
  // ½(M_ij + M_ji) is M_ji bit for bit, and one read does: row j, the lanes along i (coalesced; the mirrored read M[i·n + j] put every
  // lane on a cache line of its own).  All reads are issued BEFORE the first value is used: left to itself the compiler waited for
  // each entry's loads in turn — 105 round trips to L2 one after the other, 0.1 ms of the launch alone and three times that beside
  // the evaluator's chip-wide searches (a wide step's second decomposition launch: 810 µs instead of 455, tools/r4_trace_c4.sh).
  double A[SI][NT], col0[SI], sli[SI], slj[NT];
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s - off;
    sli[s] = i >= 0 ? a.sqrt_lambda[i] : 1.0;
    col0[s] = i >= 0 ? a.M[i] : 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = w + NW * (t + TOFF) - off;
      A[s][t] = i >= 0 && j >= 0 ? a.M[(size_t)j * n + i] : 0.0;
    }
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int j = w + NW * (t + TOFF) - off;
    slj[t] = j >= 0 ? a.sqrt_lambda[j] : 1.0;
  }
  const double sl0 = a.sqrt_lambda[0], m00 = a.M[0];
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s - off;
    col0[s] = i >= 0 ? col0[s] / (sli[s] * sl0) : 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = w + NW * (t + TOFF) - off;
      A[s][t] = i >= 0 && j >= 0 ? A[s][t] / (sli[s] * slj[t]) : 0.0;
      if (a.Nout && i >= 0 && j >= 0) a.Nout[(size_t)j * n + i] = A[s][t];
    }
  }
  const double d0 = m00 / (sl0 * sl0);
  EIG_STAMP(0);
  tridiagonalise<NW, SI, NT, TOFF>(a, A, col0, d0, lds);
  EIG_STAMP(1);
}
template <int NW, int SI, int NT, int TOFF>
__global__ void __launch_bounds__(NW * 64) k_tridiag(TridiagIO a) { tridiag_kernel_body<NW, SI, NT, TOFF>(a); }
// the same for up to kTriMany matrices side by side, one workgroup each (the chains of a wide step: kernels_wide.hip)
constexpr int kTriMany = kTriManyMax;  // (32, icp_kernels.hpp: the largest of the records below, TriSolveMany, is 3.8 KB of the 4 KB argument segment)
struct TridiagMany { TridiagIO p[kTriMany]; };
template <int NW, int SI, int NT, int TOFF>
__global__ void __launch_bounds__(NW * 64) k_tridiag_many(TridiagMany m, const int* __restrict__ skip_all) {
  if (skip_all && skip_all[blockIdx.x] != 0) return;  // (the on-device loop: a chain that did not move)
  const TridiagIO a = m.p[blockIdx.x];  // (a copy: scalar registers, as a by-value kernel argument)
  tridiag_kernel_body<NW, SI, NT, TOFF>(a);
}

// Ranks <= 64, up to two decompositions per launch (one workgroup each), with the front end of a decomposition that was
// enqueued ahead of its input (EigenSpec, see k_posterior_eigen_rr): wait on the device for the regression launch's word, give up
// on cancellation or after 5 ms, sum the split-K partials.
struct TriSmallProblem {
  const double* M;            // r×r matrix — or, splits > 0, the partials: splits × (r+1)² row-major, lower triangle, identity not added
  int splits;
  const int* ready;           // (optional) device word raised to ready_seq or beyond when the partials are complete
  int ready_seq;
  const int* cancel;          // (optional) pinned host word: give up once *cancel == seq
  int seq;
  const double* sqrt_lambda;
  TridiagIO out;              // d, e, beta, Hv (M / sqrt_lambda unused here)
  int* sync;                  // the solve launch's words: [2] = skip
};
struct TriSmallBatch { int r; TriSmallProblem p[2]; };

__global__ void __launch_bounds__(256) k_tridiag_small(TriSmallBatch b) {
  constexpr int NW = 4, SI = 1, NT = 16;
  __shared__ double lds[TridiagLds<NW, SI>::doubles];
  __shared__ int s_cancel;
  const TriSmallProblem& pb = b.p[blockIdx.x];
  const int n = b.r, off = 64 - n;
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid == 0) s_cancel = 0;
  __syncthreads();
  const bool is_poll = pb.cancel != nullptr && tid == 255;
  if (tid == 255) {
    if (pb.ready) {
      const long long t0 = __builtin_amdgcn_s_memrealtime();  // 100 MHz
      for (;;) {
        if (__hip_atomic_load(pb.ready, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - pb.ready_seq >= 0) break;
        if (pb.cancel && __hip_atomic_load(pb.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == pb.seq) { s_cancel = 1; break; }
        if (__builtin_amdgcn_s_memrealtime() - t0 > 500000) { s_cancel = 2; break; }
        __builtin_amdgcn_s_sleep(32);
      }
    } else if (pb.cancel) {
      if (__hip_atomic_load(pb.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == pb.seq) s_cancel = 1;
    }
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (acquire side for the plain loads of the partials below)
  if (s_cancel) {
    if (tid == 0) __hip_atomic_store(pb.sync + 2, s_cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  // ---- the matrix into registers: lane l holds row l − off, wave w columns w + 4t − off
  const double* __restrict__ sl = pb.sqrt_lambda;
  double A[SI][NT], col0[SI];
  const int i = l - off;
  double d0;
  if (pb.splits > 0) {
    // Σ over the splits in split order from 0.0, like the factorisation; four splits × seventeen entries in flight
    const size_t nn = (size_t)(n + 1) * (n + 1);
    size_t offs[NT + 1];
    bool live[NT + 1];
#pragma unroll
    for (int t = 0; t <= NT; ++t) {
      const int j = t < NT ? w + NW * t - off : 0;  // (slot NT: column 0, for the first reflector)
      live[t] = i >= 0 && j >= 0;
      const int hi = max(i, j), lo = min(i, j);
      offs[t] = live[t] ? (size_t)hi * (n + 1) + lo : 0;
    }
    double acc[NT + 1];
#pragma unroll
    for (int t = 0; t <= NT; ++t) acc[t] = 0.0;
    for (int sp = 0; sp < pb.splits; sp += 4) {
      double q[4][NT + 1];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const double* src = pb.M + (size_t)min(sp + u, pb.splits - 1) * nn;
#pragma unroll
        for (int t = 0; t <= NT; ++t) q[u][t] = src[offs[t]];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
        if (sp + u < pb.splits) {
#pragma unroll
          for (int t = 0; t <= NT; ++t) acc[t] += q[u][t];
        }
    }
    const double sli = i >= 0 ? sl[i] : 1.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = w + NW * t - off;
      A[0][t] = live[t] ? (acc[t] + (i == j ? 1.0 : 0.0)) / (sli * sl[j]) : 0.0;
    }
    col0[0] = live[NT] ? (acc[NT] + (i == 0 ? 1.0 : 0.0)) / (sli * sl[0]) : 0.0;
    d0 = readlane_f64(col0[0], off);  // (N_00: lane off of every wave holds row 0)
  } else {
    auto entry = [&](int ii, int jj) { return 0.5 * (pb.M[(size_t)jj * n + ii] + pb.M[(size_t)ii * n + jj]) / (sl[ii] * sl[jj]); };
    col0[0] = i >= 0 ? entry(i, 0) : 0.0;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int j = w + NW * t - off;
      A[0][t] = i >= 0 && j >= 0 ? entry(i, j) : 0.0;
    }
    d0 = entry(0, 0);
  }
  TridiagIO a = pb.out;
  a.n = n;
  EIG_STAMP(0);
  tridiagonalise<NW, SI, NT, 0>(a, A, col0, d0, lds);
  EIG_STAMP(1);
  // cancelled meanwhile? (the solve launch then only tidies up)
  if (is_poll && __hip_atomic_load(pb.cancel, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) == pb.seq)
    __hip_atomic_store(pb.sync + 2, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------------------------------------------------
// One wave per eigenpair.
struct TriSolveIO {
  int n;
  const double* d;
  const double* e;
  const double* beta;
  const double* Hv;   // [n][64·SI], by position (index + 64·SI − n)
  double* V;          // n×n: column `rank` = eigenvector
  double* Vt;         // n×n: row `rank`
  double* S;          // [n] = 1/mu, descending
  double* mu;         // [n] scratch: the eigenvalues, ascending (the last wave checks the gaps)
  const double* wy;   // [ceil((n−2)/8)][64] T factors of the reflector blocks (k_tri_wy), row-major 8×8, upper triangles
  int* sync;          // [3] {waves finished, trouble flags, skip: 1 = cancelled, 2 = its input never came}: zero between launches
  int* status;        // status[0]: 0 ok, 2 = not trustworthy (gaps below resolution / non-finite); status[-1]: 0
  // (ranks <= 64, where nothing follows this launch) published by the last wave:
  int* host_status;   // pinned copy of the status (kEigenGaveUp for skip = 2)
  int* done_word;     // set to done_value when the outputs are complete — or the decomposition was dropped
  int done_value;
};

// Seven multisection passes and TWO twisted factorisations.  Round 3 had dropped the second factorisation (two chains of n − 1
// dependent divisions, 12 µs each at rank 200) and left what remained to the refinement step; round 5 made a factorisation cheap (the
// pivots as ratios of the three-term recurrence, the vector by a product scan: 10 µs with everything around it) and brought the second
// one back: with the eigenvalue corrected by the first round's Rayleigh quotient the second round's vector is accurate to eps·‖T‖/gap
// (rank 200, gaps down to 1e-4 of the norm: |ΔV| 5e-13, orthogonality 2e-13 WITHOUT the refinement step — after one round 4e-10 and
// 2e-9), and the refinement launches run only where a gap is narrow (kTriRefineGap).
constexpr int kTriPasses = 7;       // multisection passes of 64 points: the bracket shrinks 65× per pass (five left the vectors of a pair 1e-4 apart 1e-10 off, and would leave one 1e-6 apart 1e-6 off: tri_bench)
constexpr int kTriRounds = 2;       // twisted factorisations (each followed by a Rayleigh-quotient correction): round 5, two again — see kTriRefineGap
constexpr int kTriMaxN = 256;
constexpr int kWyBlock = 8;         // reflectors per compact-WY block of the back-transformation
// The refinement step behind the solve launch (Ogita & Aishima, below) exists for eigenvectors of CLOSE eigenvalues: out of the twisted
// factorisation a vector is accurate to eps·‖T‖/gap, and two vectors are orthogonal to each other only that far.  With every gap above
// this fraction of the norm that is 1e-10 — a thousandth of what the parity tests allow (V to 1e-7, the north star 1e-5) — and the
// four launches of the step (three r³ products, one elementwise pass: 40 µs at rank 200, 90 µs for 16 posteriors side by side) return at
// once, the last one handing X on as V.  Smaller gaps: the step runs as before.
constexpr double kTriRefineGap = 1e-6;

// number of eigenvalues of the (scaled) tridiagonal matrix below x: sign changes of the leading principal minors, by the
// three-term recurrence (one dependent fma per row), rescaled every eighth row
__device__ __forceinline__ int sturm_count(const double* __restrict__ ds, const double* __restrict__ e2, int n, double x) {
  double p0 = 1.0, p1 = ds[0] - x;
  int cnt = (unsigned)__double2hiint(p1) >> 31;
  // eight rows' coefficients fetched together: the recurrence itself is one dependent fma per row.  Whole groups of eight run
  // without a test per row (a taken branch costs more than the row); the remainder is one guarded group.
  // (An exact zero needs no care: by its sign bit it counts as positive, and its successor −e²·p0 has the sign opposite to its
  // predecessor's — one change across the three, whichever way round.)
  int i0 = 1;
  for (; i0 + 8 <= n; i0 += 8) {
    double dd[8], ee[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { dd[u] = ds[i0 + u] - x; ee[u] = e2[i0 + u - 1]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const double p2 = fma(dd[u], p1, -(ee[u] * p0));
      cnt += (unsigned)(__double2hiint(p2) ^ __double2hiint(p1)) >> 31;
      p0 = p1;
      p1 = p2;
    }
    const int ex = max(__builtin_amdgcn_frexp_exp(p0), __builtin_amdgcn_frexp_exp(p1));
    p0 = ldexp(p0, -ex);
    p1 = ldexp(p1, -ex);
  }
  if (i0 < n) {
    double dd[8], ee[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      const int i = min(i0 + u, n - 1);
      dd[u] = ds[i] - x;
      ee[u] = e2[i - 1];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (i0 + u < n) {
        const double p2 = fma(dd[u], p1, -(ee[u] * p0));
        cnt += (unsigned)(__double2hiint(p2) ^ __double2hiint(p1)) >> 31;
        p0 = p1;
        p1 = p2;
      }
    }
  }
  return cnt;
}

// The reflectors in blocks of eight, compact WY (dlarft, forward / columnwise): H_k0 ··· H_k0+7 = I − V·T·Vᵀ with T upper triangular,
// T_ii = β_i, T_{0:i,i} = −β_i · T_{0:i,0:i} · (V_{:,0:i}ᵀ v_i).  One wave per block (28 inner products, reduced seven at a time); the
// eigenpair waves of the solve launch then apply EIGHT reflectors per round of reductions (tri_solve_body).  Runs between the
// reduction and the solve.
struct TriWyIO {
  int n;
  const double* beta;
  const double* Hv;   // [n][64·SI]
  double* wy;         // [ceil((n−2)/8)][64]
  const int* sync;    // sync[2] != 0: the decomposition was dropped — nothing to do
};
template <int SI>
__device__ __forceinline__ void tri_wy_body(const TriWyIO& a, const int blk) {  // (one wave: threadIdx.x < 64)
  constexpr int LD = 64 * SI;
  const int n = a.n, l = threadIdx.x;
  const int k0 = blk * kWyBlock;
  if (k0 >= n - 2) return;
  if (__hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;
  double v[kWyBlock][SI], bet[kWyBlock];
#pragma unroll
  for (int q = 0; q < kWyBlock; ++q) {
    bet[q] = k0 + q < n - 2 ? a.beta[k0 + q] : 0.0;
#pragma unroll
    for (int s = 0; s < SI; ++s) v[q][s] = k0 + q < n - 2 ? a.Hv[(size_t)(k0 + q) * LD + l + 64 * s] : 0.0;
  }
  double T[kWyBlock][kWyBlock];
#pragma unroll
  for (int i = 0; i < kWyBlock; ++i) {
    double g[kWyBlock - 1];
#pragma unroll
    for (int jj = 0; jj < kWyBlock - 1; ++jj) {
      g[jj] = 0.0;
      if (jj < i) {
#pragma unroll
        for (int s = 0; s < SI; ++s) g[jj] = fma(v[jj][s], v[i][s], g[jj]);
      }
    }
    if (i > 0) wave_sum_many<kWyBlock - 1>(g);
#pragma unroll
    for (int jj = 0; jj < i; ++jj) {
      double t = 0.0;
#pragma unroll
      for (int m = jj; m < i; ++m) t = fma(T[jj][m], g[m], t);
      T[jj][i] = -bet[i] * t;
    }
    T[i][i] = bet[i];
  }
  double* out = a.wy + (size_t)blk * kWyBlock * kWyBlock;
  if (l == 0) {
#pragma unroll
    for (int jj = 0; jj < kWyBlock; ++jj)
#pragma unroll
      for (int i = 0; i < kWyBlock; ++i) out[jj * kWyBlock + i] = i >= jj ? T[jj][i] : 0.0;
  }
}
template <int SI>
__global__ void __launch_bounds__(64) k_tri_wy(TriWyIO a0, TriWyIO a1) { tri_wy_body<SI>(blockIdx.y ? a1 : a0, blockIdx.x); }
struct TriWyMany { TriWyIO p[kTriMany]; };
template <int SI>
__global__ void __launch_bounds__(64) k_tri_wy_many(TriWyMany m, const int* __restrict__ skip_all) {
  if (skip_all && skip_all[blockIdx.y] != 0) return;
  const TriWyIO a = m.p[blockIdx.y];
  tri_wy_body<SI>(a, blockIdx.x);
}

struct TriSolveLds {  // the solve launch's dynamic LDS (doubles): offsets for an n x n problem
  int oDs, oE2, oDsr, oE2r, oEs, oRed, oWy, oScr, scr_stride, oCarry, carry_stride, total;
  __host__ __device__ explicit TriSolveLds(int n) {
    const int np = n + 8, nblk = (n - 2 + kWyBlock - 1) / kWyBlock;
    oDs = 0; oE2 = np; oDsr = 2 * np; oE2r = 3 * np; oEs = 4 * np; oRed = oEs + n; oWy = oRed + 8;
    oScr = oWy + (nblk > 0 ? nblk : 0) * kWyBlock * kWyBlock;
    scr_stride = 2 * n + 2 * np;       // per wave: D⁺ | D⁻ by row, then the two chains' values (a group of eight filed whole)
    oCarry = oScr + 4 * scr_stride;
    carry_stride = n / 8 + 2;
    total = oCarry + 8 * carry_stride;
  }
};
inline size_t tri_solve_lds_bytes(int n) { return sizeof(double) * (size_t)TriSolveLds(n).total; }

// kBack: the back-transformation inside this launch, eigenvector by eigenvector (one row slot per lane: ranks <= 64, where the launch's
// last wave also publishes the completion words).  Otherwise the wave hands the tridiagonal matrix's eigenvector on as row j of Vt, and
// k_tri_back — sixteen vectors per wave on the matrix cores — takes it from there.
template <int SI, bool kBack>
__device__ __forceinline__ void tri_solve_body(const TriSolveIO& a) {
  constexpr int LD = 64 * SI;
  const int off = LD - a.n;  // position of index 0 (see tridiagonalise)
  // LDS sized by n, not by the largest rank (round 5: 64 KB of static arrays let two workgroups share a compute unit; at rank 200 the
  // arrays below take 49 KB and three do — the launch of 16 posteriors side by side, 800 workgroups of four f64-issue-bound waves,
  // is a throughput kernel): tri_solve_lds_bytes(n) of dynamic shared memory
  extern __shared__ double tri_dyn[];
  const int n = a.n;
  const TriSolveLds L(n);
  double* ds = tri_dyn + L.oDs;      // [n + 8] diagonal (scaled); the eight entries past the end are read, never used, by a chain's last group
  double* e2 = tri_dyn + L.oE2;      // [n + 8] squared off-diagonal
  double* dsr = tri_dyn + L.oDsr;    // [n + 8] … and bottom-up (the second chain of the factorisation reads forward, too)
  double* e2r = tri_dyn + L.oE2r;
  double* es = tri_dyn + L.oEs;      // [n] off-diagonal
  double* red = tri_dyn + L.oRed;    // [8]
  double* wy = tri_dyn + L.oWy;      // T factors of the reflector blocks
  const int tid = threadIdx.x, l = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int j = blockIdx.x * 4 + w;  // this wave's eigenvalue (ascending), rank j of the output
  TRI_STAMP(8);
  // a decomposition that was cancelled or never got its input (the reduction launch says so): nothing is computed, the waves only
  // count themselves in, the last one tidies up
  const int skip = __hip_atomic_load(a.sync + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (skip) {
    if (j >= a.n) return;
    int last = 0;
    if (l == 0) last = atomicAdd(a.sync, 1) == a.n - 1;
    if (last) {
      a.sync[0] = 0; a.sync[1] = 0; a.sync[2] = 0;
      a.sync[4] = 1;  // (no vectors: k_tri_back has nothing to do)
      if (skip == 2) {  // timed out: tell the host, and mark the basis that was never written so that nothing starts from it
        if (a.host_status) __hip_atomic_store(a.host_status, 3 /* kEigenGaveUp */, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        a.V[0] = __builtin_nan("");
      }
      if (a.done_word) __hip_atomic_store(a.done_word, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
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
    dsr[n - 1 - i] = a.d[i] * inv;
    if (i < n - 1) e2r[n - 2 - i] = fmax(ee * ee, 1e-280);
  }
  if (tid < 8) { ds[n + tid] = 0.0; e2[n + tid] = 0.0; dsr[n + tid] = 0.0; e2r[n - 1 + tid] = 0.0; }  // (read, never used, by the chains' last group)
  const int nblk = (n - 2 + kWyBlock - 1) / kWyBlock;
  if constexpr (kBack) {
    for (int i = tid; i < nblk * kWyBlock * kWyBlock; i += 256) wy[i] = a.wy[i];
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
  // ---- eigenvector of T by twisted factorisation, (T − λI) z = γ_r e_r with the twist r where |γ| is smallest; γ_r/‖z‖² corrects λ.
  // Round 5: the two pivot sequences D⁺ (from the top) and D⁻ (from the bottom) as RATIOS of the three-term recurrence's values,
  // p_m = (d_m − λ)·p_{m−1} − e²·p_{m−2}, D_m = p_m / p_{m−1} — the numbers of the pivot recurrence D_m = (d_m − λ) − e²/D_{m−1}, each step
  // perturbing d_m − λ and e² by an ulp as that one does — so that the chain of n − 1 dependent steps is one multiply-add per row
  // instead of a division (it was 144 cycles per row, 12 µs at rank 200), lane 0 on the matrix and lane 1 on its bottom-up copy with
  // one instruction stream; all divisions follow at once, a few per lane.  The vector's entries are products of the multipliers
  // away from the twist: an exclusive product scan over the lanes (four consecutive rows per lane, mantissa and exponent apart —
  // a partial product far from the twist may leave the range of a double where the entry itself is harmless) instead of two chains
  // of up to n − 1 dependent multiplications.  36 → 5 µs per eigenpair wave at rank 200.
  double* Dq = tri_dyn + L.oScr + w * L.scr_stride;  // [2][n]: D⁺ | D⁻ by row
  double* Pc = Dq + 2 * n;       // [2][n + 8]: the chains' values in chain order (top-down | bottom-up; a group of eight is filed whole)
  const int np = n + 8;
  double* zb = Dq;                   // z overwrites D⁺ once γ is known
  double lam = 0.5 * (lo + hi);
  double znorm2 = 1.0;
  bool trouble = false;
  for (int round = 0; round < kTriRounds; ++round) {
    if (l < 2) {
      const int dir = l;
      const double* dd_ = dir ? dsr : ds;
      const double* ee_ = dir ? e2r : e2;
      double* pc = Pc + dir * np;
      double* cg = tri_dyn + L.oCarry + (w * 2 + dir) * L.carry_stride;  // per wave and chain: the rescaled value a group of eight rows hands on
      double p0 = 1.0, p1 = dd_[0] - lam;
      pc[0] = p1;
      int g = 0;
      for (int m0 = 1; m0 < n; m0 += 8, ++g) {  // rows m0 .. m0+7 of the chain (past the end: padded coefficients, results unused)
        double dd[8], ee[8];  // (fetched where they are used: requested a group ahead the loop was SLOWER — 6.5 → 8.6 µs at rank 200; a lone
        // wave's f64 instructions issue every eight cycles, the loop is bound by their number, not by the LDS round trip)
#pragma unroll
        for (int u = 0; u < 8; ++u) { dd[u] = dd_[m0 + u] - lam; ee[u] = ee_[m0 + u - 1]; }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const double p2 = fma(dd[u], p1, -(ee[u] * p0));
          pc[m0 + u] = p2;
          p0 = p1;
          p1 = p2;
        }
        // rescaled by a power of two (both values: the recurrence is linear); the next group's first ratio divides by the rescaled value
        const int ex = max(__builtin_amdgcn_frexp_exp(p0), __builtin_amdgcn_frexp_exp(p1));
        p0 = ldexp(p0, -ex);
        p1 = ldexp(p1, -ex);
        cg[g] = p1;
      }
    }
    wave_lds_sync();
    TRI_STAMP(30);
    // D by row: chain position m of the top-down chain is row m, of the bottom-up chain row n − 1 − m; the denominator of a group's first
    // row is the value the group before handed over
#pragma unroll
    for (int s = 0; s < SI; ++s) {
      const int m = l + 64 * s;
      if (m < n) {
#pragma unroll
        for (int dir = 0; dir < 2; ++dir) {
          const double* pc = Pc + dir * np;
          const double den = m == 0 ? 1.0 : (((m - 1) & 7) == 0 && m > 1) ? (tri_dyn + L.oCarry + (w * 2 + dir) * L.carry_stride)[(m - 1) / 8 - 1] : pc[m - 1];
          const double D = pc[m] / den;
          Dq[dir * n + (dir ? n - 1 - m : m)] = D;
        }
      }
    }
    wave_lds_sync();
    TRI_STAMP(31);
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
    TRI_STAMP(32);
    // z_rt = 1; above the twist z_i = −(e_i / D⁺_i)·z_{i+1}, below it z_i = −(e_{i−1} / D⁻_i)·z_{i−1}: lane l takes rows SI·l .. SI·l + SI − 1
    {
      double fa[SI], fb[SI];  // the row's factor towards the twist from below (prefix side) / from above (suffix side); 1 elsewhere
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        const int i = SI * l + s;
        fa[s] = (i > rt && i < n) ? -(es[i - 1] / Dq[n + i]) : 1.0;
        fb[s] = (i < rt) ? -(es[i] / Dq[i]) : 1.0;
      }
      double pa[SI], pb[SI];  // inclusive products inside the lane: from its first row down / from its last row up
      pa[0] = fa[0];
#pragma unroll
      for (int s = 1; s < SI; ++s) pa[s] = pa[s - 1] * fa[s];
      pb[SI - 1] = fb[SI - 1];
#pragma unroll
      for (int s = SI - 2; s >= 0; --s) pb[s] = pb[s + 1] * fb[s];
      ScaledF64 xa = scaled_of(pa[SI - 1]), xb = scaled_of(pb[0]);
#pragma unroll
      for (int d = 1; d < 64; d <<= 1) {  // inclusive scans over the lanes (up for the prefix side, down for the suffix side)
        const ScaledF64 oa = scaled_shfl_up(xa, d), ob = scaled_shfl_down(xb, d);
        if (l >= d) xa = scaled_mul(xa, oa);
        if (l + d < 64) xb = scaled_mul(xb, ob);
      }
      ScaledF64 ea = scaled_shfl_up(xa, 1), eb = scaled_shfl_down(xb, 1);  // exclusive: what lies before / behind this lane
      if (l == 0) ea = ScaledF64{1.0, 0};
      if (l == 63) eb = ScaledF64{1.0, 0};
      wave_lds_sync();  // (every lane has read the pivots: z overwrites D⁺)
#pragma unroll
      for (int s = 0; s < SI; ++s) {
        const int i = SI * l + s;
        if (i < n) zb[i] = ldexp((ea.m * pa[s]) * (eb.m * pb[s]), ea.e + eb.e);
      }
    }
    wave_lds_sync();
    TRI_STAMP(33);
    double zz = 0.0;
    for (int i = l; i < n; i += 64) zz = fma(zb[i], zb[i], zz);
    znorm2 = wave_sum(zz);
    const double lam_new = lam + gamma / znorm2;
    if (lam_new >= lo && lam_new <= hi) lam = lam_new;
    if (!(znorm2 > 0.0) || !(znorm2 < 1e300)) trouble = true;
    if (round + 1 < kTriRounds) wave_lds_sync();
  }
  TRI_STAMP(11);
  const double muj = lam * anorm;
  if constexpr (!kBack) {
    // the vector of T, normalised, as row j of Vt (contiguous); back-transformation, sign and the transposed copy: k_tri_back
    const double sc = 1.0 / sqrt(znorm2);
    for (int i = l; i < n; i += 64) a.Vt[(size_t)j * n + i] = zb[i] * sc;
    if (!(muj > 0.0)) trouble = true;
    if (l == 0) { a.S[j] = 1.0 / muj; a.mu[j] = muj; }
  } else {
  // ---- back-transformation: z ← H_0 H_1 ··· H_{n−3} z, one row slot per lane, reflectors fetched eight ahead
  double zs[SI];
  {
    const double sc = 1.0 / sqrt(znorm2);
#pragma unroll
    for (int s = 0; s < SI; ++s) zs[s] = l + 64 * s >= off ? zb[l + 64 * s - off] * sc : 0.0;
  }
  {
    // in blocks of eight reflectors, z ← (I − V·T·Vᵀ) z with the T factors of k_tri_wy, last block first: eight inner products reduced
    // together (wave_sum_many) instead of eight dependent (dot product → wave reduction → update) rounds; the next block's
    // reflectors are requested a block ahead (L2: ≈ 500 cycles)
    // (two register sets, one in use while the other is on its way: every load is unconditional — a clamped row, zeroed afterwards
    // where the block runs past the last reflector — and issued before the block in hand is touched)
    double va[kWyBlock][SI], vb2[kWyBlock][SI];
    auto fetch = [&](double (&dst)[kWyBlock][SI], int blk) {
      const int b = max(blk, 0);
#pragma unroll
      for (int q = 0; q < kWyBlock; ++q) {
        const int k = min(b * kWyBlock + q, n - 3);
#pragma unroll
        for (int s = 0; s < SI; ++s) dst[q][s] = a.Hv[(size_t)k * LD + l + 64 * s];
      }
    };
    auto apply = [&](double (&v)[kWyBlock][SI], int blk) {
      const double* Tb = wy + blk * kWyBlock * kWyBlock;
      double u[kWyBlock];
#pragma unroll
      for (int q = 0; q < kWyBlock; ++q) {
        const bool live = blk * kWyBlock + q < n - 2;
        u[q] = 0.0;
#pragma unroll
        for (int s = 0; s < SI; ++s) {
          v[q][s] = live ? v[q][s] : 0.0;
          u[q] = fma(v[q][s], zs[s], u[q]);
        }
      }
      wave_sum_many<kWyBlock>(u);  // u = Vᵀz
#pragma unroll
      for (int jj = 0; jj < kWyBlock; ++jj) {  // t = T·u (upper triangle), then z −= V·t
        double t = 0.0;
#pragma unroll
        for (int i = jj; i < kWyBlock; ++i) t = fma(Tb[jj * kWyBlock + i], u[i], t);
#pragma unroll
        for (int s = 0; s < SI; ++s) zs[s] = fma(-t, v[jj][s], zs[s]);
      }
    };
    fetch(va, nblk - 1);
    for (int blk = nblk - 1; blk >= 0; blk -= 2) {
      fetch(vb2, blk - 1);
      __builtin_amdgcn_sched_barrier(0);
      apply(va, blk);
      if (blk - 1 >= 0) {
        fetch(va, blk - 2);
        __builtin_amdgcn_sched_barrier(0);
        apply(vb2, blk - 1);
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
    if (i >= off && av > bv) { bv = av; bi = i; }
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
  if (!(bv >= 0.0) || !(muj > 0.0)) trouble = true;
#pragma unroll
  for (int s = 0; s < SI; ++s) {
    const int i = l + 64 * s - off;
    if (i >= 0) {
      const double v = zs[s] * sgn;
      a.V[(size_t)i * n + j] = v;
      a.Vt[(size_t)j * n + i] = v;
    }
  }
  if (l == 0) { a.S[j] = 1.0 / muj; a.mu[j] = muj; }
  }  // kBack
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
  int bad = 0, close = 0;
  for (int i = l; i < n - 1; i += 64) {
    const double m0 = __hip_atomic_load(a.mu + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const double m1 = __hip_atomic_load(a.mu + i + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (!(m1 - m0 > 1e-10 * anorm)) bad = 1;
    if (!(m1 - m0 > kTriRefineGap * anorm)) close = 1;
  }
  bad = __any(bad);
  close = __any(close);
  if (l == 0) {
    if (__hip_atomic_load(a.sync + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) bad = 1;
    const int st = bad ? 2 : 0;
    a.status[0] = st;
    a.status[-1] = 0;
    a.sync[0] = 0;
    a.sync[1] = 0;
    // the refinement launches behind this one look at sync[3]: 0 = refine, 1 = every gap is wide enough for the vectors as they are
    // (hand X on as V), written anew by every solve launch
    a.sync[3] = (close || bad) ? 0 : 1;
    a.sync[4] = 0;
    if (a.host_status) __hip_atomic_store(a.host_status, st, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (a.done_word) __hip_atomic_store(a.done_word, a.done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
// Ranks above 64 (the back-transformation is k_tri_back's): the reflector blocks' T factors are nobody's input inside this launch, so the
// workgroups behind the eigenpairs' — one wave each — are k_tri_wy's: one launch and its boundary less per decomposition.
template <int SI>
__device__ __forceinline__ void tri_solve_or_wy(const TriSolveIO& a) {
  const int nwg = (a.n + 3) / 4;
  if constexpr (SI > 1) {
    if ((int)blockIdx.x >= nwg) {
      if (threadIdx.x < 64) tri_wy_body<SI>(TriWyIO{a.n, a.beta, a.Hv, const_cast<double*>(a.wy), a.sync}, (int)blockIdx.x - nwg);
      return;
    }
  }
  tri_solve_body<SI, SI == 1>(a);
}
template <int SI>
__global__ void __launch_bounds__(256) k_tri_solve(TriSolveIO a0, TriSolveIO a1) {
  tri_solve_or_wy<SI>(blockIdx.y ? a1 : a0);  // (up to two decompositions side by side: the two ICP directions of a chain step)
}
struct TriSolveMany { TriSolveIO p[kTriMany]; };
static_assert(sizeof(TriSolveMany) + 16 <= 4096, "the records of a launch must fit the kernel argument segment");
template <int SI>
__global__ void __launch_bounds__(256) k_tri_solve_many(TriSolveMany m, const int* __restrict__ skip_all) {
  if (skip_all && skip_all[blockIdx.y] != 0) return;
#ifdef ICP_TRI_SETPRIO
  __builtin_amdgcn_s_setprio(ICP_TRI_SETPRIO);
#endif
  const TriSolveIO a = m.p[blockIdx.y];
  tri_solve_or_wy<SI>(a);
}

// ---------------------------------------------------------------------------------------------------------------------
// Back-transformation on the matrix cores (round 5; ranks above 64): X = H_0 H_1 ··· H_{n−3} Z for SIXTEEN eigenvectors per wave.
// The vectors sit in the wave's registers as 16 x 16 tiles in the accumulator layout of v_mfma_f64_16x16x4 (rows = positions of the
// index space of tridiagonalise, columns = eigenvectors; tile t, register g of lane l: row 16t + (l >> 4) + 4g, column l & 15) —
// which is also the layout of the B operand of a product's k-step (B[k = l >> 4][j = l & 15]: register g IS k-step g).  Per block of
// eight reflectors, last block first:  U = V_bᵀ Z (four instructions per tile, A = the reflectors' entries straight from Hv),
// W = T_b U (two), Z −= V_b W (two per tile).  An eigenvector at a time (tri_solve_body<·, true>) a block is eight inner products, a
// wave reduction and eight updates per vector — 2,700 cycles; here ≈ 80 matrix instructions for sixteen vectors.  With 30 decompositions
// side by side the solve launch is the chip's throughput kernel (6,000 eigenpair waves, 510 µs); a third of that was this.
// Then the canonical sign (largest-|.| component positive, the first among equals), X and its transpose.
typedef double tri_d4 __attribute__((ext_vector_type(4)));
struct TriBackIO {
  int n;
  const double* Hv;   // [n][64·SI]
  const double* wy;   // [ceil((n−2)/8)][64] T factors (k_tri_wy)
  double* X;          // [n][n] out: eigenvectors in columns
  double* Xt;         // [n][n] in: the tridiagonal matrix's eigenvectors in rows (the solve launch); out: the transpose of X
  int* status;        // status[0] = 2 if a vector is not finite
  int* sync;          // sync[4] != 0: the solve launch computed nothing
};
// A workgroup = four waves = SIXTEEN eigenvectors: wave w owns the row tiles t ≡ w (mod 4) of the sixteen — as the blocks proceed
// (last block first) the reflectors reach further up, and the live tiles stay spread over the four waves — so that one decomposition
// by itself is 4·⌈n/16⌉ waves at work, not ⌈n/16⌉ (v_mfma_f64_16x16x4 takes 64 cycles on gfx950: the f64 matrix peak is the vector
// peak; one wave per sixteen vectors measured 90 µs at rank 200 where the eigenvector-at-a-time form takes 28).  Per block: every wave
// multiplies its tiles into a partial U, the four partials meet in LDS (summed in wave order by everybody), W = T·U by every wave,
// each updates its own tiles.  A block's eight reflectors (and its T factor) are staged in LDS by all 256 threads — row stride LD + 4:
// the two access patterns of the A operands then meet no more than two ways in a bank —, requested two blocks ahead.
template <int SI> struct TriBackLds {
  static constexpr int LD = 64 * SI, P = LD + 4;
  static constexpr int oT = 2 * kWyBlock * P;      // [2][64] T factors
  static constexpr int oU = oT + 2 * 64;           // [2][4 waves][4][64] partial products
  static constexpr int doubles = oU + 2 * 4 * 256;
};
template <int SI>
__device__ __forceinline__ void tri_back_body(const TriBackIO& a) {
  constexpr int LD = 64 * SI, P = TriBackLds<SI>::P;
  constexpr int PER = kWyBlock * LD / 256;  // staged entries per thread and block
  __shared__ double s_v[TriBackLds<SI>::doubles];
  double* s_t = s_v + TriBackLds<SI>::oT;
  double* s_u = s_v + TriBackLds<SI>::oU;
  const int n = a.n, off = LD - n;
  const int tid = threadIdx.x, l = tid & 63, jl = l & 15, kq = l >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int jc = blockIdx.x * 16 + jl;
  if (__hip_atomic_load(a.sync + 4, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return;  // (uniform per workgroup)
  // this wave's tiles: t = 4·i + w, i < SI; rows 16t + (l >> 4) + 4g of the position space, the matrix from position `off` on
  tri_d4 Z[SI];
#pragma unroll
  for (int i = 0; i < SI; ++i) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * (4 * i + w) + kq + 4 * g - off;
      Z[i][g] = (row >= 0 && jc < n) ? a.Xt[(size_t)jc * n + row] : 0.0;
    }
  }
  const int nblk = (n - 2 + kWyBlock - 1) / kWyBlock;
  // staging: thread tid carries entries tid, tid + 256, … of a block's 8 x LD reflector entries (reflectors past the last one: a clamped
  // row — their rows and columns of T are zero) and, the first 64 threads, one entry of its T factor.  The reflectors were written by
  // ONE compute unit (the reduction): most workgroups fetch them through another XCD's L2, so a block's entries are requested TWO
  // blocks ahead (three register sets in turn) and put into LDS one block ahead.
  struct Stage { double v[PER]; double t; };
  auto fetch = [&](Stage& st, int blk) {
    const int b = max(blk, 0), k0 = b * kWyBlock;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u, q = e / LD, pos = e - q * LD;
      st.v[u] = a.Hv[(size_t)min(k0 + q, n - 3) * LD + pos];
    }
    st.t = a.wy[(size_t)b * kWyBlock * kWyBlock + (tid & 63)];
  };
  auto put = [&](const Stage& st, int blk) {
    if (blk < 0) return;  // (uniform)
    const int buf = blk & 1;
#pragma unroll
    for (int u = 0; u < PER; ++u) {
      const int e = tid + 256 * u, q = e / LD, pos = e - q * LD;
      s_v[(buf * kWyBlock + q) * P + pos] = st.v[u];
    }
    if (tid < 64) s_t[buf * 64 + tid] = st.t;
  };
  // one block: U's part, exchange, W, update.  `nxt`: the register set to put into LDS for the block after this one.
  auto block = [&](int blk, const Stage& nxt) {
    if (blk < 0) return;  // (uniform)
    const int par = blk & 1;
    const double* vb = s_v + par * kWyBlock * P;
    const double* tb = s_t + par * 64;
    const int tmin = (blk * kWyBlock + off + 1) >> 4;  // reflector k is zero at the positions up to k + off
    // A operands: U's — lane (q = l & 15, k = l >> 4): entry 16t + 4s + k of reflector k0 + q (q < 8); the update's — lane (i = l & 15,
    // k = l >> 4): entry 16t + i of reflector k0 + 4s + k; T's — lane (jj = l & 15, k): T[jj][4s + k]
    const double qmask = jl < 8 ? 1.0 : 0.0;
    const double* hq = vb + (jl & 7) * P + kq;
    const double* hu0 = vb + kq * P + jl;
    const double* hu1 = vb + (4 + kq) * P + jl;
    double av[SI][4], uv[SI][2];
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      const int t = 4 * i + w;
#pragma unroll
      for (int sgm = 0; sgm < 4; ++sgm) av[i][sgm] = hq[16 * t + 4 * sgm] * qmask;
      uv[i][0] = -hu0[16 * t]; uv[i][1] = -hu1[16 * t];
    }
    const double ta0 = tb[(jl & 7) * kWyBlock + kq] * qmask, ta1 = tb[(jl & 7) * kWyBlock + 4 + kq] * qmask;
    tri_d4 U0 = {0.0, 0.0, 0.0, 0.0}, U1 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      if (4 * i + w < tmin) continue;  // (uniform per wave)
      U0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i][0], Z[i][0], U0, 0, 0, 0);
      U1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i][1], Z[i][1], U1, 0, 0, 0);
      U0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i][2], Z[i][2], U0, 0, 0, 0);
      U1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i][3], Z[i][3], U1, 0, 0, 0);
    }
    double* mine = s_u + (par * 4 + w) * 256;
#pragma unroll
    for (int g = 0; g < 4; ++g) mine[g * 64 + l] = U0[g] + U1[g];
    __syncthreads();
    tri_d4 U;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const double* pu = s_u + par * 4 * 256 + g * 64 + l;
      U[g] = ((pu[0] + pu[256]) + pu[512]) + pu[768];  // (wave order: the same sum in every wave)
    }
    tri_d4 W = {0.0, 0.0, 0.0, 0.0};
    W = __builtin_amdgcn_mfma_f64_16x16x4f64(ta0, U[0], W, 0, 0, 0);  // rows 0..3 of U
    W = __builtin_amdgcn_mfma_f64_16x16x4f64(ta1, U[1], W, 0, 0, 0);  // rows 4..7
#pragma unroll
    for (int i = 0; i < SI; ++i) {
      if (4 * i + w < tmin) continue;
      Z[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv[i][0], W[0], Z[i], 0, 0, 0);
      Z[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(uv[i][1], W[1], Z[i], 0, 0, 0);
    }
    put(nxt, blk - 1);
    __syncthreads();
  };
  Stage sa, sb, sc;
  fetch(sa, nblk - 1);
  fetch(sb, nblk - 2);
  fetch(sc, nblk - 3);
  put(sa, nblk - 1);
  __syncthreads();
  for (int blk = nblk - 1; blk >= 0; blk -= 3) {
    fetch(sa, blk - 3); block(blk, sb);
    fetch(sb, blk - 4); block(blk - 1, sc);
    fetch(sc, blk - 5); block(blk - 2, sa);
  }
  // ---- canonical sign per column: the largest-|.| component positive, the first among equals — over the four waves' rows
  double bv = -1.0, lv = 0.0;
  int bi = 0x7fffffff;
  bool nan = false;
#pragma unroll
  for (int i = 0; i < SI; ++i)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int row = 16 * (4 * i + w) + kq + 4 * g - off;
      const double av = fabs(Z[i][g]);
      if (row >= 0 && !(av >= 0.0)) nan = true;
      if (row >= 0 && (av > bv || (av == bv && row < bi))) { bv = av; bi = row; lv = Z[i][g]; }
    }
#pragma unroll
  for (int o = 16; o < 64; o <<= 1) {
    const double ov = __shfl_xor(bv, o, 64), olv = __shfl_xor(lv, o, 64);
    const int oi = __shfl_xor(bi, o, 64);
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; lv = olv; }
  }
  // (the exchange area is free: [wave][16 columns] of (|v|, index, v))
  double* sx = s_u;
  if (kq == 0) { sx[(w * 16 + jl) * 3] = bv; sx[(w * 16 + jl) * 3 + 1] = (double)bi; sx[(w * 16 + jl) * 3 + 2] = lv; }
  __syncthreads();
#pragma unroll
  for (int ww = 0; ww < 4; ++ww) {
    const double ov = sx[(ww * 16 + jl) * 3], olv = sx[(ww * 16 + jl) * 3 + 2];
    const int oi = (int)sx[(ww * 16 + jl) * 3 + 1];
    if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; lv = olv; }
  }
  const double sgn = lv < 0.0 ? -1.0 : 1.0;
  if (__any(nan && jc < n)) {
    if (l == 0) { a.status[0] = 2; a.sync[3] = 0; }
  }
  if (jc < n) {
#pragma unroll
    for (int i = 0; i < SI; ++i)
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const int row = 16 * (4 * i + w) + kq + 4 * g - off;
        if (row >= 0) {
          const double v = Z[i][g] * sgn;
          a.X[(size_t)row * n + jc] = v;
          a.Xt[(size_t)jc * n + row] = v;
        }
      }
  }
}
template <int SI>
__global__ void __launch_bounds__(256) k_tri_back(TriBackIO a0, TriBackIO a1) { tri_back_body<SI>(blockIdx.y ? a1 : a0); }
struct TriBackMany { TriBackIO p[kTriMany]; };
template <int SI>
__global__ void __launch_bounds__(256) k_tri_back_many(TriBackMany m, const int* __restrict__ skip_all) {
  if (skip_all && skip_all[blockIdx.y] != 0) return;
#ifdef ICP_TRI_SETPRIO
  __builtin_amdgcn_s_setprio(ICP_TRI_SETPRIO);
#endif
  const TriBackIO a = m.p[blockIdx.y];
  tri_back_body<SI>(a);
}

// ---------------------------------------------------------------------------------------------------------------------
// One refinement step (Ogita & Aishima 2018) on the eigenvector matrix X of the step before: eigenvectors of close eigenvalues
// come out of the twisted factorisations accurate to eps/gap each but not orthogonal to each other beyond that (1e-9 for the
// face posteriors); with R = I − XᵀX, S = XᵀNX, mu_i = S_ii/(1 − R_ii), E_ij = (S_ij + mu_j R_ij)/(mu_j − mu_i), E_ii = R_ii/2,
// X' = X + X·E converges quadratically — one step takes orthogonality and residual to working precision.  Four small products
// on the f64 matrix cores (one 16×16 tile per wave, operands straight from L2: lane l supplies P[k = l>>4][i = l&15] and
// Q[k][j = l&15], result register g is C[(l>>4) + 4g][l&15]) and one elementwise launch.
struct TriGemm {
  const double* P;   // [n][n] row-major: C = PᵀQ (+ mode)
  const double* Q;
  double* C;
  int mode;          // 0: C = PᵀQ; 1: C = I − PᵀQ; 2: C = X + PᵀQ, and Ct = Cᵀ
  const double* X;
  double* Ct;
  const int* skip;   // (optional) the solve launch's sync[3]: 1 = no refinement — modes 0, 1 return, mode 2 writes C = X, Ct = Xᵀ
};
__device__ __forceinline__ void tri_gemm_body(int n, const TriGemm& g) {
  const int l = threadIdx.x, l15 = l & 15, l4 = l >> 4;
  const int i0 = 16 * blockIdx.y, j0 = 16 * blockIdx.x;
  const bool vi = i0 + l15 < n, vj = j0 + l15 < n;
  if (g.skip && __hip_atomic_load(g.skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) {  // (uniform over the launch)
    if (g.mode == 2) {
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const int i = i0 + l4 + 4 * r4, j = j0 + l15;
        if (i < n && j < n) {
          const double v = g.X[(size_t)i * n + j];
          g.C[(size_t)i * n + j] = v;
          g.Ct[(size_t)j * n + i] = v;
        }
      }
    }
    return;
  }
  const double* p = g.P + i0 + l15;
  const double* q = g.Q + j0 + l15;
  tri_d4 acc = {0.0, 0.0, 0.0, 0.0};
  for (int k0 = 0; k0 < n; k0 += 32) {
    double a[8], b[8];
#pragma unroll
    for (int st = 0; st < 8; ++st) {
      const int k = k0 + 4 * st + l4;
      a[st] = vi && k < n ? p[(size_t)k * n] : 0.0;
      b[st] = vj && k < n ? q[(size_t)k * n] : 0.0;
    }
#pragma unroll
    for (int st = 0; st < 8; ++st) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[st], b[st], acc, 0, 0, 0);
  }
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    const int i = i0 + l4 + 4 * r4, j = j0 + l15;
    if (i < n && j < n) {
      double v = acc[r4];
      if (g.mode == 1) v = (i == j ? 1.0 : 0.0) - v;
      if (g.mode == 2) { v += g.X[(size_t)i * n + j]; g.Ct[(size_t)j * n + i] = v; }
      g.C[(size_t)i * n + j] = v;
    }
  }
}
__global__ void __launch_bounds__(64) k_tri_gemm(int n, TriGemm g0, TriGemm g1) {
  const TriGemm g = blockIdx.z ? g1 : g0;
  tri_gemm_body(n, g);
}
struct TriGemmMany { TriGemm g[2 * kTriMany]; };  // blockIdx.z = product
static_assert(sizeof(TriGemmMany) + 32 <= 4096, "the records of a launch must fit the kernel argument segment");
__global__ void __launch_bounds__(64) k_tri_gemm_many(int n, TriGemmMany m, const int* __restrict__ skip_all, int per_problem) {
  if (skip_all && skip_all[blockIdx.z / per_problem] != 0) return;
  const TriGemm g = m.g[blockIdx.z];
  tri_gemm_body(n, g);
}
__global__ void __launch_bounds__(256) k_tri_correction(int n, const double* __restrict__ S, const double* __restrict__ R, double* __restrict__ E,
                                                        double* __restrict__ Sout, const int* __restrict__ skip) {
  const int e = blockIdx.x * 256 + threadIdx.x;
  if (e >= n * n) return;
  if (skip && __hip_atomic_load(skip, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) return;  // (S stays the solve launch's)
  const int i = e / n, j = e - i * n;
  const double rii = R[(size_t)i * n + i], rjj = R[(size_t)j * n + j];
  const double mi = S[(size_t)i * n + i] / (1.0 - rii), mj = S[(size_t)j * n + j] / (1.0 - rjj);
  double v;
  if (i == j) {
    v = 0.5 * rii;
    Sout[i] = 1.0 / mi;
  } else {
    const double den = mj - mi;
    v = fabs(den) > 1e-11 * (fabs(mi) + fabs(mj)) ? fma(mj, R[e], S[e]) / den : 0.5 * R[e];
  }
  E[e] = v;
}
struct TriCorrMany { const double* S[kTriMany]; const double* R[kTriMany]; double* E[kTriMany]; double* Sout[kTriMany]; const int* skip[kTriMany]; };
__global__ void __launch_bounds__(256) k_tri_correction_many(int n, TriCorrMany m, const int* __restrict__ skip_all) {
  const int e = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  if (e >= n * n) return;
  if (skip_all && skip_all[q] != 0) return;
  if (m.skip[q] && __hip_atomic_load(m.skip[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 1) return;
  const double* __restrict__ S = m.S[q];
  const double* __restrict__ R = m.R[q];
  const int i = e / n, j = e - i * n;
  const double rii = R[(size_t)i * n + i], rjj = R[(size_t)j * n + j];
  const double mi = S[(size_t)i * n + i] / (1.0 - rii), mj = S[(size_t)j * n + j] / (1.0 - rjj);
  double v;
  if (i == j) {
    v = 0.5 * rii;
    m.Sout[q][i] = 1.0 / mi;
  } else {
    const double den = mj - mi;
    v = fabs(den) > 1e-11 * (fabs(mi) + fabs(mj)) ? fma(mj, R[e], S[e]) / den : 0.5 * R[e];
  }
  m.E[q][e] = v;
}
struct TriDoneMany { const int* status[kTriMany]; int* host_status[kTriMany]; int* done_word[kTriMany]; int done_value[kTriMany]; };
__global__ void k_tri_done_many(TriDoneMany m, const int* __restrict__ skip_all) {
  const int q = blockIdx.x;
  if (skip_all && skip_all[q] != 0) return;
  if (m.host_status[q]) __hip_atomic_store(m.host_status[q], m.status[q][0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (m.done_word[q]) __hip_atomic_store(m.done_word[q], m.done_value[q], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
// M = I + the summed split-K partial of a posterior, both triangles (k_assemble_posterior_matrix), for up to kTriMany posteriors
struct AssembleMany { const double* P[kTriMany]; double* M[kTriMany]; };
__global__ void __launch_bounds__(256) k_assemble_many(int r, AssembleMany m, const int* __restrict__ skip_all) {
  const int e = blockIdx.x * 256 + threadIdx.x, q = blockIdx.y;
  if (e >= r * r || !m.P[q]) return;
  if (skip_all && skip_all[q] != 0) return;
  const int i = e / r, j = e - i * r;
  const int hi = max(i, j), lo = min(i, j);
  m.M[q][e] = m.P[q][(size_t)hi * (r + 1) + lo] + (i == j ? 1.0 : 0.0);
}
__global__ void k_tri_done(const int* status, int* host_status, int* done_word, int done_value) {
  if (host_status) __hip_atomic_store(host_status, status[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
  if (done_word) __hip_atomic_store(done_word, done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tri
}  // namespace icp
