// tri_bench.hip — accuracy and timing of the posterior eigen-decomposition against a host Jacobi iteration (dev tool, not product).
// usage: tri_bench <rank> ; route chosen by ICP_EIGEN_TRIDIAG=0|1 (dev switch of kernels_posterior.hip)
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../icp-proposal_amd/csrc/icp_kernels.hpp"

namespace icp { extern __device__ long long g_eigen_stamps[64]; }
#define CK(x) do{hipError_t e_=(x); if(e_!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e_), __LINE__); return 1;}}while(0)

static void host_jacobi(int n, std::vector<double> a, std::vector<double>& w, std::vector<double>& V) {
  V.assign((size_t)n * n, 0.0);
  for (int i = 0; i < n; ++i) V[(size_t)i * n + i] = 1.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; ++i) { diag += a[(size_t)i * n + i] * a[(size_t)i * n + i]; for (int j = i + 1; j < n; ++j) off += a[(size_t)i * n + j] * a[(size_t)i * n + j]; }
    if (off <= 1e-60 * diag) break;
    for (int p = 0; p < n - 1; ++p)
      for (int q = p + 1; q < n; ++q) {
        const double apq = a[(size_t)p * n + q];
        if (apq == 0.0) continue;
        const double tau = (a[(size_t)q * n + q] - a[(size_t)p * n + p]) / (2.0 * apq);
        const double t = (tau >= 0 ? 1.0 : -1.0) / (std::fabs(tau) + std::sqrt(1.0 + tau * tau));
        const double c = 1.0 / std::sqrt(1.0 + t * t), s = t * c;
        for (int k = 0; k < n; ++k) { const double x = a[(size_t)k * n + p], y = a[(size_t)k * n + q]; a[(size_t)k * n + p] = c * x - s * y; a[(size_t)k * n + q] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const double x = a[(size_t)p * n + k], y = a[(size_t)q * n + k]; a[(size_t)p * n + k] = c * x - s * y; a[(size_t)q * n + k] = s * x + c * y; }
        for (int k = 0; k < n; ++k) { const double x = V[(size_t)k * n + p], y = V[(size_t)k * n + q]; V[(size_t)k * n + p] = c * x - s * y; V[(size_t)k * n + q] = s * x + c * y; }
      }
  }
  w.resize(n);
  for (int i = 0; i < n; ++i) w[i] = a[(size_t)i * n + i];
}

// A chip-wide launch to run BESIDE the decomposition (argv[2]): what of a busy chip slows the one-workgroup kernels down?
// 1 = multiply-adds in registers, 2 = loads that miss the L1 (a 64 MiB buffer, L2 / fabric), 3 = LDS traffic, 4 = stores,
// 5 = f64 matrix instructions; every workgroup runs until `until` (s_memrealtime, 100 MHz)
__global__ void __launch_bounds__(256) k_hog(int kind, unsigned long long ticks, double* buf, size_t n, double* sink) {
  __shared__ double lds[4096];
  const unsigned long long until = __builtin_amdgcn_s_memrealtime() + ticks;
  const int tid = threadIdx.x;
  double x = tid * 1e-3, y = 1.0;
  size_t pos = ((size_t)blockIdx.x * 256 + tid) * 8 % n;
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {0, 0, 0, 0};
  lds[tid] = x;
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() < until) {
    if (kind == 1) {
#pragma unroll
      for (int i = 0; i < 64; ++i) { x = fma(x, 1.0000001, y); y = fma(y, 0.9999999, x); }
    } else if (kind == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { x += buf[pos]; pos = (pos + 256 * 1031) % n; }
    } else if (kind == 3) {
#pragma unroll
      for (int i = 0; i < 32; ++i) { x += lds[(tid * 17 + i * 33) & 4095]; }
      lds[(tid + 1) & 4095] = x;
    } else if (kind == 4) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { buf[pos] = x; pos = (pos + 256 * 1031) % n; }
    } else if (kind == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, acc, 0, 0, 0);
    }
  }
  if (x + y + acc[0] == 12345.678) sink[0] = x;
}

int main(int argc, char** argv) {
  const int r = argc > 1 ? atoi(argv[1]) : 200;
  const int hog = argc > 2 ? atoi(argv[2]) : 0;
  const int hog_blocks = argc > 3 ? atoi(argv[3]) : 2048;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  const int K = 6 * r;
  std::vector<double> sl(r), B((size_t)K * r), M((size_t)r * r, 0.0);
  for (int j = 0; j < r; ++j) sl[j] = std::sqrt(28.0 * std::pow(0.182 / 28.0, (double)j / (r - 1)));
  for (auto& b : B) b = nd(rng);
  for (int k = 0; k < K; ++k) for (int j = 0; j < r; ++j) B[(size_t)k * r + j] *= sl[j];
  for (int i = 0; i < r; ++i) M[(size_t)i * r + i] = 1.0;
  for (int k = 0; k < K; ++k) for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) M[(size_t)i * r + j] += 0.02 * B[(size_t)k * r + i] * B[(size_t)k * r + j];
  std::vector<double> N((size_t)r * r), mu, Vh;
  for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) N[(size_t)i * r + j] = 0.5 * (M[(size_t)i * r + j] + M[(size_t)j * r + i]) / (sl[i] * sl[j]);
  host_jacobi(r, N, mu, Vh);
  // order by mu ascending (S descending), canonical signs
  std::vector<int> ord(r);
  for (int i = 0; i < r; ++i) ord[i] = i;
  std::sort(ord.begin(), ord.end(), [&](int a, int b) { return mu[a] < mu[b]; });
  std::vector<double> Vref((size_t)r * r), Sref(r);
  double mingap = 1e300;
  for (int c = 0; c < r; ++c) {
    const int p = ord[c];
    int bi = 0; double bv = -1;
    for (int k = 0; k < r; ++k) if (std::fabs(Vh[(size_t)k * r + p]) > bv) { bv = std::fabs(Vh[(size_t)k * r + p]); bi = k; }
    const double sg = Vh[(size_t)bi * r + p] < 0 ? -1.0 : 1.0;
    for (int k = 0; k < r; ++k) Vref[(size_t)k * r + c] = sg * Vh[(size_t)k * r + p];
    Sref[c] = 1.0 / mu[p];
    if (c) mingap = std::fmin(mingap, mu[p] - mu[ord[c - 1]]);
  }
  printf("rank %d: mu in [%.4g, %.4g], smallest gap %.3e (%.2e of the norm)\n", r, mu[ord[0]], mu[ord[r - 1]], mingap, mingap / mu[ord[r - 1]]);

  double *dM, *dsl, *dV, *dVt, *dS, *dwork; int* dstat;
  CK(hipMalloc(&dM, 8 * r * r)); CK(hipMalloc(&dsl, 8 * r)); CK(hipMalloc(&dV, 8 * r * r)); CK(hipMalloc(&dVt, 8 * r * r)); CK(hipMalloc(&dS, 8 * r));
  CK(hipMalloc(&dwork, 8 * icp::eigen_work_doubles(r))); CK(hipMemset(dwork, 0, 8 * icp::eigen_work_doubles(r))); CK(hipMalloc(&dstat, 64));
  CK(hipMemcpy(dM, M.data(), 8 * r * r, hipMemcpyHostToDevice)); CK(hipMemcpy(dsl, sl.data(), 8 * r, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  CK(hipMemset(dV, 0, 8 * r * r)); CK(hipMemset(dS, 0, 8 * r));
  icp::launch_posterior_eigen(st, r, dM, dsl, nullptr, dV, dVt, dS, dwork, dstat + 1);
  CK(hipStreamSynchronize(st));
  std::vector<double> V((size_t)r * r), Vt((size_t)r * r), S(r);
  CK(hipMemcpy(V.data(), dV, 8 * r * r, hipMemcpyDeviceToHost)); CK(hipMemcpy(Vt.data(), dVt, 8 * r * r, hipMemcpyDeviceToHost)); CK(hipMemcpy(S.data(), dS, 8 * r, hipMemcpyDeviceToHost));
  int stat[2]; CK(hipMemcpy(stat, dstat, 8, hipMemcpyDeviceToHost));
  double dv = 0, dsr = 0, res = 0, orth = 0, dvt = 0;
  for (size_t i = 0; i < V.size(); ++i) dv = std::fmax(dv, std::fabs(V[i] - Vref[i]));
  for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) dvt = std::fmax(dvt, std::fabs(V[(size_t)i * r + j] - Vt[(size_t)j * r + i]));
  for (int i = 0; i < r; ++i) dsr = std::fmax(dsr, std::fabs(S[i] - Sref[i]) / Sref[i]);
  for (int c = 0; c < r; ++c) {
    for (int i = 0; i < r; ++i) {
      double t = 0;
      for (int j = 0; j < r; ++j) t += N[(size_t)i * r + j] * V[(size_t)j * r + c];
      res = std::fmax(res, std::fabs(t - V[(size_t)i * r + c] / S[c]));
    }
    for (int c2 = 0; c2 < r; ++c2) {
      double t = 0;
      for (int i = 0; i < r; ++i) t += V[(size_t)i * r + c] * V[(size_t)i * r + c2];
      orth = std::fmax(orth, std::fabs(t - (c == c2)));
    }
  }
  printf("status %d (sweeps %d) | vs host Jacobi: max|dV| %.3e, max rel dS %.3e | residual %.3e, orthogonality %.3e, V vs Vt %.1e\n", stat[1], stat[0], dv, dsr, res,
         orth, dvt);
  float ms = 0; const int reps = 20;
  hipEventRecord(a, st);
  for (int i = 0; i < reps; ++i) icp::launch_posterior_eigen(st, r, dM, dsl, nullptr, dV, dVt, dS, dwork, dstat + 1);
  hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
  long long s[64]; hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
  printf("%.1f us per decomposition (cold) | stamps: reduction %.1f | solve: setup %.1f multisection %.1f vectors %.1f back-transformation %.1f output %.1f\n", ms * 1000 / reps,
         (s[1] - s[0]) * 0.01, (s[9] - s[8]) * 0.01, (s[10] - s[9]) * 0.01, (s[11] - s[10]) * 0.01, (s[12] - s[11]) * 0.01, (s[13] - s[12]) * 0.01);
  printf("vectors in detail: p-chains %.1f | ratios %.1f | twist search %.1f | z scan %.1f | norm %.1f\n", (s[30] - s[10]) * 0.01, (s[31] - s[30]) * 0.01,
         (s[32] - s[31]) * 0.01, (s[33] - s[32]) * 0.01, (s[11] - s[33]) * 0.01);
  if (hog) {
    double *hbuf, *sink; const size_t hn = (size_t)8 << 20;
    CK(hipMalloc(&hbuf, 8 * hn)); CK(hipMemset(hbuf, 0, 8 * hn)); CK(hipMalloc(&sink, 64));
    hipStream_t st2; CK(hipStreamCreateWithFlags(&st2, hipStreamNonBlocking));
    for (int rep = 0; rep < 3; ++rep) {
      CK(hipDeviceSynchronize());
      hipEventRecord(a, st);
      icp::launch_posterior_eigen(st, r, dM, dsl, nullptr, dV, dVt, dS, dwork, dstat + 1);
      hipEventRecord(b, st);
      hipLaunchKernelGGL(k_hog, dim3(hog_blocks), dim3(256), 0, st2, hog, 150000ull /* 1.5 ms */, hbuf, hn, sink);
      hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
      CK(hipDeviceSynchronize());
      hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
      printf("beside hog %d (%d workgroups): %.1f us | reduction %.1f | solve: setup %.1f multisection %.1f vectors %.1f back-transformation %.1f output %.1f\n", hog, hog_blocks,
             ms * 1000, (s[1] - s[0]) * 0.01, (s[9] - s[8]) * 0.01, (s[10] - s[9]) * 0.01, (s[11] - s[10]) * 0.01, (s[12] - s[11]) * 0.01, (s[13] - s[12]) * 0.01);
    }
  }
  if (r <= 64) {  // the speculative form of the same decomposition: split-K partials, ready word raised, cancel word in pinned memory
    const int n1 = r + 1, SP = 13;
    std::vector<double> Mp((size_t)SP * n1 * n1, 0.0);
    for (int sp = 0; sp < SP; ++sp)
      for (int i = 0; i < r; ++i) for (int j = 0; j <= i; ++j) Mp[(size_t)sp * n1 * n1 + (size_t)i * n1 + j] = (M[(size_t)i * r + j] - (i == j ? 1.0 : 0.0)) / SP;
    double* dMp; int *dready, *hcancel, *hstat, *ddone;
    CK(hipMalloc(&dMp, 8 * Mp.size())); CK(hipMemcpy(dMp, Mp.data(), 8 * Mp.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dready, 64)); CK(hipMemset(dready, 0x01, 64)); CK(hipMalloc(&ddone, 64)); CK(hipMemset(ddone, 0, 64));
    CK(hipHostMalloc((void**)&hcancel, 64, hipHostMallocDefault)); CK(hipHostMalloc((void**)&hstat, 64, hipHostMallocDefault));
    hcancel[0] = 0; hstat[0] = -1;
    icp::EigenSpec spec{SP, hcancel, 7, dready, 1};
    CK(hipMemset(dV, 0, 8 * r * r));
    icp::EigenRequest rq{dMp, nullptr, dV, dVt, dS, dwork, dstat + 1, &spec, hstat, ddone, 41, dsl};
    rq.direct = true;
    icp::launch_posterior_eigen_pair(st, r, dsl, 1, &rq);
    CK(hipStreamSynchronize(st));
    std::vector<double> V2((size_t)r * r);
    CK(hipMemcpy(V2.data(), dV, 8 * r * r, hipMemcpyDeviceToHost));
    int done = 0; CK(hipMemcpy(&done, ddone, 4, hipMemcpyDeviceToHost));
    double dv2 = 0;
    for (size_t i = 0; i < V2.size(); ++i) dv2 = std::fmax(dv2, std::fabs(V2[i] - Vref[i]));
    float ms2 = 0;
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) icp::launch_posterior_eigen_pair(st, r, dsl, 1, &rq);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms2, a, b);
    // cancelled before it starts: nothing written, the completion word still set
    hcancel[0] = 7; CK(hipMemset(ddone, 0, 64)); rq.done_value = 42; CK(hipMemset(dV, 0, 8 * r * r));
    icp::launch_posterior_eigen_pair(st, r, dsl, 1, &rq); CK(hipStreamSynchronize(st));
    int done2 = 0; CK(hipMemcpy(&done2, ddone, 4, hipMemcpyDeviceToHost));
    double v00 = 1; CK(hipMemcpy(&v00, dV, 8, hipMemcpyDeviceToHost));
    printf("speculative form (13 partials): max|dV| %.3e vs host Jacobi, pinned status %d, completion word %d; %.1f us per decomposition | cancelled: completion word %d, V untouched %d\n",
           dv2, hstat[0], done, ms2 * 1000 / reps, done2, v00 == 0.0);
  }
  if (r >= 3) {
    const double st_ = r - 2;
    printf("reduction, shader cycles per step: reflector %.0f | barrier %.0f | scalars %.0f | rows + private LDS %.0f | norm %.0f | pass %.0f | publish + xAx %.0f | (tail %.0f)\n", s[16] / st_,
           s[17] / st_, s[18] / st_, s[19] / st_, s[20] / st_, s[21] / st_, s[22] / st_, s[23] / st_);
  }
  return 0;
}
