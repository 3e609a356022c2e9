// root_bench.hip — stand-alone timing of k_posterior_root (the opt-in Cholesky-root sampler), dev tool.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../icp-proposal_amd/csrc/icp_kernels.hpp"
namespace icp { extern __device__ long long g_eigen_stamps[64]; }
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
int main(int argc, char** argv) {
  const int r = argc > 1 ? atoi(argv[1]) : 51, S = 13, n = r + 1;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  std::vector<double> sl(r), M((size_t)r * r, 0.0);
  for (int j = 0; j < r; ++j) sl[j] = std::sqrt(28.0 * std::pow(0.182 / 28.0, (double)j / (r - 1)));
  for (int i = 0; i < r; ++i) M[(size_t)i * r + i] = 1.0;
  for (int k = 0; k < 6 * r; ++k) {
    std::vector<double> b(r);
    for (int j = 0; j < r; ++j) b[j] = nd(rng) * sl[j];
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) M[(size_t)i * r + j] += 0.02 * b[i] * b[j];
  }
  std::vector<double> Mp((size_t)S * n * n, 0.0);
  for (int sp = 0; sp < S; ++sp)
    for (int i = 0; i < r; ++i) for (int j = 0; j <= i; ++j) Mp[(size_t)sp * n * n + (size_t)i * n + j] = (M[(size_t)i * r + j] - (i == j ? 1.0 : 0.0)) / S;
  double *dM, *dMp, *dsl, *dV, *dVt, *dS, *dwork; int *dstat, *dready, *hcancel, *hstat, *dword;
  CK(hipMalloc(&dM, 8 * r * r)); CK(hipMalloc(&dMp, 8 * Mp.size())); CK(hipMalloc(&dsl, 8 * r)); CK(hipMalloc(&dV, 8 * r * r)); CK(hipMalloc(&dVt, 8 * r * r));
  CK(hipMalloc(&dS, 8 * r)); CK(hipMalloc(&dwork, 8 * icp::eigen_work_doubles(r))); CK(hipMalloc(&dstat, 64)); CK(hipMalloc(&dready, 64)); CK(hipMalloc(&dword, 64));
  CK(hipMemset(dready, 0x01, 64));
  CK(hipHostMalloc((void**)&hcancel, 64, hipHostMallocDefault)); CK(hipHostMalloc((void**)&hstat, 64, hipHostMallocDefault));
  hcancel[0] = 0;
  CK(hipMemcpy(dM, M.data(), 8 * r * r, hipMemcpyHostToDevice)); CK(hipMemcpy(dMp, Mp.data(), 8 * Mp.size(), hipMemcpyHostToDevice));
  CK(hipMemcpy(dsl, sl.data(), 8 * r, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  icp::EigenSpec spec{S, hcancel, 7, dready, 1};
  for (int mode = 0; mode < 2; ++mode) {
    icp::EigenRequest rq{mode ? dMp : dM, nullptr, dV, dVt, dS, dwork, dstat + 1, mode ? &spec : nullptr, hstat, dword, 1, dsl};
    rq.root = true;
    icp::launch_posterior_eigen_pair(st, r, dsl, 1, &rq); hipStreamSynchronize(st);
    float ms = 0; const int reps = 20;
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) icp::launch_posterior_eigen_pair(st, r, dsl, 1, &rq);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    long long s[64]; hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
    std::vector<double> V((size_t)r * r); hipMemcpy(V.data(), dV, 8 * r * r, hipMemcpyDeviceToHost);
    // check: V = L with L Lᵀ = M
    double err = 0;
    for (int a2 = 0; a2 < r; ++a2) for (int b2 = 0; b2 < r; ++b2) {
      double t = 0;
      for (int k = 0; k < r; ++k) t += V[(size_t)a2 * r + k] * V[(size_t)b2 * r + k];
      err = std::fmax(err, std::fabs(t - M[(size_t)a2 * r + b2]) / std::fabs(M[(size_t)a2 * r + a2]));
    }
    printf("root %s r=%d: %.1f us/call | wait %.1f factor %.1f store L %.1f tail %.1f | max rel |L L' - M| %.2e\n", mode ? "(partials)" : "(plain M) ", r,
           ms * 1000 / reps, (s[1] - s[0]) * 0.01, (s[3] - s[1]) * 0.01, (s[4] - s[3]) * 0.01, (s[5] - s[4]) * 0.01, err);
  }
  return 0;
}
