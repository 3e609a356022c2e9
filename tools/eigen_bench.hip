// eigen_bench.hip — stand-alone timing of the posterior eigen-decomposition kernel (dev tool, not product).
// build: see tools/Makefile (compiles ../icp-proposal_amd/csrc/kernels_posterior.hip with -DICP_EIGEN_TIMING)
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
  const int r = argc > 1 ? atoi(argv[1]) : 51;
  std::mt19937_64 rng(7);
  std::normal_distribution<double> nd;
  // M = I + Bᵀ diag(w) B  (K = 6r rows), lambda decaying like the femur model
  auto make_M = [&](const std::vector<double>& B, int K) {
    std::vector<double> M((size_t)r * r, 0.0);
    for (int i = 0; i < r; ++i) M[(size_t)i * r + i] = 1.0;
    for (int k = 0; k < K; ++k)
      for (int i = 0; i < r; ++i)
        for (int j = 0; j < r; ++j) M[(size_t)i * r + j] += 0.02 * B[(size_t)k * r + i] * B[(size_t)k * r + j];
    return M;
  };
  const int K = 6 * r;
  std::vector<double> sl(r), B((size_t)K * r);
  for (int j = 0; j < r; ++j) sl[j] = std::sqrt(28.0 * std::pow(0.182 / 28.0, (double)j / (r - 1)));
  for (auto& b : B) b = nd(rng);
  for (int k = 0; k < K; ++k) for (int j = 0; j < r; ++j) B[(size_t)k * r + j] *= sl[j];
  std::vector<double> M0 = make_M(B, K);
  std::vector<double> B2 = B;
  for (auto& b : B2) b *= 1.0 + 0.03 * nd(rng);
  std::vector<double> M1 = make_M(B2, K);

  double *dM0, *dM1, *dsl, *dV0, *dV1, *dVt, *dS, *dwork; int* dstat;
  CK(hipMalloc(&dM0, 8 * r * r)); CK(hipMalloc(&dM1, 8 * r * r)); CK(hipMalloc(&dsl, 8 * r));
  CK(hipMalloc(&dV0, 8 * r * r)); CK(hipMalloc(&dV1, 8 * r * r)); CK(hipMalloc(&dVt, 8 * r * r)); CK(hipMalloc(&dS, 8 * r));
  CK(hipMalloc(&dwork, 8 * icp::eigen_work_doubles(r))); CK(hipMemset(dwork, 0, 8 * icp::eigen_work_doubles(r))); CK(hipMalloc(&dstat, 64));
  CK(hipMemcpy(dM0, M0.data(), 8 * r * r, hipMemcpyHostToDevice)); CK(hipMemcpy(dM1, M1.data(), 8 * r * r, hipMemcpyHostToDevice));
  CK(hipMemcpy(dsl, sl.data(), 8 * r, hipMemcpyHostToDevice));
  hipStream_t st; CK(hipStreamCreate(&st));
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  auto run = [&](const char* name, const double* M, const double* warm, double* V) {
    icp::launch_posterior_eigen(st, r, M, dsl, warm, V, dVt, dS, dwork, dstat + 1);
    hipStreamSynchronize(st);
    float ms = 0; const int reps = 20;
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) icp::launch_posterior_eigen(st, r, M, dsl, warm, V, dVt, dS, dwork, dstat + 1);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    int stat[2]; hipMemcpy(stat, dstat, 8, hipMemcpyDeviceToHost);
    long long s[64]; hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
    printf("%-22s r=%d: %.1f us/call, sweeps %d, status %d | init %.1f warm %.1f setup %.1f", name, r, ms * 1000 / reps, stat[0], stat[1],
           (s[1] - s[0]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01);
    for (int w = 0; w < stat[0] && w < 12; ++w) printf(" | sweep%d %.1f chk %.1f", w, (s[4 + 2 * w] - (w ? s[3 + 2 * w] : s[3])) * 0.01, (s[5 + 2 * w] - s[4 + 2 * w]) * 0.01);
    printf(" | final %.1f || replay after the producer's end: sees it %.1f, rounds done %.1f, output written %.1f\n", (s[63] - s[62]) * 0.01,
           (s[40] - s[63]) * 0.01, (s[41] - s[63]) * 0.01, (s[42] - s[63]) * 0.01);

  };
  {  // ---- posterior factorisation (Cholesky + solve) on the same matrix: Mpart = [M − I, b; bᵀ, 0], one split
    const int n = r + 1;
    std::vector<double> Mp((size_t)n * n, 0.0);
    for (int i = 0; i < r; ++i) for (int j = 0; j < r; ++j) Mp[(size_t)i * n + j] = M0[(size_t)i * r + j] - (i == j ? 1.0 : 0.0);
    for (int i = 0; i < r; ++i) Mp[(size_t)r * n + i] = Mp[(size_t)i * n + r] = nd(rng);
    double *dMp, *dMo, *dal; int* dst2;
    CK(hipMalloc(&dMp, 8 * n * n)); CK(hipMalloc(&dMo, 8 * r * r)); CK(hipMalloc(&dal, 8 * r)); CK(hipMalloc(&dst2, 64));
    CK(hipMemcpy(dMp, Mp.data(), 8 * n * n, hipMemcpyHostToDevice));
    icp::PosteriorFactorIO io{dMp, 1, dMo, dal, dst2, dwork};
    icp::launch_posterior_factor(st, r, 1, &io); hipStreamSynchronize(st);
    float ms = 0; hipEventRecord(a, st);
    for (int i = 0; i < 20; ++i) icp::launch_posterior_factor(st, r, 1, &io);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    long long s[64]; hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
    std::vector<double> al(r); hipMemcpy(al.data(), dal, 8 * r, hipMemcpyDeviceToHost);
    double res = 0;
    for (int i = 0; i < r; ++i) { double t = -Mp[(size_t)r * n + i]; for (int j = 0; j < r; ++j) t += M0[(size_t)i * r + j] * al[j]; res = std::fmax(res, std::fabs(t)); }
    if (r > 127) printf("blocked factor r=%d: assembly %.1f diagonal blocks %.1f panels %.1f trailing %.1f back substitution %.1f us\n", r, s[24] * 0.01, s[25] * 0.01,
                        s[26] * 0.01, s[27] * 0.01, (s[29] - s[28]) * 0.01);
    printf("factor r=%d: %.1f us/call | setup %.1f columns %.1f store %.1f dinv %.1f backsolve %.1f | residual %.2e\n", r, ms * 1000 / 20,
           (s[17] - s[16]) * 0.01, (s[18] - s[17]) * 0.01, (s[19] - s[18]) * 0.01, 0.0, (s[20] - s[19]) * 0.01, res);
  }
  {  // ---- transition tail on the same matrix (fixed-point form)
    std::vector<double> G((size_t)r * r, 0.0), Gi((size_t)r * r, 0.0), cf(r), ct(r), al(r);
    for (int i = 0; i < r; ++i) { G[(size_t)i * r + i] = 1500.0 * sl[i] * sl[i]; Gi[(size_t)i * r + i] = 1.0 / G[(size_t)i * r + i]; cf[i] = nd(rng); ct[i] = cf[i] + 0.1 * nd(rng); al[i] = nd(rng); }
    double *dGi, *dcf, *dct, *dal, *dout; int* dst3;
    CK(hipMalloc(&dGi, 8 * r * r)); CK(hipMalloc(&dcf, 8 * r)); CK(hipMalloc(&dct, 8 * r)); CK(hipMalloc(&dal, 8 * r)); CK(hipMalloc(&dout, 64)); CK(hipMalloc(&dst3, 64));
    CK(hipMemcpy(dGi, Gi.data(), 8 * r * r, hipMemcpyHostToDevice)); CK(hipMemcpy(dcf, cf.data(), 8 * r, hipMemcpyHostToDevice));
    CK(hipMemcpy(dct, ct.data(), 8 * r, hipMemcpyHostToDevice)); CK(hipMemcpy(dal, al.data(), 8 * r, hipMemcpyHostToDevice));
    icp::TransitionTailIO io{dal, dM0, dcf, dct, 0.1, dout, dst3};
    icp::launch_transition_tails(st, r, 1, &io, dGi, 1e-5); hipStreamSynchronize(st);
    float ms = 0; hipEventRecord(a, st);
    for (int i = 0; i < 20; ++i) icp::launch_transition_tails(st, r, 1, &io, dGi, 1e-5);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    int stt; double o; hipMemcpy(&stt, dst3, 4, hipMemcpyDeviceToHost); hipMemcpy(&o, dout, 8, hipMemcpyDeviceToHost);
    printf("tail r=%d: %.1f us/call (status %d, value %.6g)\n", r, ms * 1000 / 20, stt, o);
  }
  {  // ---- the speculative form: input = split-K partials of the regression launch, ready word already raised, cancel word in pinned memory
    const int n = r + 1, S = 13;
    std::vector<double> Mp((size_t)S * n * n, 0.0);
    for (int sp = 0; sp < S; ++sp)
      for (int i = 0; i < r; ++i) for (int j = 0; j <= i; ++j) Mp[(size_t)sp * n * n + (size_t)i * n + j] = (M1[(size_t)i * r + j] - (i == j ? 1.0 : 0.0)) / S;
    double* dMp; int *dready, *hcancel, *hstat;
    CK(hipMalloc(&dMp, 8 * Mp.size())); CK(hipMemcpy(dMp, Mp.data(), 8 * Mp.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&dready, 64)); CK(hipMemset(dready, 0x01, 64));
    CK(hipHostMalloc((void**)&hcancel, 64, hipHostMallocDefault)); CK(hipHostMalloc((void**)&hstat, 64, hipHostMallocDefault));
    hcancel[0] = 0;
    icp::launch_posterior_eigen(st, r, dM0, dsl, nullptr, dV0, dVt, dS, dwork, dstat + 1); hipStreamSynchronize(st);
    icp::EigenSpec spec{S, hcancel, 7, dready, 1};
    float ms = 0; const int reps = 20;
    icp::launch_posterior_eigen(st, r, dMp, dsl, dV0, dV1, dVt, dS, dwork, dstat + 1, &spec, hstat); hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; ++i) icp::launch_posterior_eigen(st, r, dMp, dsl, dV0, dV1, dVt, dS, dwork, dstat + 1, &spec, hstat);
    hipEventRecord(b, st); hipEventSynchronize(b); hipEventElapsedTime(&ms, a, b);
    int stat[2]; hipMemcpy(stat, dstat, 8, hipMemcpyDeviceToHost);
    long long s[64]; hipMemcpyFromSymbol(s, HIP_SYMBOL(icp::g_eigen_stamps), sizeof(s));
    printf("speculative (partials)  r=%d: %.1f us/call, sweeps %d, status %d | staging %.1f wait %.1f assembly %.1f warm %.1f setup %.1f", r, ms * 1000 / reps, stat[0], stat[1],
           (s[50] - s[0]) * 0.01, (s[51] - s[50]) * 0.01, (s[1] - s[51]) * 0.01, (s[2] - s[1]) * 0.01, (s[3] - s[2]) * 0.01);
    for (int w = 0; w < stat[0] && w < 12; ++w) printf(" | sweep%d %.1f chk %.1f", w, (s[4 + 2 * w] - (w ? s[3 + 2 * w] : s[3])) * 0.01, (s[5 + 2 * w] - s[4 + 2 * w]) * 0.01);
    printf(" | final %.1f || replay after the producer's end: sees it %.1f, rounds done %.1f, output written %.1f\n", (s[63] - s[62]) * 0.01,
           (s[40] - s[63]) * 0.01, (s[41] - s[63]) * 0.01, (s[42] - s[63]) * 0.01);
  }
  run("cold", dM0, nullptr, dV0);
  run("warm (3% perturbed)", dM1, dV0, dV1);
  run("warm (same matrix)", dM0, dV0, dV1);
  {  // warm-started decomposition (loose stop + first-order correction) against the cold one of the same matrix
    std::vector<double> Va((size_t)r * r), Vb((size_t)r * r), Sa(r), Sb(r);
    icp::launch_posterior_eigen(st, r, dM0, dsl, nullptr, dV0, dVt, dS, dwork, dstat + 1); hipStreamSynchronize(st);
    icp::launch_posterior_eigen(st, r, dM1, dsl, dV0, dV1, dVt, dS, dwork, dstat + 1); hipStreamSynchronize(st);
    hipMemcpy(Va.data(), dV1, 8 * r * r, hipMemcpyDeviceToHost); hipMemcpy(Sa.data(), dS, 8 * r, hipMemcpyDeviceToHost);
    int stat[2]; hipMemcpy(stat, dstat, 8, hipMemcpyDeviceToHost);
    icp::launch_posterior_eigen(st, r, dM1, dsl, nullptr, dV1, dVt, dS, dwork, dstat + 1); hipStreamSynchronize(st);
    hipMemcpy(Vb.data(), dV1, 8 * r * r, hipMemcpyDeviceToHost); hipMemcpy(Sb.data(), dS, 8 * r, hipMemcpyDeviceToHost);
    double dv = 0, ds = 0, orth = 0, res = 0;
    for (size_t i = 0; i < Va.size(); ++i) dv = std::fmax(dv, std::fabs(Va[i] - Vb[i]));
    for (int i = 0; i < r; ++i) ds = std::fmax(ds, std::fabs(Sa[i] - Sb[i]) / Sb[i]);
    for (int c = 0; c < r; ++c) {
      for (int i = 0; i < r; ++i) {
        double t = 0;
        for (int j = 0; j < r; ++j) t += M1[(size_t)i * r + j] / (sl[i] * sl[j]) * Va[(size_t)j * r + c];
        res = std::fmax(res, std::fabs(t - Va[(size_t)i * r + c] / Sa[c]));
      }
      for (int c2 = 0; c2 < r; ++c2) {
        double t = 0;
        for (int i = 0; i < r; ++i) t += Va[(size_t)i * r + c] * Va[(size_t)i * r + c2];
        orth = std::fmax(orth, std::fabs(t - (c == c2)));
      }
    }
    printf("warm (%d sweeps) vs cold on the perturbed matrix: max|dV| %.3e, max rel dS %.3e, residual %.3e, orthogonality %.3e\n", stat[0], dv, ds, res, orth);
  }
  // check: S of M0 against a host Jacobi-free residual ‖N V − V diag(μ)‖
  std::vector<double> V((size_t)r * r), S(r);
  icp::launch_posterior_eigen(st, r, dM0, dsl, nullptr, dV0, dVt, dS, dwork, dstat + 1); hipStreamSynchronize(st);
  hipMemcpy(V.data(), dV0, 8 * r * r, hipMemcpyDeviceToHost); hipMemcpy(S.data(), dS, 8 * r, hipMemcpyDeviceToHost);
  double maxres = 0, maxorth = 0;
  for (int c = 0; c < r; ++c) {
    for (int i = 0; i < r; ++i) {
      double s = 0;
      for (int j = 0; j < r; ++j) s += M0[(size_t)i * r + j] / (sl[i] * sl[j]) * V[(size_t)j * r + c];
      maxres = std::fmax(maxres, std::fabs(s - V[(size_t)i * r + c] / S[c]));
    }
    for (int c2 = 0; c2 < r; ++c2) {
      double s = 0;
      for (int i = 0; i < r; ++i) s += V[(size_t)i * r + c] * V[(size_t)i * r + c2];
      maxorth = std::fmax(maxorth, std::fabs(s - (c == c2)));
    }
  }
  printf("residual max|N v - mu v| = %.3e, orthogonality %.3e, S[0]=%.6g S[r-1]=%.6g\n", maxres, maxorth, S[0], S[r - 1]);
  return 0;
}
