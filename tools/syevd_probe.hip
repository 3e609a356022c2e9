// dev probe: how long does the library eigensolver take for the ranks the hand-written kernel does not cover?
#include <hip/hip_runtime.h>
#include <rocsolver/rocsolver.h>
#include <cstdio>
#include <vector>
#include <random>
int main(int argc, char** argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 200;
  std::mt19937_64 rng(1); std::normal_distribution<double> nd;
  std::vector<double> B((size_t)n * n), A((size_t)n * n, 0.0);
  for (auto& x : B) x = nd(rng);
  for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { double s = i == j ? 1.0 : 0.0; for (int k = 0; k < n; ++k) s += B[(size_t)i * n + k] * B[(size_t)j * n + k] / n; A[(size_t)i * n + j] = s; }
  double *dA, *dD, *dE; int* dinfo;
  hipMalloc(&dA, sizeof(double) * n * n); hipMalloc(&dD, sizeof(double) * n); hipMalloc(&dE, sizeof(double) * n); hipMalloc(&dinfo, 4);
  rocblas_handle h; rocblas_create_handle(&h);
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int which = 0; which < 2; ++which) {
    for (int rep = 0; rep < 4; ++rep) {
      hipMemcpy(dA, A.data(), sizeof(double) * n * n, hipMemcpyHostToDevice);
      hipEventRecord(a, 0);
      if (which == 0) rocsolver_dsyevd(h, rocblas_evect_original, rocblas_fill_upper, n, dA, n, dD, dE, dinfo);
      else { double res; int sw; rocsolver_dsyevj(h, rocblas_esort_ascending, rocblas_evect_original, rocblas_fill_upper, n, dA, n, 1e-14, (double*)dE, 30, (int*)dinfo + 0, dD, dinfo); (void)res; (void)sw; }
      hipEventRecord(b, 0); hipEventSynchronize(b);
      float ms; hipEventElapsedTime(&ms, a, b);
      printf("%s n=%d rep %d: %.3f ms\n", which == 0 ? "dsyevd" : "dsyevj", n, rep, ms);
    }
  }
  return 0;
}
