// v_mfma_f64_16x16x4_f64 issue rate on gfx950: NW waves per SIMD, four independent accumulators, back to back.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/mfma_f64_bench tools/src/mfma_f64_bench.hip ; run: tools/mfma_f64_bench
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4_t __attribute__((ext_vector_type(4)));
__global__ void __launch_bounds__(256) k(int iters, double* out, long long* cycles) {
  d4_t a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + threadIdx.x * 1e-6;
  const long long t0 = __builtin_readcyclecounter();
  for (int i = 0; i < iters; ++i) {
    a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
    a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, x, a1, 0, 0, 0);
    a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, x, a2, 0, 0, 0);
    a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, a3, 0, 0, 0);
  }
  const long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = a0[0] + a1[1] + a2[2] + a3[3];
  if (threadIdx.x == 0 && blockIdx.x == 0) cycles[0] = t1 - t0;
}
int main() {
  double* out; long long* cyc;
  hipMalloc(&out, sizeof(double) * 256 * 4096); hipMalloc(&cyc, 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 20000;
  for (int blocks : {1, 256, 512, 1024}) {
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, 100, out, cyc);
    hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, iters, out, cyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double n = 4.0 * iters;
    printf("blocks %4d (x4 waves): %.3f ms, %.1f ns per MFMA per wave, %.1f shader-clock counts per MFMA, %.1f TFLOP/s\n", blocks, ms, ms * 1e6 / n,
           (double)c / n, blocks * 4 * n * 2048 / (ms * 1e-3) / 1e12);
  }
  return 0;
}
