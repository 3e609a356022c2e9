// ubench.hip — latency calibration of the building blocks of the single-workgroup r-space kernels (dev tool, not product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__global__ void k_fma_chain(double* out, int n) {
  double a = out[0], b = 1.0000001, c = 1e-9;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) a = fma(a, b, c);
  long long t1 = clock64();
  out[threadIdx.x] = a; if (threadIdx.x == 0) ((long long*)out)[600] = t1 - t0;
}
__global__ void k_barrier(double* out, int n) {
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) __syncthreads();
  long long t1 = clock64();
  if (threadIdx.x == 0) ((long long*)out)[600] = t1 - t0;
}
__global__ void k_lds_chain(double* out, int n) {
  __shared__ int idx[1024];
  for (int i = threadIdx.x; i < 1024; i += blockDim.x) idx[i] = (i * 7 + 1) & 1023;
  __syncthreads();
  int j = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) j = idx[j];
  long long t1 = clock64();
  out[threadIdx.x] = j; if (threadIdx.x == 0) ((long long*)out)[600] = t1 - t0;
}
__global__ void k_lds_barrier_step(double* out, int n) {  // mimics a factor column: lds read -> rcp-ish chain -> lds rw -> barrier
  __shared__ double col[2][520];
  for (int i = threadIdx.x; i < 520; i += blockDim.x) { col[0][i] = 1.0 + i; col[1][i] = 2.0 + i; }
  __syncthreads();
  double v[6]; for (int m = 0; m < 6; ++m) v[m] = threadIdx.x + m;
  long long t0 = clock64();
  for (int j = 0; j < n; ++j) {
    const double* cur = col[j & 1]; double* nxt = col[(j + 1) & 1];
    double a = cur[j % 50];
    double y = __builtin_amdgcn_rcp(a); double e = fma(-a, y, 1.0); y = fma(y, e, y); e = fma(-a, y, 1.0); y = fma(y, e, y);
#pragma unroll
    for (int m = 0; m < 6; ++m) { v[m] = fma(-(cur[(threadIdx.x + m) & 63] * y), cur[(threadIdx.x * 3 + m) & 63], v[m]); }
    if ((threadIdx.x & 7) == 0) nxt[threadIdx.x & 63] = v[0];
    __syncthreads();
  }
  long long t1 = clock64();
  out[threadIdx.x] = v[0] + v[5]; if (threadIdx.x == 0) ((long long*)out)[600] = t1 - t0;
}
__global__ void k_shfl_chain(double* out, int n) {
  double x = threadIdx.x;
  long long t0 = clock64();
  for (int i = 0; i < n; ++i) x = __shfl(x, (i * 5) & 63, 64) + 1.0;
  long long t1 = clock64();
  out[threadIdx.x] = x; if (threadIdx.x == 0) ((long long*)out)[600] = t1 - t0;
}
__global__ void k_empty(double* out) { if (out == nullptr) out[0] = 1; }

template <class F> float timeit(F f, int reps) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f(); hipDeviceSynchronize();
  hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 1000.f / reps;
}
int main() {
  double* d; CK(hipMalloc(&d, 8192)); CK(hipMemset(d, 0, 8192));
  long long cyc; const int n = 20000;
  auto rd = [&]() { hipMemcpy(&cyc, ((long long*)d) + 600, 8, hipMemcpyDeviceToHost); return cyc; };
  float us;
  us = timeit([&] { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, 0, d); }, 200); printf("empty kernel back-to-back: %.2f us/launch\n", us);
  us = timeit([&] { hipLaunchKernelGGL(k_fma_chain, dim3(1), dim3(64), 0, 0, d, n); }, 5); rd();
  printf("fma f64 chain: %.1f clk/op  (%.1f us total, clock64 rate %.1f MHz)\n", (double)cyc / n, us, cyc / us);
  for (int nt : {64, 256, 1024}) {
    us = timeit([&] { hipLaunchKernelGGL(k_barrier, dim3(1), dim3(nt), 0, 0, d, n); }, 5); rd();
    printf("barrier nt=%4d: %.1f clk  %.3f us each\n", nt, (double)cyc / n, us / n);
  }
  for (int nt : {64, 256, 1024}) {
    us = timeit([&] { hipLaunchKernelGGL(k_lds_chain, dim3(1), dim3(nt), 0, 0, d, n); }, 5); rd();
    printf("lds dependent read nt=%4d: %.1f clk  %.3f us each\n", nt, (double)cyc / n, us / n);
  }
  for (int nt : {256, 1024}) {
    us = timeit([&] { hipLaunchKernelGGL(k_lds_barrier_step, dim3(1), dim3(nt), 0, 0, d, n); }, 5); rd();
    printf("factor-like column step nt=%4d: %.1f clk  %.3f us each\n", nt, (double)cyc / n, us / n);
  }
  us = timeit([&] { hipLaunchKernelGGL(k_shfl_chain, dim3(1), dim3(64), 0, 0, d, n); }, 5); rd();
  printf("shfl chain: %.1f clk  %.3f us each\n", (double)cyc / n, us / n);
  return 0;
}
