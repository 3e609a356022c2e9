// gpu_hog.hip — test support (tests/test_gpu_shared.py): a SECOND PROCESS that keeps every compute unit of the device busy for a given
// number of seconds, the way another tenant's job would: back-to-back chip-wide launches on two streams, each workgroup a mix of f64
// multiply-adds, loads that miss the L1 and LDS traffic for ~40 µs.  Prints "running" once the first launches are on the device and
// "done <launches>" at the end.  Not part of the product, not linked into it.
// usage: gpu_hog <seconds> [workgroups per launch, default 4096]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("error %s (line %d)\n", hipGetErrorString(e_), __LINE__); return 2; } } while (0)

__global__ void __launch_bounds__(256) k_hog(unsigned long long ticks, const double* __restrict__ buf, size_t n, double* sink) {
  __shared__ double lds[2048];
  const unsigned long long until = __builtin_amdgcn_s_memrealtime() + ticks;  // 100 MHz
  const int tid = threadIdx.x;
  double x = tid * 1e-3, y = 1.0;
  size_t pos = ((size_t)blockIdx.x * 256 + tid) * 8 % n;
  lds[tid] = x;
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() < until) {
#pragma unroll
    for (int i = 0; i < 32; ++i) { x = fma(x, 1.0000001, y); y = fma(y, 0.9999999, x); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { x += buf[pos]; pos = (pos + 256 * 1031) % n; }
#pragma unroll
    for (int i = 0; i < 8; ++i) x += lds[(tid * 17 + i * 33) & 2047];
    lds[(tid + 1) & 2047] = x;
  }
  if (x == 123.456) sink[0] = x + y;
}

int main(int argc, char** argv) {
  const double seconds = argc > 1 ? std::atof(argv[1]) : 5.0;
  const int wgs = argc > 2 ? std::atoi(argv[2]) : 4096;
  CK(hipSetDevice(0));
  const size_t n = (size_t)8 << 20;  // 64 MiB of doubles
  double *buf = nullptr, *sink = nullptr;
  CK(hipMalloc(&buf, n * sizeof(double)));
  CK(hipMalloc(&sink, 64));
  CK(hipMemset(buf, 0, n * sizeof(double)));
  hipStream_t st[2];
  for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  const auto t0 = std::chrono::steady_clock::now();
  long launches = 0;
  bool said = false;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < seconds) {
    for (auto& s : st)
      for (int k = 0; k < 8; ++k) { hipLaunchKernelGGL(k_hog, dim3(wgs), dim3(256), 0, s, 4000ull /* 40 µs */, buf, n, sink); ++launches; }
    if (!said) { std::printf("running\n"); std::fflush(stdout); said = true; }
    for (auto& s : st) CK(hipStreamSynchronize(s));
  }
  std::printf("done %ld\n", launches);
  return 0;
}
