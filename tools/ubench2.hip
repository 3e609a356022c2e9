// ubench2.hip — latency calibration, second round (dev tool, not product): dependent-chain latencies of the
// f64/f32 VALU ops the single-workgroup kernels are bound by, in-kernel clock, and host<->device signalling costs.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench2 tools/ubench2.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

struct Stamp { long long clk, rt; };

template <int ILP, class F>
__device__ void chain(double* out, int n, Stamp* st, F f) {
  double a[ILP];
  for (int u = 0; u < ILP; ++u) a[u] = out[u] + 1.0 + 0.001 * u + 1e-3 * threadIdx.x;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int u = 0; u < ILP; ++u) a[u] = f(a[u]);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  double s = 0;
  for (int u = 0; u < ILP; ++u) s += a[u];
  out[64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}

#define KCHAIN(name, ILP, expr)                                                         \
  __global__ void name(double* out, int n, Stamp* st) {                                \
    chain<ILP>(out, n, st, [](double x) { return expr; });                             \
  }

KCHAIN(k_fma1, 1, fma(x, 1.0000001, 1e-9))
KCHAIN(k_fma2, 2, fma(x, 1.0000001, 1e-9))
KCHAIN(k_fma4, 4, fma(x, 1.0000001, 1e-9))
KCHAIN(k_fma8, 8, fma(x, 1.0000001, 1e-9))
KCHAIN(k_add1, 1, x + 1e-9)
KCHAIN(k_mul1, 1, x * 1.0000001)
KCHAIN(k_rcp1, 1, __builtin_amdgcn_rcp(x) + 0.5)
KCHAIN(k_rsq1, 1, __builtin_amdgcn_rsq(x) + 0.5)
KCHAIN(k_sqrt1, 1, sqrt(x) + 0.5)
KCHAIN(k_div1, 1, 1.0 / x + 0.5)

template <int ILP>
__global__ void k_f32(float* out, int n, Stamp* st) {
  float a[ILP];
  for (int u = 0; u < ILP; ++u) a[u] = out[u] + 1.0f + 0.001f * u;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k)
#pragma unroll
      for (int u = 0; u < ILP; ++u) a[u] = fmaf(a[u], 1.0000001f, 1e-9f);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  float s = 0;
  for (int u = 0; u < ILP; ++u) s += a[u];
  out[64 + threadIdx.x] = s;
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}
__global__ void k_f32rsq(float* out, int n, Stamp* st) {
  float a = out[0] + 1.5f;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) a = __builtin_amdgcn_rsqf(a) + 0.5f; }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = a;
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}
__global__ void k_readlane(double* out, int n, Stamp* st) {
  double x = threadIdx.x + out[0];
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
    int lo = __builtin_amdgcn_readlane(__double2loint(x), (i + k) & 63), hi = __builtin_amdgcn_readlane(__double2hiint(x), (i + k) & 63);
    x = x + __hiloint2double(hi, lo);
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = x;
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}
__global__ void k_lds_rw(double* out, int n, Stamp* st) {  // write -> (wave-synchronous) read -> fma chain through LDS, one wave
  __shared__ double buf[128];
  double x = threadIdx.x + out[0];
  buf[threadIdx.x] = x;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) {
    buf[threadIdx.x] = x;
    x = buf[(threadIdx.x + 1 + k) & 63] + 1.0;
    }
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = x;
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}
__global__ void k_mfma_f64_chain(double* out, int n, Stamp* st) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc = {0, 0, 0, 0};
  double a = out[0] + 1e-3 * threadIdx.x, b = 1.0;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
#pragma unroll
    for (int k = 0; k < 16; ++k) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}
__global__ void k_mfma_f64_indep(double* out, int n, Stamp* st) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  d4 acc0 = {0, 0, 0, 0}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
  double a = out[0] + 1e-3 * threadIdx.x, b = 1.0;
  long long r0 = __builtin_amdgcn_s_memrealtime();
  long long t0 = __builtin_amdgcn_s_memtime();
  for (int i = 0; i < n; ++i) {
    acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
    acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc2, 0, 0, 0);
    acc3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc3, 0, 0, 0);
  }
  long long t1 = __builtin_amdgcn_s_memtime();
  long long r1 = __builtin_amdgcn_s_memrealtime();
  out[64 + threadIdx.x] = acc0[0] + acc1[1] + acc2[2] + acc3[3];
  if (threadIdx.x == 0) { st->clk = t1 - t0; st->rt = r1 - r0; }
}

// busy kernel for the "is the clock load dependent" test: many CUs spin on fma for ~ms
__global__ void k_busy(double* out, int n) {
  double a = threadIdx.x * 1e-3, b = 1.0000001;
  for (int i = 0; i < n; ++i) { a = fma(a, b, 1e-9); b = fma(b, 1.0000001, 1e-12); }
  if (a == 12345.678) out[0] = a + b;
}

__global__ void k_signal(volatile int* flag, int v) {
  flag[0] = v;
  __threadfence_system();
}
__global__ void k_empty() {}
__global__ void k_write_host(double* host, double v) { host[threadIdx.x] = v; }

template <class K, class T>
static void run(const char* name, K kern, T* d, Stamp* dst, int n, int per_iter_ops) {
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, n, dst);
  hipDeviceSynchronize();
  hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, d, n, dst);
  hipDeviceSynchronize();
  Stamp s; hipMemcpy(&s, dst, sizeof(s), hipMemcpyDeviceToHost);
  double ns = s.rt * 10.0;  // s_memrealtime: 100 MHz
  printf("%-34s %7.2f clk/step %7.2f ns/step   (clock %.0f MHz, %d ops/step; 16 steps unrolled)\n", name, (double)s.clk / n / 16, ns / n / 16,
         s.clk / (ns * 1e-3), per_iter_ops);
}

int main() {
  double* d; CK(hipMalloc(&d, 65536)); CK(hipMemset(d, 0, 65536));
  Stamp* dst; CK(hipMalloc(&dst, sizeof(Stamp)));
  const int n = 2000;
  run("f64 fma dep, ILP1", k_fma1, d, dst, n, 1);
  run("f64 fma dep, ILP2", k_fma2, d, dst, n, 2);
  run("f64 fma dep, ILP4", k_fma4, d, dst, n, 4);
  run("f64 fma dep, ILP8", k_fma8, d, dst, n, 8);
  run("f64 add dep", k_add1, d, dst, n, 1);
  run("f64 mul dep", k_mul1, d, dst, n, 1);
  run("f64 rcp seed + add", k_rcp1, d, dst, n, 2);
  run("f64 rsq seed + add", k_rsq1, d, dst, n, 2);
  run("f64 sqrt() + add", k_sqrt1, d, dst, n, 2);
  run("f64 1/x + add", k_div1, d, dst, n, 2);
  run("f32 fma dep, ILP1", k_f32<1>, (float*)d, dst, n, 1);
  run("f32 fma dep, ILP4", k_f32<4>, (float*)d, dst, n, 4);
  run("f32 rsq + add", k_f32rsq, (float*)d, dst, n, 2);
  run("readlane(f64) + add", k_readlane, d, dst, n, 3);
  run("lds write->read + add (1 wave)", k_lds_rw, d, dst, n, 3);
  run("mfma f64 16x16x4 dep chain", k_mfma_f64_chain, d, dst, n, 1);
  run("mfma f64 16x16x4 x4 indep", k_mfma_f64_indep, d, dst, n, 4);

  // clock under background load: a busy kernel on another stream while the chain runs
  hipStream_t s2; CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipLaunchKernelGGL(k_busy, dim3(2000), dim3(256), 0, s2, d + 4096, 400000);
  hipLaunchKernelGGL(k_fma1, dim3(1), dim3(64), 0, 0, d, n, dst);
  hipStreamSynchronize(0);
  { Stamp s; hipMemcpy(&s, dst, sizeof(s), hipMemcpyDeviceToHost);
    printf("f64 fma dep under background load: %.2f clk/iter %.2f ns/iter (clock %.0f MHz)\n", (double)s.clk / n, s.rt * 10.0 / n, s.clk / (s.rt * 10.0 * 1e-3)); }
  hipDeviceSynchronize();

  // ---- host <-> device signalling
  int* hflag; CK(hipHostMalloc((void**)&hflag, 64, hipHostMallocDefault));
  double* hbuf; CK(hipHostMalloc((void**)&hbuf, 4096, hipHostMallocDefault));
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  using clk = std::chrono::steady_clock;
  auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
  for (int rep = 0; rep < 2; ++rep) {
    const int R = 2000;
    auto t0 = clk::now();
    for (int i = 0; i < R; ++i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st); hipStreamSynchronize(st); }
    auto t1 = clk::now();
    printf("launch empty + hipStreamSynchronize: %.2f us\n", us(t0, t1) / R);
    t0 = clk::now();
    for (int i = 0; i < R; ++i) {
      hflag[0] = 0;
      hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, (volatile int*)hflag, i + 1);
      while (((volatile int*)hflag)[0] != i + 1) {}
    }
    t1 = clk::now();
    printf("launch signal kernel + host poll on pinned flag: %.2f us\n", us(t0, t1) / R);
    hipStreamSynchronize(st);
    t0 = clk::now();
    for (int i = 0; i < R; ++i) {
      for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
      hflag[0] = 0;
      hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, st, (volatile int*)hflag, i + 1);
      while (((volatile int*)hflag)[0] != i + 1) {}
    }
    t1 = clk::now();
    printf("5 empty + signal kernel + host poll: %.2f us  (per launch %.2f)\n", us(t0, t1) / R, us(t0, t1) / R / 6);
    hipStreamSynchronize(st);
    t0 = clk::now();
    for (int i = 0; i < R; ++i) {
      for (int k = 0; k < 5; ++k) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
      hipLaunchKernelGGL(k_write_host, dim3(1), dim3(64), 0, st, hbuf, (double)i);
      hipStreamSynchronize(st);
    }
    t1 = clk::now();
    printf("5 empty + write-host kernel + hipStreamSynchronize: %.2f us\n", us(t0, t1) / R);
    double* dsrc; hipMalloc(&dsrc, 4096);
    t0 = clk::now();
    for (int i = 0; i < R; ++i) {
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
      hipMemcpyAsync(hbuf, dsrc, 512, hipMemcpyDeviceToHost, st);
      hipStreamSynchronize(st);
    }
    t1 = clk::now();
    printf("empty + memcpyAsync D2H 512B + sync: %.2f us\n", us(t0, t1) / R);
    t0 = clk::now();
    for (int i = 0; i < R; ++i) {
      hipMemcpyAsync(dsrc, hbuf, 512, hipMemcpyHostToDevice, st);
      hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
      hipStreamSynchronize(st);
    }
    t1 = clk::now();
    printf("memcpyAsync H2D 512B + empty + sync: %.2f us\n", us(t0, t1) / R);
    hipFree(dsrc);
  }
  return 0;
}
