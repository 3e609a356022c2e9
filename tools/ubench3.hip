// ubench3.hip — dev: what a streaming 16-byte-per-lane read costs a workgroup when the chip is full of such workgroups
// (the batched filter launch: 7,296 workgroups of 256 threads, each 8 KB of a 1.9 MB array that 32 of them share)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("err %s line %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
__device__ unsigned long long g_acc[2048][4];  // (spread: one line for all workgroups would time the atomics)
template <int LDS_KB>
__global__ void __launch_bounds__(256) k_read(const float4* __restrict__ a, int n_x, float* out, int dep) {
  __shared__ float s_pad[LDS_KB * 256];
  const long long t0 = __builtin_amdgcn_s_memrealtime();
  const int t = (blockIdx.x * 256 + threadIdx.x) * 2;
  float4 u = a[t], v = a[t + 1];
  float acc = u.x + u.y + u.z + u.w + v.x + v.y + v.z + v.w;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t1 = __builtin_amdgcn_s_memrealtime();
  if (dep) {  // a second, dependent trip (the query records)
    const int j = ((int)fabsf(acc) & 1023) + blockIdx.y * 1024;
    acc += a[j & (n_x * 512 - 1)].x;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const long long t2 = __builtin_amdgcn_s_memrealtime();
  s_pad[threadIdx.x] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long* g = g_acc[(blockIdx.x + 977u * blockIdx.y) % 2048];
    atomicAdd(&g[0], 1ull); atomicAdd(&g[1], (unsigned long long)(t1 - t0)); atomicAdd(&g[2], (unsigned long long)(t2 - t1));
    out[blockIdx.y * n_x + blockIdx.x] = s_pad[(threadIdx.x + 17) & 255];
  }
}
int main() {
  const int n_x = 228;
  const size_t n = (size_t)n_x * 512 + 1024;
  float4* a; float* out;
  CK(hipMalloc(&a, n * sizeof(float4))); CK(hipMemset(a, 0, n * sizeof(float4)));
  CK(hipMalloc(&out, sizeof(float) * n_x * 64));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int lds = 0; lds < 2; ++lds)
    for (int dep = 0; dep < 2; ++dep)
      for (int B : {1, 4, 8, 16, 32, 64}) {
        static unsigned long long z[2048][4];
        float ms = 0;
        for (int rep = 0; rep < 3; ++rep) {
          CK(hipMemcpyToSymbol(HIP_SYMBOL(g_acc), z, sizeof(z)));
          CK(hipEventRecord(e0));
          if (lds) hipLaunchKernelGGL(k_read<10>, dim3(n_x, B), dim3(256), 0, 0, a, n_x, out, dep);
          else hipLaunchKernelGGL(k_read<1>, dim3(n_x, B), dim3(256), 0, 0, a, n_x, out, dep);
          CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
          CK(hipEventElapsedTime(&ms, e0, e1));
        }
        static unsigned long long h[2048][4];
        CK(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_acc), sizeof(h)));
        unsigned long long acc[4] = {0, 0, 0, 0};
        for (int i = 0; i < 2048; ++i) for (int k = 0; k < 4; ++k) acc[k] += h[i][k];
        printf("lds %2d KB dep %d  grid %3d x %2d (%5d workgroups): kernel %.1f us | per workgroup: first trip %.2f us, second %.2f us\n", lds ? 10 : 1, dep, n_x, B,
               n_x * B, ms * 1e3, acc[1] / (double)acc[0] / 100.0, acc[2] / (double)acc[0] / 100.0);
      }
  return 0;
}
