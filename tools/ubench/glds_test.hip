// checks that LDS-DMA (global_load_lds_dwordx4) reaches every part of the 160 KB LDS (M0 base above 64 KB) on gfx950
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
__global__ __launch_bounds__(256) void k(const float* __restrict__ src, float* __restrict__ dst, int npieces) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int p = wave; p < npieces; p += 4)
    __builtin_amdgcn_global_load_lds(src + (size_t)blockIdx.x * npieces * 256 + (p * 64 + lane) * 4, lds + p * 256, 16, 0, 0);
  __syncthreads();
  for (int i = threadIdx.x; i < npieces * 256; i += 256) dst[(size_t)blockIdx.x * npieces * 256 + i] = lds[i];
}
int main() {
  const int npieces = 156, blocks = 512;  // 156 KiB of LDS per workgroup
  const size_t n = (size_t)blocks * npieces * 256;
  float* h = (float*)malloc(n * 4);
  for (size_t i = 0; i < n; ++i) h[i] = (float)(i % 1000003) * 0.5f;
  float *s, *d;
  (void)hipMalloc(&s, n * 4); (void)hipMalloc(&d, n * 4);
  (void)hipMemcpy(s, h, n * 4, hipMemcpyHostToDevice);
  (void)hipMemset(d, 0, n * 4);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL(k, dim3(blocks), dim3(256), npieces * 1024, 0, s, d, npieces);
  hipError_t e = hipDeviceSynchronize();
  float* o = (float*)malloc(n * 4);
  (void)hipMemcpy(o, d, n * 4, hipMemcpyDeviceToHost);
  size_t bad = 0, first = 0;
  for (size_t i = 0; i < n; ++i) if (o[i] != h[i]) { if (!bad) first = i; ++bad; }
  printf("status %s, mismatches %zu of %zu (first at %zu, piece %zu)\n", hipGetErrorString(e), bad, n, first, (first / 256) % npieces);
  return bad != 0;
}
