// micro-benchmark: MFMA rate when operands are re-read from LDS like the conv inner loop (5 x ds_read_b128 per 16 MFMAs)
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int PAT, int WPS>
__global__ __launch_bounds__(256, WPS) void k(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 8 * 1080 + 6912; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[4];
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int abase[4];
  for (int m = 0; m < 4; ++m) {
    int idx = (wave + 4 * m) * 32 + li; int tx = idx % 16, t = idx / 16, ty = t % 8, tz = t / 8;
    if (PAT == 0) abase[m] = (lh * 1080 + (tz * 10 + ty) * 18 + tx) * 4;          // conv layout: rows of 18 voxels
    else if (PAT == 1) abase[m] = (lh * 1080 + idx) * 4;                            // fully contiguous per half
    else abase[m] = (lh * 1080 + (tz * 10 + ty) * 20 + tx) * 4;                     // padded row stride 20
  }
  const float* ws = lds + 8 * 1080;
  const int bbase = (lh * 32 + li) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int tapoff = ((kz * 10 + ky) * 18 + kx) * 4;
      const f32x4 bw = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase);
      f32x4 av[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(lds + abase[m] + tapoff);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bw[r], acc[m], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k2(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  for (int i = tid; i < 8 * 1080 + 2 * 6912; i += 256) lds[i] = 0.001f * (i % 97);
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  int abase[4];
  for (int m = 0; m < 4; ++m) { int idx = (wave + 4 * m) * 32 + li; int tx = idx % 16, t = idx / 16, ty = t % 8, tz = t / 8; abase[m] = (lh * 1080 + (tz * 10 + ty) * 18 + tx) * 4; }
  const float* ws = lds + 8 * 1080;
  const int bbase = (lh * 32 + li) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int tapoff = ((kz * 10 + ky) * 18 + kx) * 4;
      const f32x4 bw0 = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase);
      const f32x4 bw1 = *reinterpret_cast<const f32x4*>(ws + 6912 + tap * 256 + bbase);
      f32x4 av[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(lds + abase[m] + tapoff);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bw0[r], acc[m], 0, 0, 0);
          acc[4 + m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bw1[r], acc[4 + m], 0, 0, 0);
        }
    }
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}
template <int WPS>
void run2(int blocks_per_cu) {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 20, grid = 256 * blocks_per_cu;
  const size_t lds = (8 * 1080 + 2 * 6912) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k2<WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k2<WPS>), dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k2<WPS>), dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)grid * 4 * iters * 27 * 32 * 4096.0;
  printf("4A+2B per 32 MFMAs, %d WG/CU: %.3f ms  %.1f TFLOP/s\n", blocks_per_cu, ms, flops / ms / 1e9);
  (void)hipFree(out);
}
template <int PAT, int WPS>
void run(int blocks_per_cu) {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  const int iters = 40, grid = 256 * blocks_per_cu;
  const size_t lds = (8 * 1080 + 6912) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k<PAT, WPS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  hipLaunchKernelGGL((k<PAT, WPS>), dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<PAT, WPS>), dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)grid * 4 * iters * 27 * 16 * 4096.0;
  printf("pattern %d: LDS-fed MFMA loop, %d WG/CU (%d waves/SIMD): %.3f ms  %.1f TFLOP/s\n", PAT, blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9);
  (void)hipFree(out);
}
int main() { run<0,1>(1); run<0,2>(2); run2<1>(1); run2<2>(2); run<0,2>(2); run2<2>(2); return 0; }
