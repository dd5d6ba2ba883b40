// micro-benchmark: sustained fp32 MFMA rate, 32x32x2 vs 16x16x4, operands re-read from LDS with ds_read_b128 on
// pseudo-random data, each arm run for about 2 s so the clock the chip actually holds under load is what is measured
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ inline void fill_lds(float* lds, int n, int tid) {
  unsigned s = 1234567u + 977u * blockIdx.x;
  for (int i = tid; i < n; i += 256) {
    unsigned h = (s + i) * 2654435761u;
    h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
    lds[i] = ((int)(h & 0xFFFF) - 32768) * (1.0f / 32768.0f);
  }
}

template <int WPS>
__global__ __launch_bounds__(256, WPS) void k32(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
  fill_lds(lds, 8 * 1080 + 2 * 6912, tid);
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
      const int tapoff = ((kz * 10 + ky) * 18 + kx) * 4 + (it & 1) * 8;
      const f32x4 bw0 = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase + (it & 1) * 8);
      const f32x4 bw1 = *reinterpret_cast<const f32x4*>(ws + 6912 + tap * 256 + bbase + (it & 1) * 8);
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

// 16x16x4: A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15]; 4 A tiles x 4 B tiles per wave, 8 b128 per 64 MFMAs
template <int WPS>
__global__ __launch_bounds__(256, WPS) void k16(float* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 15, lq = lane >> 4;
  fill_lds(lds, 8 * 1080 + 2 * 6912, tid);
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) acc[i][r] = 0.f;
  int abase[4];
  for (int m = 0; m < 4; ++m) { int idx = (wave + 4 * m) * 16 + li; int tx = idx % 16, t = idx / 16, ty = t % 8, tz = t / 8; abase[m] = ((lq & 1) * 1080 + (tz * 10 + ty) * 18 + tx) * 4 + (lq >> 1) * 2160 * 4; }
  const float* ws = lds + 8 * 1080;
  const int bbase = (lq * 16 + li) * 4;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
      const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
      const int tapoff = ((kz * 10 + ky) * 18 + kx) * 4 + (it & 1) * 8;
      f32x4 av[4], bv[4];
#pragma unroll
      for (int n = 0; n < 4; ++n) bv[n] = *reinterpret_cast<const f32x4*>(ws + tap * 256 + n * 1700 + bbase + (it & 1) * 8);
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = *reinterpret_cast<const f32x4*>(lds + abase[m] + tapoff);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n)
            acc[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m][r], bv[n][r], acc[m * 4 + n], 0, 0, 0);
    }
  }
  float s = 0.f;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + tid] = s;
}

template <typename K>
void run(const char* name, K kern, int blocks_per_cu, double flops_per_wave_iter) {
  float* out; (void)hipMalloc(&out, 256 * 8 * 256 * 4);
  const int grid = 256 * blocks_per_cu;
  const size_t lds = (8 * 1080 + 2 * 6912) * 4;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int iters = 400;   // ~ 10 ms per launch
  for (int r = 0; r < 20; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipDeviceSynchronize();
  const int reps = 150;
  (void)hipEventRecord(e0);
  for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, 0, out, iters);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= reps;
  double flops = (double)grid * 4 * iters * flops_per_wave_iter;
  printf("%-28s %d WG/CU: %.3f ms/launch  %.1f TFLOP/s (sustained over %.1f s)\n", name, blocks_per_cu, ms, flops / ms / 1e9, ms * reps / 1e3);
  (void)hipFree(out);
}
int main() {
  const double f = 27 * 32 * 4096.0;   // both kernels: 131072 * 27 flops per wave-iteration
  run("32x32x2 f32, 6 b128/32 MFMA", k32<1>, 1, f);
  run("32x32x2 f32, 6 b128/32 MFMA", k32<2>, 2, f);
  run("16x16x4 f32, 8 b128/64 MFMA", k16<1>, 1, f);
  run("16x16x4 f32, 8 b128/64 MFMA", k16<2>, 2, f);
  run("32x32x2 f32, 6 b128/32 MFMA", k32<2>, 2, f);
  run("16x16x4 f32, 8 b128/64 MFMA", k16<2>, 2, f);
  return 0;
}
