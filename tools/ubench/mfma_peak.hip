// micro-benchmark: achievable fp32 MFMA rate on gfx950 for different accumulator counts / waves per SIMD
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u)
#pragma unroll
      for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, const char* tag) {
  float* out; hipMalloc(&out, 256 * 8 * 256 * 4);
  int iters = 2000 / NACC;
  int grid = 256 * blocks_per_cu;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<NACC>, dim3(grid), dim3(256), 0, 0, out, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
  double flops = (double)grid * 4 * iters * 16 * NACC * 4096.0;
  printf("%s NACC=%d waves/SIMD=%d : %.3f ms  %.1f TFLOP/s\n", tag, NACC, blocks_per_cu, ms, flops / ms / 1e9);
  hipFree(out);
}
int main() {
  run<1>(1, "dependent chain"); run<2>(1, "2 acc"); run<4>(1, "4 acc"); run<8>(1, "8 acc");
  run<1>(2, "dependent chain"); run<4>(2, "4 acc"); run<1>(4, "dependent chain"); run<4>(3, "4 acc");
  return 0;
}
