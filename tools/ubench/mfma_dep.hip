#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
// MODE 0: one VGPR accumulator, every MFMA depends on the previous; 1: the same in an AGPR; 2: pairs of dependent MFMAs over 16 AGPR accumulators (the 2-D Winograd pattern)
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  const int tid = threadIdx.x;
  float a = 1.0f + tid * 1e-3f, b = 0.5f - tid * 1e-3f;
  f32x16 c[16];
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 16; ++i) c[j][i] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 32; ++u) {
      if (MODE == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[0]) : "v"(a), "v"(b));
      if (MODE == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[0]) : "v"(a), "v"(b));
      if (MODE == 2) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[(u >> 1) & 15]) : "v"(a), "v"(b));
      if (MODE == 3) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+a"(c[u & 15]) : "v"(a), "v"(b));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int j = 0; j < 16; ++j) for (int i = 0; i < 16; ++i) s += c[j][i];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char* name, float* out, long long* cyc) {
  const int iters = 2000;
  hipLaunchKernelGGL((k<MODE>), dim3(256), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  long long h[2];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-70s %6.1f cycles per MFMA\n", name, (double)h[0] / (iters * 32.0));
}
int main() {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  run<0>("one VGPR accumulator, dependent chain", out, cyc);
  run<1>("one AGPR accumulator, dependent chain", out, cyc);
  run<2>("16 AGPR accumulators, dependent pairs (2-D Winograd pattern)", out, cyc);
  run<3>("16 AGPR accumulators, round robin", out, cyc);
  return 0;
}
