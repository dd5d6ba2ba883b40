// micro-benchmark: do fp32 VALU FMAs issue in the shadow of MFMAs?  One wave (or two) per SIMD runs a loop of
// [1 MFMA + V independent v_fma_f32] groups written as one asm block per group, for the fp32 MFMAs (16x16x4, 32x32x2) and
// the bf16 32x32x16 MFMA.  If the VALU work hides, cycles per group stay at the MFMA's issue interval until V x 4 cycles
// exceed it; if fp32 MFMA and fp32 VALU share the datapath, cycles per group grow as interval + V x c from V = 1 on.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/mfma_valu.hip -o tools/ubench/mfma_valu
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// FILL: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_mov_b32, 3 v_mov_b32_dpp, 4 v_add_u32, 5 v_cndmask_b32, 6 ds_write_b32, 7 s_add_u32,
//       8 ds_read_b64, 9 ds_read_b128 (results waited for at the end of the loop only)
template <int KIND, int V, int FILL = 0>
__global__ __launch_bounds__(256) void k(float* out, long long* cyc, int iters) {
  __shared__ float lds[1024];
  const int tid = threadIdx.x;
  float a = 1.0f + tid * 1e-3f, b = 0.5f - tid * 1e-3f;
  f32x4 c4[2] = {{0, 0, 0, 0}, {0, 0, 0, 0}};
  f32x16 c16[2];
  for (int i = 0; i < 16; ++i) c16[0][i] = c16[1][i] = 0.f;
  f32x4 ab = {a, b, a, b}, bb = {b, a, b, a};
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = tid * 0.25f + i;
  const float m = 1.0001f, d = 1e-3f;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  f32x2 p2[4] = {{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}, {7.f, 8.f}};
  const f32x2 pm = {1.0001f, 0.9999f}, pd = {1e-3f, 2e-3f};
  int iv[8];
  for (int i = 0; i < 8; ++i) iv[i] = tid + i;
  int sv = 0;
  lds[tid] = 0.f;
  const long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 16; ++u) {
      if (KIND == 0) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+v"(c4[u & 1]) : "v"(a), "v"(b));
      if (KIND == 1) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c16[u & 1]) : "v"(a), "v"(b));
      if (KIND == 2) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c16[u & 1]) : "v"(ab), "v"(bb));
#pragma unroll
      for (int v = 0; v < V; ++v) {
        if (FILL == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f[v & 7]) : "v"(m), "v"(d));
        if (FILL == 1) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p2[v & 3]) : "v"(pm), "v"(pd));
        if (FILL == 2) asm volatile("v_mov_b32 %0, %1" : "=v"(f[v & 7]) : "v"(m));
        if (FILL == 3) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(f[v & 7]) : "v"(m));
        if (FILL == 4) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iv[v & 7]) : "v"(tid));
        if (FILL == 5) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(f[v & 7]) : "v"(m));
        if (FILL == 6) asm volatile("ds_write_b32 %0, %1" : : "v"(tid * 4), "v"(m) : "memory");
        if (FILL == 7) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv));
        if (FILL == 8) asm volatile("ds_read_b64 %0, %1" : "=v"(p2[v & 3]) : "v"(tid * 8) : "memory");
        if (FILL == 9) asm volatile("ds_read_b128 %0, %1" : "=v"(ab) : "v"(tid * 16) : "memory");
      }
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += c4[0][i] + c4[1][i];
  for (int i = 0; i < 16; ++i) s += c16[0][i] + c16[1][i];
  for (int i = 0; i < 4; ++i) s += p2[i][0] + p2[i][1];
  for (int i = 0; i < 8; ++i) s += (float)iv[i];
  out[blockIdx.x * 256 + tid] = s + (float)sv + lds[(tid + 1) & 255];
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND, int V, int FILL = 0>
static void run(const char* name, int wgs_per_cu, float* out, long long* cyc) {
  const int iters = 2000;
  const int grid = 256 * wgs_per_cu;
  hipLaunchKernelGGL((k<KIND, V, FILL>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0, 0);
  hipLaunchKernelGGL((k<KIND, V, FILL>), dim3(grid), dim3(256), 0, 0, out, cyc, iters);
  hipEventRecord(e1, 0);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  long long h[8];
  hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  static const char* fills[] = {"v_fma_f32", "v_pk_fma_f32", "v_mov_b32", "v_mov_b32_dpp", "v_add_u32", "v_cndmask_b32", "ds_write_b32", "s_add_u32", "ds_read_b64", "ds_read_b128"};
  printf("%-14s + %-14s V=%2d waves/SIMD=%d  %8.1f cycles per (MFMA + V fma) group (s_memtime, wave 0 of block 0)  %7.3f ms\n", name, fills[FILL], V,
         wgs_per_cu, (double)h[0] / (iters * 16.0), ms);
}

int main() {
  float* out;
  long long* cyc;
  hipMalloc(&out, 256 * 8 * 256 * 4);
  hipMalloc(&cyc, 256 * 8 * 8);
  for (int w = 1; w <= 2; ++w) {
    run<0, 0>("f32 16x16x4", w, out, cyc); run<0, 2>("f32 16x16x4", w, out, cyc); run<0, 4>("f32 16x16x4", w, out, cyc);
    run<0, 8>("f32 16x16x4", w, out, cyc);
    run<1, 0>("f32 32x32x2", w, out, cyc); run<1, 4>("f32 32x32x2", w, out, cyc); run<1, 8>("f32 32x32x2", w, out, cyc);
    run<1, 16>("f32 32x32x2", w, out, cyc);
    run<2, 0>("bf16 32x32x16", w, out, cyc); run<2, 2>("bf16 32x32x16", w, out, cyc); run<2, 4>("bf16 32x32x16", w, out, cyc);
    run<2, 8>("bf16 32x32x16", w, out, cyc);
  }
  // what each instruction class costs beside the fp32 16x16x4 MFMA (one wave per SIMD, 4 fillers per MFMA)
  run<0, 4, 1>("f32 16x16x4", 1, out, cyc); run<0, 4, 2>("f32 16x16x4", 1, out, cyc); run<0, 4, 3>("f32 16x16x4", 1, out, cyc);
  run<0, 4, 4>("f32 16x16x4", 1, out, cyc); run<0, 4, 5>("f32 16x16x4", 1, out, cyc); run<0, 4, 6>("f32 16x16x4", 1, out, cyc);
  run<0, 4, 7>("f32 16x16x4", 1, out, cyc);
  run<1, 8, 1>("f32 32x32x2", 1, out, cyc); run<1, 8, 2>("f32 32x32x2", 1, out, cyc); run<1, 8, 4>("f32 32x32x2", 1, out, cyc);
  run<1, 8, 6>("f32 32x32x2", 1, out, cyc); run<1, 8, 7>("f32 32x32x2", 1, out, cyc);
  run<1, 1, 8>("f32 32x32x2", 1, out, cyc); run<1, 2, 8>("f32 32x32x2", 1, out, cyc); run<1, 4, 8>("f32 32x32x2", 1, out, cyc);
  run<1, 1, 9>("f32 32x32x2", 1, out, cyc); run<1, 2, 9>("f32 32x32x2", 1, out, cyc); run<1, 1, 6>("f32 32x32x2", 1, out, cyc);
  return 0;
}
