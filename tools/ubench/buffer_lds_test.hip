// does `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) zero-fill the LDS slots of lanes whose offset is
// out of range, and where do the 16 bytes of lane l land?  One wave: lane l reads 16 bytes at byte offset off[l] of a buffer whose
// resource covers `nbytes` bytes; lanes with off[l] >= nbytes (also "negative" = wrapped offsets) must leave zeros.
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/buffer_lds_test.hip -o tools/ubench/buffer_lds_test
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) float lds_float;

__global__ void k(const float* src, unsigned nbytes, const unsigned* off, float* out) {
  __shared__ __attribute__((aligned(16))) float lds[64 * 4 + 64];
  const int lane = threadIdx.x;
  for (int i = lane; i < 64 * 4 + 64; i += 64) lds[i] = -7.0f;   // poison
  __syncthreads();
  const uint64_t b = (uint64_t)(uintptr_t)src;
  i32x4 srd;
  srd[0] = (int)(unsigned)b;
  srd[1] = (int)(unsigned)((b >> 32) & 0xffff);   // stride 0
  srd[2] = (int)nbytes;
  srd[3] = 0x00020000;
  srd[0] = __builtin_amdgcn_readfirstlane(srd[0]);
  srd[1] = __builtin_amdgcn_readfirstlane(srd[1]);
  srd[2] = __builtin_amdgcn_readfirstlane(srd[2]);
  srd[3] = __builtin_amdgcn_readfirstlane(srd[3]);
  const unsigned ldsoff = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_float*)(lds + 16));   // destination: lds + 16 floats
  const unsigned vo = off[lane];
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "v"(vo), "s"(srd), "s"(ldsoff) : "memory", "m0");
  __syncthreads();
  for (int i = lane; i < 64 * 4 + 64; i += 64) out[i] = lds[i];
}

int main() {
  const int n = 1024;   // floats in the buffer
  std::vector<float> h(n);
  for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
  std::vector<unsigned> off(64);
  for (int l = 0; l < 64; ++l) off[l] = 16u * ((l * 7) % 256);                     // in range, permuted
  off[3] = 4096u; off[5] = 4096u + 16u; off[9] = 0xfffffff0u; off[17] = 0x80000000u;   // at the end, past it, "negative", huge
  float *src, *out; unsigned* doff;
  (void)hipMalloc(&src, n * 4 + 4096); (void)hipMalloc(&out, (64 * 4 + 64) * 4); (void)hipMalloc(&doff, 64 * 4);
  (void)hipMemset(src, 0x7f, n * 4 + 4096);   // bytes behind the resource's range: must never show up
  (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice); (void)hipMemcpy(doff, off.data(), 64 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, src, (unsigned)(n * 4), doff, out);
  std::vector<float> r(64 * 4 + 64);
  (void)hipMemcpy(r.data(), out, r.size() * 4, hipMemcpyDeviceToHost);
  int bad = 0;
  for (int i = 0; i < 16; ++i) if (r[i] != -7.0f) { printf("front guard %d touched: %g\n", i, r[i]); ++bad; }
  for (int i = 16 + 256; i < 64 * 4 + 64; ++i) if (r[i] != -7.0f) { printf("tail guard %d touched: %g\n", i, r[i]); ++bad; }
  for (int l = 0; l < 64; ++l)
    for (int c = 0; c < 4; ++c) {
      const bool oob = off[l] >= (unsigned)(n * 4);
      const float want = oob ? 0.0f : h[off[l] / 4 + c];
      const float got = r[16 + l * 4 + c];
      if (got != want) { if (bad < 20) printf("lane %d comp %d: got %g want %g (offset %u%s)\n", l, c, got, want, off[l], oob ? ", out of range" : ""); ++bad; }
    }
  printf(bad ? "FAILED: %d mismatches\n" : "ok: lane l's 16 bytes land at M0 + 16 l, out-of-range lanes leave zeros (%d)\n", bad);
  return bad != 0;
}
