// What does a 16-byte-per-lane global load (or store) cost when the lanes of an instruction do not cover contiguous memory?
// The stride-2 convolution kernels read / write one 16-byte piece of a voxel row per lane, rows 64-256 bytes apart: every lane
// of an instruction in its own cache line.  This bench streams a 1-GiB buffer with 64-lane dwordx4 instructions whose lane l
// touches bytes [l * S, l * S + 16) of a 64 * S-byte window, the 16-byte column c = 0 .. S/16 - 1 of the window visited by S/16
// consecutive instructions (so every byte is read exactly once, like the K chunks of those kernels), for S = 16 (contiguous),
// 32, 64, 128, 256; and the same with PAIRS (lanes 2p, 2p + 1 take 32 contiguous bytes at stride S).
// build: hipcc -O3 --offload-arch=gfx950 tools/ubench/stride_load.hip -o tools/ubench/stride_load
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE 0: load, 1: store.  W = contiguous bytes per lane group (16: single lanes, 32: pairs, 64: quads)
template <int MODE, int W>
__global__ __launch_bounds__(256) void k(float* buf, long long nwin, int S, float* sink) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long long)gridDim.x * 4;
  constexpr int LPG = W / 16;                 // lanes per group
  const int grp = lane / LPG, sub = lane % LPG;
  const int ngroups = 64 / LPG;               // rows per window
  const int cols = S / W;                     // instructions per window
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  for (long long w = wave; w < nwin; w += nwaves) {
    char* base = reinterpret_cast<char*>(buf) + w * (long long)ngroups * S;
    for (int c0 = 0; c0 < cols; c0 += 4) {
      f32x4 v[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int c = c0 + u < cols ? c0 + u : cols - 1;
        f32x4* p = reinterpret_cast<f32x4*>(base + (long long)grp * S + c * W + sub * 16);
        if (MODE == 0) v[u] = *p; else if (c0 + u < cols) *p = f32x4{1.f, 2.f, 3.f, (float)c};
      }
      if (MODE == 0) {
#pragma unroll
        for (int u = 0; u < 4; ++u) if (c0 + u < cols) acc += v[u];
      }
    }
  }
  if (MODE == 0 && acc[0] + acc[1] + acc[2] + acc[3] == 123.456f) sink[0] = acc[0];
}

template <int MODE, int W>
static void run(float* buf, size_t bytes, int S, float* sink) {
  if (S < W) return;
  const long long nwin = (long long)(bytes / ((size_t)(64 / (W / 16)) * S));
  hipEvent_t a, b;
  hipEventCreate(&a); hipEventCreate(&b);
  for (int it = 0; it < 2; ++it) hipLaunchKernelGGL((k<MODE, W>), dim3(256 * 8), dim3(256), 0, 0, buf, nwin, S, sink);
  hipEventRecord(a);
  const int iters = 5;
  for (int it = 0; it < iters; ++it) hipLaunchKernelGGL((k<MODE, W>), dim3(256 * 8), dim3(256), 0, 0, buf, nwin, S, sink);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0.f;
  hipEventElapsedTime(&ms, a, b);
  ms /= iters;
  printf("%s  %2d contiguous bytes per lane group, stride %3d B: %7.3f ms  %6.0f GB/s\n", MODE ? "store" : "load ", W, S, ms, bytes / (ms * 1e-3) / 1e9);
}

int main() {
  const size_t bytes = 1ull << 30;
  float *buf, *sink;
  hipMalloc(&buf, bytes);
  hipMalloc(&sink, 64);
  hipMemset(buf, 0, bytes);
  const int strides[] = {16, 32, 64, 128, 256, 512};
  for (int S : strides) { run<0, 16>(buf, bytes, S, sink); }
  for (int S : strides) { run<0, 32>(buf, bytes, S, sink); }
  for (int S : strides) { run<0, 64>(buf, bytes, S, sink); }
  for (int S : strides) { run<1, 16>(buf, bytes, S, sink); }
  for (int S : strides) { run<1, 32>(buf, bytes, S, sink); }
  for (int S : strides) { run<1, 64>(buf, bytes, S, sink); }
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
