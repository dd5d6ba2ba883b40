#include <hip/hip_runtime.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int DMA>
__global__ __launch_bounds__(256) void k(const float* g, float* out) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x;
  f32x16 acc = {0};
  if (DMA == 1) __builtin_amdgcn_global_load_lds(g + lane * 4, lds + 8192, 16, 0, 0);
  if (DMA == 2) { lds[lane + 9000] = g[lane]; }
  float a0 = lds[lane], a1 = lds[lane + 64];
#pragma unroll
  for (int s = 0; s < 16; ++s) {
    float a2 = lds[lane + 64 * (s + 2)];
    __builtin_amdgcn_sched_barrier(0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, a0, acc, 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
    a0 = a1; a1 = a2;
  }
  __syncthreads();
  for (int i = 0; i < 16; ++i) out[lane * 16 + i] = acc[i] + a0 + a1;
}
template __global__ void k<0>(const float*, float*);
template __global__ void k<1>(const float*, float*);
template __global__ void k<2>(const float*, float*);
