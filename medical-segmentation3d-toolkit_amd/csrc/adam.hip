// adam.hip -- fused Adam step over one flat fp32 parameter buffer, plus the small flat-buffer helpers the
// data-parallel train step needs (scale, zero).
//
// Reference: optim.Adam(net.parameters(), lr, betas) + opt.step() (core/seg_train.py:83,127); defaults eps 1e-8,
// weight_decay 0, amsgrad off.  Update rule restated from torch.optim.Adam (single-tensor path):
//   g += wd * p;  m = lerp(m, g, 1-b1);  v = b2 v + (1-b2) g g;
//   p -= (lr / (1 - b1^t)) * m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// HBM-bound: 4 reads + 3 writes of 4 B per parameter (28 B/param; 0.41 GB for the 14.56 M-parameter V-Net).
#include "seg3d_common.h"
#include "seg3d_hip.h"

__device__ __forceinline__ void adam_step_body(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                               float* __restrict__ v, i64 n, float lr, float beta1, float beta2, float eps,
                                               float weight_decay, float bc1, float bc2_sqrt, float grad_scale) {
  const float step_size = lr / bc1;
  const i64 n4 = n >> 2;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
    float4 pv = reinterpret_cast<float4*>(p)[i];
    float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float* pp = &pv.x; float* gp = &gv.x; float* mp = &mv.x; float* vp = &vv.x;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float gg = gp[k] * grad_scale;
      if (weight_decay != 0.f) gg = fmaf(weight_decay, pp[k], gg);
      mp[k] = mp[k] + (gg - mp[k]) * (1.0f - beta1);
      vp[k] = vp[k] * beta2 + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(vp[k]) / bc2_sqrt + eps;
      pp[k] = pp[k] - step_size * (mp[k] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = pv;
    reinterpret_cast<float4*>(m)[i] = mv;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  // tail
  for (i64 i = (n4 << 2) + (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    float gg = g[i] * grad_scale;
    if (weight_decay != 0.f) gg = fmaf(weight_decay, p[i], gg);
    const float mm = m[i] + (gg - m[i]) * (1.0f - beta1);
    const float vv = v[i] * beta2 + (1.0f - beta2) * gg * gg;
    m[i] = mm;
    v[i] = vv;
    p[i] = p[i] - step_size * (mm / (sqrtf(vv) / bc2_sqrt + eps));
  }
}

__global__ __launch_bounds__(256) void adam_step_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                          float* __restrict__ m, float* __restrict__ v, i64 n, float lr,
                                                          float beta1, float beta2, float eps, float weight_decay,
                                                          float bc1, float bc2_sqrt, float grad_scale) {
  adam_step_body(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2_sqrt, grad_scale);
}

// Device-resident step counter (a train step captured in a hipGraph cannot take the step number as a launch argument):
// one thread advances *step and writes the two bias corrections, the update kernel reads them from memory.
__global__ void adam_advance_kernel(int* __restrict__ step, float* __restrict__ bc, float beta1, float beta2) {
  const int t = *step + 1;
  *step = t;
  bc[0] = (float)(1.0 - pow((double)beta1, (double)t));
  bc[1] = (float)sqrt(1.0 - pow((double)beta2, (double)t));
}

__global__ __launch_bounds__(256) void adam_step_devstep_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                                  float* __restrict__ m, float* __restrict__ v, i64 n,
                                                                  float lr, float beta1, float beta2, float eps,
                                                                  float weight_decay, const float* __restrict__ bc,
                                                                  float grad_scale) {
  adam_step_body(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bc[0], bc[1], grad_scale);
}

// step_dev: device int holding the number of steps taken so far (advanced here); bc_dev: 2 floats of device scratch.
// Same update as seg3d_adam_step with step = *step_dev + 1.
extern "C" int seg3d_adam_step_devstep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                                       int* step_dev, float* bc_dev, float lr, float beta1, float beta2, float eps,
                                       float weight_decay, float grad_scale, void* stream) {
  SEG3D_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step_dev && bc_dev,
                "seg3d_adam_step_devstep: bad arguments");
  SEG3D_REQUIRE(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                    ((uintptr_t)exp_avg_sq % 16) == 0,
                "seg3d_adam_step_devstep: buffers must be 16-byte aligned");
  hipLaunchKernelGGL(adam_advance_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, step_dev, bc_dev, beta1, beta2);
  hipLaunchKernelGGL(adam_step_devstep_kernel, dim3(seg3d_ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream,
                     params, grads, exp_avg, exp_avg_sq, (i64)n, lr, beta1, beta2, eps, weight_decay, bc_dev, grad_scale);
  SEG3D_LAUNCH_CHECK("seg3d_adam_step_devstep");
  return SEG3D_OK;
}

// step >= 1 is the 1-based step count AFTER increment (torch increments before use).
// grad_scale multiplies the gradient first (1/world_size after a sum all-reduce; 1.0 otherwise).
extern "C" int seg3d_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step,
                               float lr, float beta1, float beta2, float eps, float weight_decay, float grad_scale,
                               void* stream) {
  SEG3D_REQUIRE(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "seg3d_adam_step: bad arguments");
  SEG3D_REQUIRE(((uintptr_t)params % 16) == 0 && ((uintptr_t)grads % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                    ((uintptr_t)exp_avg_sq % 16) == 0,
                "seg3d_adam_step: buffers must be 16-byte aligned");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  hipLaunchKernelGGL(adam_step_kernel, dim3(seg3d_ew_grid(n / 4 + 1, 256)), dim3(256), 0, (hipStream_t)stream, params, grads,
                     exp_avg, exp_avg_sq, (i64)n, lr, beta1, beta2, eps, weight_decay, (float)bc1, (float)sqrt(bc2),
                     grad_scale);
  SEG3D_LAUNCH_CHECK("seg3d_adam_step");
  return SEG3D_OK;
}
