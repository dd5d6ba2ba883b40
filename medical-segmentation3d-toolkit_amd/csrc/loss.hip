// loss.hip -- channel softmax (network head), multi-class Dice loss, Focal loss; forward + backward.
//
// Reference semantics restated (file:line relative to /root/reference/segmentation3d):
//   * nn.Softmax(dim=1) at the end of OutputBlock                     network/module/vnet_outblock.py:18,23
//   * MultiDiceLoss = sum_c w_c * BinaryDice(cat([1/C, p_c]), target == c)   loss/multi_dice_loss.py:31-41
//     BinaryDiceLoss: (v, idx) = max over the 2 channels, v *= idx  ==>  phat = p * [p > 1/C] (ties -> 0);
//     per sample 1 - (2 sum(phat t) + 1e-6) / (sum(phat^2) + sum(t^2) + 1e-6), mean over batch
//                                                                      loss/binary_dice_loss.py:13-34
//   * FocalLoss: p_t = p[target] + 1e-10; -alpha_t (1 - p_t)^gamma log(p_t); mean (or sum) over voxels
//                                                                      loss/focal_loss.py:44-59
// Probabilities are NCDHW planar (the plugin API's output layout), targets are float class ids [N][1][S].
// All kernels are HBM-bound single passes; spatial sums use wave64 shuffles + one slot per workgroup and an
// fp64 finalize in fixed order.
#include "seg3d_common.h"
#include "seg3d_hip.h"

#define SEG3D_MAXC 16

// ---- softmax over channels: in NDHWC [V][C] -> out NCDHW [N][C][S] ----------------------------------------------
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ in, float* __restrict__ out, int C,
                                                            i64 S, i64 total_vox) {
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total_vox; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    float e[SEG3D_MAXC];
    float mx = -INFINITY;
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) {
        e[c] = in[v * C + c];
        mx = fmaxf(mx, e[c]);
      }
    float sum = 0.f;
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) {
        e[c] = expf(e[c] - mx);
        sum += e[c];
      }
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) out[(n * C + c) * S + s] = e[c] / sum;
  }
}

// din[v][c] = p_c (dp_c - sum_k p_k dp_k);  probs, dprobs NCDHW planar, din NDHWC
__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ probs,
                                                            const float* __restrict__ dprobs, float* __restrict__ din,
                                                            int C, i64 S, i64 total_vox) {
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total_vox; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    float p[SEG3D_MAXC], d[SEG3D_MAXC];
    float dot = 0.f;
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) {
        p[c] = probs[(n * C + c) * S + s];
        d[c] = dprobs[(n * C + c) * S + s];
        dot += p[c] * d[c];
      }
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) din[v * C + c] = p[c] * (d[c] - dot);
  }
}

extern "C" int seg3d_softmax_fwd(const float* in_ndhwc, float* probs_ncdhw, int N, int C, long long S, void* stream) {
  SEG3D_REQUIRE(in_ndhwc && probs_ncdhw && N > 0 && S > 0, "seg3d_softmax_fwd: bad arguments");
  SEG3D_REQUIRE(C >= 1 && C <= SEG3D_MAXC, "seg3d_softmax_fwd: num_classes %d not in [1, %d]", C, SEG3D_MAXC);
  const i64 total = (i64)N * S;
  hipLaunchKernelGGL(softmax_fwd_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in_ndhwc,
                     probs_ncdhw, C, (i64)S, total);
  SEG3D_LAUNCH_CHECK("seg3d_softmax_fwd");
  return SEG3D_OK;
}

extern "C" int seg3d_softmax_bwd(const float* probs_ncdhw, const float* dprobs_ncdhw, float* din_ndhwc, int N, int C,
                                 long long S, void* stream) {
  SEG3D_REQUIRE(probs_ncdhw && dprobs_ncdhw && din_ndhwc && N > 0 && S > 0, "seg3d_softmax_bwd: bad arguments");
  SEG3D_REQUIRE(C >= 1 && C <= SEG3D_MAXC, "seg3d_softmax_bwd: num_classes %d not in [1, %d]", C, SEG3D_MAXC);
  const i64 total = (i64)N * S;
  hipLaunchKernelGGL(softmax_bwd_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, probs_ncdhw,
                     dprobs_ncdhw, din_ndhwc, C, (i64)S, total);
  SEG3D_LAUNCH_CHECK("seg3d_softmax_bwd");
  return SEG3D_OK;
}

// ---- Dice ---------------------------------------------------------------------------------------------------------
#define DICE_VPB 4096  // voxels per workgroup

// part[n][blk][c][3] = (sum phat*t, sum phat^2, sum t) over the block's voxels
__global__ __launch_bounds__(256) void dice_partial_kernel(const float* __restrict__ probs,
                                                             const float* __restrict__ target, float* __restrict__ part,
                                                             int C, i64 S, int nblk, float thresh) {
  __shared__ float red[12];
  const int n = blockIdx.y;
  const i64 s0 = (i64)blockIdx.x * DICE_VPB;
  i64 s1 = s0 + DICE_VPB;
  if (s1 > S) s1 = S;
  float I[SEG3D_MAXC], P2[SEG3D_MAXC], T[SEG3D_MAXC];
#pragma unroll
  for (int c = 0; c < SEG3D_MAXC; ++c) { I[c] = 0.f; P2[c] = 0.f; T[c] = 0.f; }
  for (i64 s = s0 + threadIdx.x; s < s1; s += 256) {
    const float t = target[(i64)n * S + s];
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) {
        const float p = probs[((i64)n * C + c) * S + s];
        const float ph = p > thresh ? p : 0.f;
        const float tc = (t == (float)c) ? 1.f : 0.f;
        I[c] += ph * tc;
        P2[c] += ph * ph;
        T[c] += tc;
      }
  }
#pragma unroll
  for (int c = 0; c < SEG3D_MAXC; ++c)
    if (c < C) {
      float v[3] = {I[c], P2[c], T[c]};
      block_sum_256<3>(v, red);
      if (threadIdx.x == 0) {
        float* dst = part + (((i64)n * nblk + blockIdx.x) * C + c) * 3;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
      }
    }
}

// sums[n][c] = (I, P2 + T) in fp32 for the backward; loss = sum_c w_c mean_n (1 - (2I + eps)/(P2 + T + eps))
// One workgroup; 32 lanes per (n, c) stride over the partial blocks (fp64, then a fixed-shape shuffle tree: reproducible).
// The first form walked the nblk blocks of a (n, c) with ONE thread -- 8 threads busy, 34 us on the loss's critical path.
__global__ __launch_bounds__(256) void dice_finalize_kernel(const float* __restrict__ part,
                                                              const float* __restrict__ weights, float* __restrict__ sums,
                                                              float* __restrict__ loss, int N, int C, int nblk) {
  __shared__ double red[8];
  const int grp = threadIdx.x >> 5, l32 = threadIdx.x & 31;
  double acc = 0.0;
  for (int idx0 = 0; idx0 < N * C; idx0 += 8) {   // uniform trip count: the shuffles below need whole waves
    const int idx = idx0 + grp;
    const bool ok = idx < N * C;
    const int n = ok ? idx / C : 0, c = ok ? idx % C : 0;
    double I = 0.0, P2 = 0.0, T = 0.0;
    if (ok) {
      const float* p = part + ((i64)n * nblk * C + c) * 3;
      for (int k = l32; k < nblk; k += 32) {
        I += (double)p[(i64)k * C * 3 + 0];
        P2 += (double)p[(i64)k * C * 3 + 1];
        T += (double)p[(i64)k * C * 3 + 2];
      }
    }
#pragma unroll
    for (int off = 16; off > 0; off >>= 1) {   // within the 32-lane group (width 32: never crosses into the other group)
      I += __shfl_down(I, off, 32);
      P2 += __shfl_down(P2, off, 32);
      T += __shfl_down(T, off, 32);
    }
    if (ok && l32 == 0) {
      const float If = (float)I, sumf = (float)(P2 + T);
      sums[2 * idx + 0] = If;
      sums[2 * idx + 1] = sumf;
      const float eps = 1e-6f;
      const float l = 1.0f - (2.0f * If + eps) / (sumf + eps);
      acc += (double)weights[c] * (double)l / (double)N;
    }
  }
  if (l32 == 0) red[grp] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int g = 0; g < 8; ++g) t += red[g];
    loss[0] = (float)t;
  }
}

// dprobs[n][c][s] = gout * w_c / N * [p > thresh] * -(2 t (sum + eps) - (2 I + eps) 2 p) / (sum + eps)^2
__global__ __launch_bounds__(256) void dice_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ target,
                                                         const float* __restrict__ sums, const float* __restrict__ weights,
                                                         const float* __restrict__ gout, float* __restrict__ dprobs, int N,
                                                         int C, i64 S, float thresh) {
  const i64 total = (i64)N * S;
  const float go = gout[0];
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    const float t = target[v];
#pragma unroll
    for (int c = 0; c < SEG3D_MAXC; ++c)
      if (c < C) {
        const i64 off = (n * C + c) * S + s;
        const float p = probs[off];
        float d = 0.f;
        if (p > thresh) {
          const float eps = 1e-6f;
          const float I = sums[2 * (n * C + c) + 0], den = sums[2 * (n * C + c) + 1] + eps;
          const float tc = (t == (float)c) ? 1.f : 0.f;
          const float num = 2.0f * I + eps;
          d = -(2.0f * tc * den - num * 2.0f * p) / (den * den);
          d *= go * weights[c] / (float)N;
        }
        dprobs[off] = d;
      }
  }
}

extern "C" long long seg3d_dice_blocks(long long S) { return (S + DICE_VPB - 1) / DICE_VPB; }

// workspace part: [N][seg3d_dice_blocks(S)][C][3]; sums: [N][C][2] (kept for backward); loss: 1 float
extern "C" int seg3d_dice_fwd(const float* probs, const float* target, const float* weights, float* part, float* sums,
                              float* loss, int N, int C, long long S, void* stream) {
  SEG3D_REQUIRE(probs && target && weights && part && sums && loss && N > 0 && S > 0, "seg3d_dice_fwd: bad arguments");
  SEG3D_REQUIRE(C >= 1 && C <= SEG3D_MAXC, "seg3d_dice_fwd: num_class %d not in [1, %d]", C, SEG3D_MAXC);
  const int nblk = (int)seg3d_dice_blocks(S);
  const float thresh = (float)(1.0 / (double)C);  // 1.0 / num_class added to float32 zeros (multi_dice_loss.py:36)
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(dice_partial_kernel, dim3(nblk, N), dim3(256), 0, s, probs, target, part, C, (i64)S, nblk, thresh);
  SEG3D_LAUNCH_CHECK("seg3d_dice_fwd(partial)");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, part, weights, sums, loss, N, C, nblk);
  SEG3D_LAUNCH_CHECK("seg3d_dice_fwd(finalize)");
  return SEG3D_OK;
}

extern "C" int seg3d_dice_bwd(const float* probs, const float* target, const float* sums, const float* weights,
                              const float* gout, float* dprobs, int N, int C, long long S, void* stream) {
  SEG3D_REQUIRE(probs && target && sums && weights && gout && dprobs && N > 0 && S > 0, "seg3d_dice_bwd: bad arguments");
  SEG3D_REQUIRE(C >= 1 && C <= SEG3D_MAXC, "seg3d_dice_bwd: num_class %d not in [1, %d]", C, SEG3D_MAXC);
  const float thresh = (float)(1.0 / (double)C);
  hipLaunchKernelGGL(dice_bwd_kernel, dim3(seg3d_ew_grid((i64)N * S, 256)), dim3(256), 0, (hipStream_t)stream, probs, target,
                     sums, weights, gout, dprobs, N, C, (i64)S, thresh);
  SEG3D_LAUNCH_CHECK("seg3d_dice_bwd");
  return SEG3D_OK;
}

// ---- BinaryDiceLoss on its own (loss/binary_dice_loss.py:9-36) ------------------------------------------------------
// probs [N][2][S]; (value, index) = max over the two channels, value *= index  ==>  pred = p1 where p1 > p0 STRICTLY (a tie
// takes index 0), else 0; target is used as a float (t * t in the area).  part[n][blk][1][3] = (sum pred t, sum pred^2,
// sum t^2): the layout of dice_partial_kernel with one class, so dice_finalize_kernel (C = 1, weight 1) finishes it.
__global__ __launch_bounds__(256) void bdice_partial_kernel(const float* __restrict__ probs, const float* __restrict__ target,
                                                              float* __restrict__ part, i64 S, int nblk) {
  __shared__ float red[12];
  const int n = blockIdx.y;
  const i64 s0 = (i64)blockIdx.x * DICE_VPB;
  i64 s1 = s0 + DICE_VPB;
  if (s1 > S) s1 = S;
  float v[3] = {0.f, 0.f, 0.f};
  for (i64 s = s0 + threadIdx.x; s < s1; s += 256) {
    const float t = target[(i64)n * S + s];
    const float p0 = probs[((i64)n * 2 + 0) * S + s], p1 = probs[((i64)n * 2 + 1) * S + s];
    const float ph = p1 > p0 ? p1 : 0.f;
    v[0] += ph * t;
    v[1] += ph * ph;
    v[2] += t * t;
  }
  block_sum_256<3>(v, red);
  if (threadIdx.x == 0) {
    float* dst = part + ((i64)n * nblk + blockIdx.x) * 3;
    dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
  }
}

// dprobs[n][1][s] = gout / N * [p1 > p0] * -(2 t (sum + eps) - (2 I + eps) 2 p1) / (sum + eps)^2;  dprobs[n][0][s] = 0
__global__ __launch_bounds__(256) void bdice_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ target,
                                                          const float* __restrict__ sums, const float* __restrict__ gout,
                                                          float* __restrict__ dprobs, int N, i64 S) {
  const i64 total = (i64)N * S;
  const float go = gout[0];
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    const float t = target[v];
    const float p0 = probs[(n * 2 + 0) * S + s], p1 = probs[(n * 2 + 1) * S + s];
    float d = 0.f;
    if (p1 > p0) {
      const float eps = 1e-6f;
      const float I = sums[2 * n + 0], den = sums[2 * n + 1] + eps;
      const float num = 2.0f * I + eps;
      d = -(2.0f * t * den - num * 2.0f * p1) / (den * den) * (go / (float)N);
    }
    dprobs[(n * 2 + 0) * S + s] = 0.f;
    dprobs[(n * 2 + 1) * S + s] = d;
  }
}

// workspace part: [N][seg3d_dice_blocks(S)][3]; sums: [N][2] (kept for backward); one: device float holding 1.0f; loss: 1 float
extern "C" int seg3d_binary_dice_fwd(const float* probs, const float* target, const float* one, float* part, float* sums,
                                     float* loss, int N, long long S, void* stream) {
  SEG3D_REQUIRE(probs && target && one && part && sums && loss && N > 0 && S > 0, "seg3d_binary_dice_fwd: bad arguments");
  const int nblk = (int)seg3d_dice_blocks(S);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(bdice_partial_kernel, dim3(nblk, N), dim3(256), 0, s, probs, target, part, (i64)S, nblk);
  SEG3D_LAUNCH_CHECK("seg3d_binary_dice_fwd(partial)");
  hipLaunchKernelGGL(dice_finalize_kernel, dim3(1), dim3(256), 0, s, part, one, sums, loss, N, 1, nblk);
  SEG3D_LAUNCH_CHECK("seg3d_binary_dice_fwd(finalize)");
  return SEG3D_OK;
}

extern "C" int seg3d_binary_dice_bwd(const float* probs, const float* target, const float* sums, const float* gout,
                                     float* dprobs, int N, long long S, void* stream) {
  SEG3D_REQUIRE(probs && target && sums && gout && dprobs && N > 0 && S > 0, "seg3d_binary_dice_bwd: bad arguments");
  hipLaunchKernelGGL(bdice_bwd_kernel, dim3(seg3d_ew_grid((i64)N * S, 256)), dim3(256), 0, (hipStream_t)stream, probs, target,
                     sums, gout, dprobs, N, (i64)S);
  SEG3D_LAUNCH_CHECK("seg3d_binary_dice_bwd");
  return SEG3D_OK;
}

// ---- Focal ----------------------------------------------------------------------------------------------------------
// probs element (n, c, s) lives at n*sn + c*sc + s*ss (planar NCDHW: sn = C*S, sc = S, ss = 1;
// [sample, class] matrices: sn = 0, sc = 1, ss = C)
__device__ __forceinline__ float focal_pow(float q, float gamma) {
  if (gamma == 2.0f) return q * q;
  if (gamma == 1.0f) return q;
  return powf(q, gamma);
}

__global__ __launch_bounds__(256) void focal_partial_kernel(const float* __restrict__ probs,
                                                              const float* __restrict__ target,
                                                              const float* __restrict__ alpha, float* __restrict__ part,
                                                              int C, i64 S, i64 total, i64 sn, i64 sc, i64 ss, float gamma) {
  __shared__ float red[4];
  float acc[1] = {0.f};
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    const int t = (int)(long long)target[v];
    if (t >= 0 && t < C) {
      const float pt = probs[n * sn + t * sc + s * ss] + 1e-10f;
      const float lp = logf(pt);
      float l = -alpha[t] * lp;
      if (gamma > 0.f) l *= focal_pow(1.0f - pt, gamma);
      acc[0] += l;
    }
  }
  block_sum_256<1>(acc, red);
  if (threadIdx.x == 0) part[blockIdx.x] = acc[0];
}

__global__ __launch_bounds__(256) void focal_finalize_kernel(const float* __restrict__ part, float* __restrict__ loss,
                                                               int nblk, double scale) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 256) acc += (double)part[k];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) * scale);
}

// d/dp_t [-alpha q^gamma log p] = alpha (gamma q^(gamma-1) log p - q^gamma / p),  q = 1 - p
__global__ __launch_bounds__(256) void focal_bwd_kernel(const float* __restrict__ probs, const float* __restrict__ target,
                                                          const float* __restrict__ alpha, const float* __restrict__ gout,
                                                          float* __restrict__ dprobs, int C, i64 S, i64 total, i64 sn,
                                                          i64 sc, i64 ss, float gamma, float scale) {
  const float go = gout[0] * scale;
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total; v += (i64)gridDim.x * 256) {
    const i64 n = v / S, s = v - n * S;
    const int t = (int)(long long)target[v];
    float dt = 0.f;
    if (t >= 0 && t < C) {
      const float pt = probs[n * sn + t * sc + s * ss] + 1e-10f;
      const float q = 1.0f - pt;
      const float lp = logf(pt);
      if (gamma > 0.f) {
        const float qg1 = (gamma == 2.0f) ? q : (gamma == 1.0f ? 1.0f : powf(q, gamma - 1.0f));
        dt = alpha[t] * (gamma * qg1 * lp - focal_pow(q, gamma) / pt);
      } else {
        dt = -alpha[t] / pt;
      }
      dt *= go;
    }
    for (int c = 0; c < C; ++c) dprobs[n * sn + c * sc + s * ss] = (c == t) ? dt : 0.f;
  }
}

extern "C" long long seg3d_focal_blocks(long long total_vox) { return seg3d_ew_grid(total_vox, 256 * 8); }

// part: [seg3d_focal_blocks(N*S)] floats; loss: 1 float. size_average != 0 -> mean over N*S voxels, else sum.
extern "C" int seg3d_focal_fwd(const float* probs, const float* target, const float* alpha, float* part, float* loss, int N,
                               int C, long long S, long long sn, long long sc, long long ss, float gamma, int size_average,
                               void* stream) {
  SEG3D_REQUIRE(probs && target && alpha && part && loss && N > 0 && S > 0 && C > 0, "seg3d_focal_fwd: bad arguments");
  const i64 total = (i64)N * S;
  const int nblk = (int)seg3d_focal_blocks(total);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(focal_partial_kernel, dim3(nblk), dim3(256), 0, s, probs, target, alpha, part, C, (i64)S, total, (i64)sn,
                     (i64)sc, (i64)ss, gamma);
  SEG3D_LAUNCH_CHECK("seg3d_focal_fwd(partial)");
  hipLaunchKernelGGL(focal_finalize_kernel, dim3(1), dim3(256), 0, s, part, loss, nblk,
                     size_average ? 1.0 / (double)total : 1.0);
  SEG3D_LAUNCH_CHECK("seg3d_focal_fwd(finalize)");
  return SEG3D_OK;
}

extern "C" int seg3d_focal_bwd(const float* probs, const float* target, const float* alpha, const float* gout, float* dprobs,
                               int N, int C, long long S, long long sn, long long sc, long long ss, float gamma,
                               int size_average, void* stream) {
  SEG3D_REQUIRE(probs && target && alpha && gout && dprobs && N > 0 && S > 0 && C > 0, "seg3d_focal_bwd: bad arguments");
  const i64 total = (i64)N * S;
  hipLaunchKernelGGL(focal_bwd_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, probs, target,
                     alpha, gout, dprobs, C, (i64)S, total, (i64)sn, (i64)sc, (i64)ss, gamma,
                     size_average ? (float)(1.0 / (double)total) : 1.0f);
  SEG3D_LAUNCH_CHECK("seg3d_focal_bwd");
  return SEG3D_OK;
}
