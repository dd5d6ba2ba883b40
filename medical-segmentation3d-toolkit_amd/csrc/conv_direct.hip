// conv_direct.hip -- VALU ("direct") 3D convolution kernels in NDHWC.
//
// Covers the shapes that are HBM-bound or too thin for the matrix cores:
//   * stem  Conv3d k3 p1, Cin 1..4 -> 16                  (network/module/vnet_inblock.py:9)
//   * head  Conv3d k3 p1, 32 -> num_classes and k1         (network/module/vnet_outblock.py:13,16)
//   * Conv3d k2 s2 (DownBlock, vnet_downblock.py:11) and ConvTranspose3d k2 s2 (UpBlock, vnet_upblock.py:11)
// and acts as the generic fallback / cross-check for the MFMA path in conv_mfma.hip.
//
// Thread mapping everywhere: one thread = (output voxel, quad of 4 output channels), quads fastest, so a
// wave-instruction stores 64 x 16 B = 1 KiB contiguous when Cout >= 256 and full voxel rows otherwise.
// Weights are "tap-major": wp[tap][a][bp] (layout.hip), bp padded to a multiple of 4.
#include "seg3d_common.h"
#include "seg3d_hip.h"

// ---- gather convolution: y[v][co] = bias[co] + sum_{tap,ci} x[v*STRIDE + tap - PAD][ci] * wp[tap][ci][co] --------
template <int KS, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void conv_fwd_direct_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                const float* __restrict__ bias, float* __restrict__ y,
                                                                int N, int Di, int Hi, int Wi, int Do, int Ho, int Wo,
                                                                int Cin, int Cout, int CoutP) {
  const int CQ = CoutP >> 2;
  const i64 total = (i64)N * Do * Ho * Wo * CQ;
  const bool vec_ci = (Cin & 3) == 0;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int q = (int)(idx % CQ);
    i64 v = idx / CQ;
    const int xo = (int)(v % Wo);
    i64 t = v / Wo;
    const int yo = (int)(t % Ho);
    t /= Ho;
    const int zo = (int)(t % Do);
    const int n = (int)(t / Do);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
      const int c0 = 4 * q;
      acc.x = c0 + 0 < Cout ? bias[c0 + 0] : 0.f;
      acc.y = c0 + 1 < Cout ? bias[c0 + 1] : 0.f;
      acc.z = c0 + 2 < Cout ? bias[c0 + 2] : 0.f;
      acc.w = c0 + 3 < Cout ? bias[c0 + 3] : 0.f;
    }
#pragma unroll 1
    for (int kz = 0; kz < KS; ++kz) {
      const int zi = zo * STRIDE + kz - PAD;
      if (zi < 0 || zi >= Di) continue;
#pragma unroll 1
      for (int ky = 0; ky < KS; ++ky) {
        const int yi = yo * STRIDE + ky - PAD;
        if (yi < 0 || yi >= Hi) continue;
#pragma unroll
        for (int kx = 0; kx < KS; ++kx) {
          const int xi = xo * STRIDE + kx - PAD;
          if (xi < 0 || xi >= Wi) continue;
          const int tap = (kz * KS + ky) * KS + kx;
          const float* xp = x + ((((i64)n * Di + zi) * Hi + yi) * Wi + xi) * Cin;
          const float* wt = wp + (i64)tap * Cin * CoutP + 4 * q;
          if (vec_ci) {
            for (int ci = 0; ci < Cin; ci += 4) {
              const float4 xv = *reinterpret_cast<const float4*>(xp + ci);
              const float4 w0 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 0) * CoutP);
              const float4 w1 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 1) * CoutP);
              const float4 w2 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 2) * CoutP);
              const float4 w3 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 3) * CoutP);
              acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
              acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
              acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
              acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
            }
          } else {
            for (int ci = 0; ci < Cin; ++ci) {
              const float xv = xp[ci];
              const float4 w0 = *reinterpret_cast<const float4*>(wt + (i64)ci * CoutP);
              acc.x = fmaf(xv, w0.x, acc.x); acc.y = fmaf(xv, w0.y, acc.y); acc.z = fmaf(xv, w0.z, acc.z); acc.w = fmaf(xv, w0.w, acc.w);
            }
          }
        }
      }
    }
    float* yp = y + v * Cout + 4 * q;
    if ((Cout & 3) == 0) {
      *reinterpret_cast<float4*>(yp) = acc;
    } else {
      const int c0 = 4 * q;
      if (c0 + 0 < Cout) yp[0] = acc.x;
      if (c0 + 1 < Cout) yp[1] = acc.y;
      if (c0 + 2 < Cout) yp[2] = acc.z;
      if (c0 + 3 < Cout) yp[3] = acc.w;
    }
  }
}

// ---- transposed k2 s2: y[2i + tap][co] = bias[co] + sum_ci x[i][ci] * wp[tap][ci][co]  (disjoint 2^3 cells) ------
__global__ __launch_bounds__(256) void convT_k2s2_fwd_direct_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ wp,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ y, int N, int Di, int Hi,
                                                                      int Wi, int Cin, int Cout, int CoutP) {
  const int CQ = CoutP >> 2;
  const int Do = 2 * Di, Ho = 2 * Hi, Wo = 2 * Wi;
  const i64 total = (i64)N * Do * Ho * Wo * CQ;
  const bool vec_ci = (Cin & 3) == 0;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int q = (int)(idx % CQ);
    i64 v = idx / CQ;
    const int xo = (int)(v % Wo);
    i64 t = v / Wo;
    const int yo = (int)(t % Ho);
    t /= Ho;
    const int zo = (int)(t % Do);
    const int n = (int)(t / Do);
    const int tap = ((zo & 1) * 2 + (yo & 1)) * 2 + (xo & 1);
    const float* xp = x + ((((i64)n * Di + (zo >> 1)) * Hi + (yo >> 1)) * Wi + (xo >> 1)) * Cin;
    const float* wt = wp + (i64)tap * Cin * CoutP + 4 * q;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (bias) {
      const int c0 = 4 * q;
      acc.x = c0 + 0 < Cout ? bias[c0 + 0] : 0.f;
      acc.y = c0 + 1 < Cout ? bias[c0 + 1] : 0.f;
      acc.z = c0 + 2 < Cout ? bias[c0 + 2] : 0.f;
      acc.w = c0 + 3 < Cout ? bias[c0 + 3] : 0.f;
    }
    if (vec_ci) {
      for (int ci = 0; ci < Cin; ci += 4) {
        const float4 xv = *reinterpret_cast<const float4*>(xp + ci);
        const float4 w0 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 0) * CoutP);
        const float4 w1 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 1) * CoutP);
        const float4 w2 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 2) * CoutP);
        const float4 w3 = *reinterpret_cast<const float4*>(wt + (i64)(ci + 3) * CoutP);
        acc.x = fmaf(xv.x, w0.x, acc.x); acc.y = fmaf(xv.x, w0.y, acc.y); acc.z = fmaf(xv.x, w0.z, acc.z); acc.w = fmaf(xv.x, w0.w, acc.w);
        acc.x = fmaf(xv.y, w1.x, acc.x); acc.y = fmaf(xv.y, w1.y, acc.y); acc.z = fmaf(xv.y, w1.z, acc.z); acc.w = fmaf(xv.y, w1.w, acc.w);
        acc.x = fmaf(xv.z, w2.x, acc.x); acc.y = fmaf(xv.z, w2.y, acc.y); acc.z = fmaf(xv.z, w2.z, acc.z); acc.w = fmaf(xv.z, w2.w, acc.w);
        acc.x = fmaf(xv.w, w3.x, acc.x); acc.y = fmaf(xv.w, w3.y, acc.y); acc.z = fmaf(xv.w, w3.z, acc.z); acc.w = fmaf(xv.w, w3.w, acc.w);
      }
    } else {
      for (int ci = 0; ci < Cin; ++ci) {
        const float xv = xp[ci];
        const float4 w0 = *reinterpret_cast<const float4*>(wt + (i64)ci * CoutP);
        acc.x = fmaf(xv, w0.x, acc.x); acc.y = fmaf(xv, w0.y, acc.y); acc.z = fmaf(xv, w0.z, acc.z); acc.w = fmaf(xv, w0.w, acc.w);
      }
    }
    float* yp = y + v * Cout + 4 * q;
    if ((Cout & 3) == 0) {
      *reinterpret_cast<float4*>(yp) = acc;
    } else {
      const int c0 = 4 * q;
      if (c0 + 0 < Cout) yp[0] = acc.x;
      if (c0 + 1 < Cout) yp[1] = acc.y;
      if (c0 + 2 < Cout) yp[2] = acc.z;
      if (c0 + 3 < Cout) yp[3] = acc.w;
    }
  }
}

extern "C" int seg3d_conv3d_fwd_direct(const float* x, const float* wp, const float* bias, float* y, int N, int Di,
                                       int Hi, int Wi, int Cin, int Cout, int ksize, int stride, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_fwd_direct: null pointer");
  SEG3D_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_fwd_direct: bad dims");
  const int CoutP = seg3d_round_up(Cout, 4);
  hipStream_t s = (hipStream_t)stream;
  if (ksize == 3 && stride == 1) {
    i64 work = (i64)N * Di * Hi * Wi * (CoutP / 4);
    hipLaunchKernelGGL((conv_fwd_direct_kernel<3, 1, 1>), dim3(seg3d_ew_grid(work, 256)), dim3(256), 0, s, x, wp, bias, y,
                       N, Di, Hi, Wi, Di, Hi, Wi, Cin, Cout, CoutP);
  } else if (ksize == 2 && stride == 2) {
    SEG3D_REQUIRE((Di % 2) == 0 && (Hi % 2) == 0 && (Wi % 2) == 0, "seg3d_conv3d_fwd_direct: k2s2 needs even input dims (got %d %d %d)", Di, Hi, Wi);
    i64 work = (i64)N * (Di / 2) * (Hi / 2) * (Wi / 2) * (CoutP / 4);
    hipLaunchKernelGGL((conv_fwd_direct_kernel<2, 2, 0>), dim3(seg3d_ew_grid(work, 256)), dim3(256), 0, s, x, wp, bias, y,
                       N, Di, Hi, Wi, Di / 2, Hi / 2, Wi / 2, Cin, Cout, CoutP);
  } else if (ksize == 1 && stride == 1) {
    i64 work = (i64)N * Di * Hi * Wi * (CoutP / 4);
    hipLaunchKernelGGL((conv_fwd_direct_kernel<1, 1, 0>), dim3(seg3d_ew_grid(work, 256)), dim3(256), 0, s, x, wp, bias, y,
                       N, Di, Hi, Wi, Di, Hi, Wi, Cin, Cout, CoutP);
  } else {
    SEG3D_UNSUPPORTED("seg3d_conv3d_fwd_direct: unsupported ksize=%d stride=%d (reference uses k3s1p1, k2s2, k1)", ksize, stride);
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_fwd_direct");
  return SEG3D_OK;
}

extern "C" int seg3d_convT3d_k2s2_fwd_direct(const float* x, const float* wp, const float* bias, float* y, int N, int Di,
                                             int Hi, int Wi, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_convT3d_k2s2_fwd_direct: null pointer");
  SEG3D_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "seg3d_convT3d_k2s2_fwd_direct: bad dims");
  const int CoutP = seg3d_round_up(Cout, 4);
  i64 work = (i64)N * Di * Hi * Wi * 8 * (CoutP / 4);
  hipLaunchKernelGGL(convT_k2s2_fwd_direct_kernel, dim3(seg3d_ew_grid(work, 256)), dim3(256), 0, (hipStream_t)stream, x,
                     wp, bias, y, N, Di, Hi, Wi, Cin, Cout, CoutP);
  SEG3D_LAUNCH_CHECK("seg3d_convT3d_k2s2_fwd_direct");
  return SEG3D_OK;
}

// ---- weight gradient, direct form -------------------------------------------------------------------------------
// dW(t, a, b) = sum_v P[pos(v, t)][a] * Q[v][b]       (P has CA channels, Q has CB channels, v runs over Q's voxels)
//   k3  : pos = v + tap - 1 (zero outside), P = x,  Q = dy            (a = ci, b = co)
//   k2s2: pos = 2v + tap,                   P = x,  Q = dy            (a = ci, b = co)
//   convT: pos = 2v + tap,                  P = dy, Q = x             (a = co, b = ci)  -- same kernel, roles swapped
//   k1  : pos = v,                          P = x,  Q = dy
// Each workgroup owns a contiguous chunk of Q voxels and produces one partial slab part[chunk][t][a][bp];
// seg3d_wgrad_reduce sums the slabs in chunk order (bitwise reproducible) into the reference weight layout.
// Threads: pair p = (a, b-quad) -> p = tid % PAIRS, voxel lane vl = tid / PAIRS; the VL lanes are combined in LDS.
template <int KS, int STRIDE, int PAD>
__global__ __launch_bounds__(256) void wgrad_direct_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                             float* __restrict__ part, int N, int Dp, int Hp, int Wp_,
                                                             int Dq, int Hq, int Wq, int CA, int CB, int CBP,
                                                             int pairs_per_block, int chunk_vox) {
  constexpr int T = KS * KS * KS;
  __shared__ float4 red[256];
  const int BQ = CBP >> 2;
  const int total_pairs = CA * BQ;
  const int PAIRS = pairs_per_block;  // divides 256
  const int VL = 256 / PAIRS;
  const int pl = threadIdx.x % PAIRS, vl = threadIdx.x / PAIRS;
  const int pair = blockIdx.y * PAIRS + pl;
  const bool pair_ok = pair < total_pairs;
  const int a = pair_ok ? pair / BQ : 0;
  const int bq = pair_ok ? pair % BQ : 0;
  const i64 nvox = (i64)N * Dq * Hq * Wq;
  const i64 v0 = (i64)blockIdx.x * chunk_vox;
  i64 v1 = v0 + chunk_vox;
  if (v1 > nvox) v1 = nvox;

  float4 acc[T];
#pragma unroll
  for (int t = 0; t < T; ++t) acc[t] = make_float4(0.f, 0.f, 0.f, 0.f);

  if (pair_ok) {
    for (i64 v = v0 + vl; v < v1; v += VL) {
      int xq = 0, yq = 0, zq = 0, n = 0;
      if constexpr (KS != 1) {   // (a 1x1x1 tap reads P at the same voxel: no coordinates, no 64-bit divisions)
        xq = (int)(v % Wq);
        i64 r = v / Wq;
        yq = (int)(r % Hq);
        r /= Hq;
        zq = (int)(r % Dq);
        n = (int)(r / Dq);
      }
      float4 qv;
      {
        const float* qp = Q + v * CB + 4 * bq;
        if ((CB & 3) == 0) {
          qv = *reinterpret_cast<const float4*>(qp);
        } else {
          const int c0 = 4 * bq;
          qv.x = c0 + 0 < CB ? qp[0] : 0.f;
          qv.y = c0 + 1 < CB ? qp[1] : 0.f;
          qv.z = c0 + 2 < CB ? qp[2] : 0.f;
          qv.w = c0 + 3 < CB ? qp[3] : 0.f;
        }
      }
#pragma unroll
      for (int kz = 0; kz < KS; ++kz) {
#pragma unroll
        for (int ky = 0; ky < KS; ++ky) {
#pragma unroll
          for (int kx = 0; kx < KS; ++kx) {
            const int zp = zq * STRIDE + kz - PAD, yp = yq * STRIDE + ky - PAD, xp = xq * STRIDE + kx - PAD;
            float pv = 0.f;
            if constexpr (KS == 1 && STRIDE == 1 && PAD == 0) {
              pv = P[v * CA + a];
            } else if (zp >= 0 && zp < Dp && yp >= 0 && yp < Hp && xp >= 0 && xp < Wp_) {
              pv = P[((((i64)n * Dp + zp) * Hp + yp) * Wp_ + xp) * CA + a];
            }
            const int t = (kz * KS + ky) * KS + kx;
            acc[t].x = fmaf(pv, qv.x, acc[t].x);
            acc[t].y = fmaf(pv, qv.y, acc[t].y);
            acc[t].z = fmaf(pv, qv.z, acc[t].z);
            acc[t].w = fmaf(pv, qv.w, acc[t].w);
          }
        }
      }
    }
  }
  // combine the VL voxel lanes per pair, tap by tap, through LDS (fixed order -> reproducible)
#pragma unroll
  for (int t = 0; t < T; ++t) {
    __syncthreads();
    red[threadIdx.x] = acc[t];
    __syncthreads();
    if (vl == 0 && pair_ok) {
      float4 s = red[pl];
      for (int k = 1; k < VL; ++k) {
        const float4 o = red[k * PAIRS + pl];
        s.x += o.x; s.y += o.y; s.z += o.z; s.w += o.w;
      }
      float* dst = part + (((i64)blockIdx.x * T + t) * CA + a) * CBP + 4 * bq;
      *reinterpret_cast<float4*>(dst) = s;
    }
  }
}

// 1x1x1 convolution between two THIN tensors (the head's second conv, vnet_outblock.py:16: classes -> classes, <= 8 channels
// on both sides): one voxel per thread and trip, the CA x CB products in registers, one partial slab per workgroup in the
// layout of wgrad_direct_kernel (part[chunk][0][a][bp]).  The generic kernel above puts (a, b quad) pairs across threads: with
// 5 x 5 channels that is 10 pairs x 16 voxel lanes per workgroup and scalar loads behind each other -- 0.37 ms per step for
// 140 MB of input (vnet(1,5), BASELINE config 3) where this pass takes the time of reading them once.
__global__ __launch_bounds__(256) void wgrad_k1_thin_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                              float* __restrict__ part, i64 nvox, int CA, int CB, int CBP,
                                                              int chunk_vox) {
  __shared__ float red[4][64];
  const i64 v0 = (i64)blockIdx.x * chunk_vox;
  i64 v1 = v0 + chunk_vox;
  if (v1 > nvox) v1 = nvox;
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = 0.f;
  for (i64 v = v0 + threadIdx.x; v < v1; v += 256) {
    float pv[8], qv[8];
#pragma unroll
    for (int a = 0; a < 8; ++a) pv[a] = a < CA ? P[v * CA + a] : 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b) qv[b] = b < CB ? Q[v * CB + b] : 0.f;
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = fmaf(pv[a], qv[b], acc[a][b]);
  }
  // wave sums (shuffles), then the four waves in fixed order through LDS: reproducible
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) {
      const float sum = wave_sum(acc[a][b]);
      if (lane == 0) red[wave][a * 8 + b] = sum;
    }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int a = threadIdx.x >> 3, b = threadIdx.x & 7;
    if (a < CA && b < CBP) {
      const float sum = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
      part[((i64)blockIdx.x * CA + a) * CBP + b] = b < CB ? sum : 0.f;
    }
  }
}

// number of partial slabs: ~2048 voxels per chunk, at most 512 chunks and at most 64 MiB of slabs
static int wgrad_direct_chunks(i64 nvox, int T, int CA, int CBP) {
  i64 chunks = (nvox + 2047) / 2048;
  if (chunks > 512) chunks = 512;
  const i64 slab_bytes = (i64)T * CA * CBP * 4;
  const i64 cap = (64ll << 20) / slab_bytes;
  if (chunks > cap) chunks = cap;
  if (chunks < 1) chunks = 1;
  return (int)chunks;
}

extern "C" long long seg3d_wgrad_direct_workspace_floats(int N, int Dq, int Hq, int Wq, int CA, int CB, int ntaps) {
  const int CBP = seg3d_round_up(CB, 4);
  i64 nvox = (i64)N * Dq * Hq * Wq;
  return (long long)wgrad_direct_chunks(nvox, ntaps, CA, CBP) * ntaps * CA * CBP;
}

// P dims = (Dp,Hp,Wp), Q dims derived: k3/k1 same, k2s2: half.
extern "C" int seg3d_wgrad_direct(const float* P, const float* Q, float* part, int N, int Dp, int Hp, int Wp_, int CA,
                                  int CB, int ksize, int stride, int* n_chunks_out, void* stream) {
  SEG3D_REQUIRE(P && Q && part && n_chunks_out, "seg3d_wgrad_direct: null pointer");
  SEG3D_REQUIRE(N > 0 && Dp > 0 && Hp > 0 && Wp_ > 0 && CA > 0 && CB > 0, "seg3d_wgrad_direct: bad dims");
  int Dq = Dp, Hq = Hp, Wq = Wp_;
  if (stride == 2) {
    SEG3D_REQUIRE((Dp % 2) == 0 && (Hp % 2) == 0 && (Wp_ % 2) == 0, "seg3d_wgrad_direct: k2s2 needs even dims");
    Dq = Dp / 2; Hq = Hp / 2; Wq = Wp_ / 2;
  }
  const int CBP = seg3d_round_up(CB, 4);
  const int total_pairs = CA * (CBP / 4);
  int PAIRS = 1;
  while (PAIRS < total_pairs && PAIRS < 256) PAIRS <<= 1;  // power of two dividing 256
  const int gy = seg3d_cdiv(total_pairs, PAIRS);
  i64 nvox = (i64)N * Dq * Hq * Wq;
  int chunks = wgrad_direct_chunks(nvox, ksize * ksize * ksize, CA, CBP);
  const int chunk_vox = (int)((nvox + chunks - 1) / chunks);
  chunks = (int)((nvox + chunk_vox - 1) / chunk_vox);
  *n_chunks_out = chunks;
  hipStream_t s = (hipStream_t)stream;
  dim3 grid(chunks, gy), block(256);
  if (ksize == 3 && stride == 1) {
    hipLaunchKernelGGL((wgrad_direct_kernel<3, 1, 1>), grid, block, 0, s, P, Q, part, N, Dp, Hp, Wp_, Dq, Hq, Wq, CA, CB,
                       CBP, PAIRS, chunk_vox);
  } else if (ksize == 2 && stride == 2) {
    hipLaunchKernelGGL((wgrad_direct_kernel<2, 2, 0>), grid, block, 0, s, P, Q, part, N, Dp, Hp, Wp_, Dq, Hq, Wq, CA, CB,
                       CBP, PAIRS, chunk_vox);
  } else if (ksize == 1 && stride == 1 && CA <= 8 && CB <= 8) {
    hipLaunchKernelGGL(wgrad_k1_thin_kernel, dim3(chunks), block, 0, s, P, Q, part, nvox, CA, CB, CBP, chunk_vox);
  } else if (ksize == 1 && stride == 1) {
    hipLaunchKernelGGL((wgrad_direct_kernel<1, 1, 0>), grid, block, 0, s, P, Q, part, N, Dp, Hp, Wp_, Dq, Hq, Wq, CA, CB,
                       CBP, PAIRS, chunk_vox);
  } else {
    SEG3D_UNSUPPORTED("seg3d_wgrad_direct: unsupported ksize=%d stride=%d", ksize, stride);
  }
  SEG3D_LAUNCH_CHECK("seg3d_wgrad_direct");
  return SEG3D_OK;
}

// dw[a*sa + b*sb + t] = sum_chunk part[chunk][t][a][bp]   (fixed chunk order)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                             int chunks, int T, int A, int B, int BP, i64 sa, i64 sb,
                                                             int accumulate) {
  const i64 total = (i64)T * A * B;
  const i64 slab = (i64)T * A * BP;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int b = (int)(idx % B);
    i64 r = idx / B;
    const int a = (int)(r % A);
    const int t = (int)(r / A);
    const float* p = part + ((i64)t * A + a) * BP + b;
    float s = 0.f;
    for (int c = 0; c < chunks; ++c) s += p[(i64)c * slab];
    float* d = dw + a * sa + b * sb + t;
    *d = accumulate ? *d + s : s;
  }
}

// few outputs, many chunks (the head's 1x1x1 conv: 4 outputs x ~14 k chunks took 134 us with one serial thread per
// output): one workgroup per output, threads stride over the chunks, fixed-shape LDS tree -- still deterministic
__global__ __launch_bounds__(256) void wgrad_reduce_wide_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                  int chunks, int T, int A, int B, int BP, i64 sa, i64 sb,
                                                                  int accumulate) {
  __shared__ float red[256];
  const i64 idx = blockIdx.x;
  const i64 slab = (i64)T * A * BP;
  const int b = (int)(idx % B);
  const i64 r = idx / B;
  const int a = (int)(r % A);
  const int t = (int)(r / A);
  const float* p = part + ((i64)t * A + a) * BP + b;
  float s = 0.f;
  for (int c = threadIdx.x; c < chunks; c += 256) s += p[(i64)c * slab];
  red[threadIdx.x] = s;
  __syncthreads();
#pragma unroll
  for (int w = 128; w >= 1; w >>= 1) {
    if ((int)threadIdx.x < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    float* d = dw + a * sa + b * sb + t;
    *d = accumulate ? *d + red[0] : red[0];
  }
}

extern "C" int seg3d_wgrad_reduce(const float* part, float* dw, int chunks, int T, int A, int B, long long sa,
                                  long long sb, int accumulate, void* stream) {
  SEG3D_REQUIRE(part && dw && chunks > 0 && T > 0 && A > 0 && B > 0, "seg3d_wgrad_reduce: bad arguments");
  const int BP = seg3d_round_up(B, 4);
  i64 total = (i64)T * A * B;
  if (chunks >= 512 && total <= 2048)
    hipLaunchKernelGGL(wgrad_reduce_wide_kernel, dim3((unsigned)total), dim3(256), 0, (hipStream_t)stream, part, dw, chunks, T, A,
                       B, BP, (i64)sa, (i64)sb, accumulate);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, part, dw,
                       chunks, T, A, B, BP, (i64)sa, (i64)sb, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_wgrad_reduce");
  return SEG3D_OK;
}
