// conv_k2_mfma.hip -- the stride-2 2x2x2 layers of the V-Net on the matrix cores (exact fp32 MFMA):
//   DownBlock  nn.Conv3d(C, 2C, kernel_size=2, stride=2)            network/module/vnet_downblock.py:11
//   UpBlock    nn.ConvTranspose3d(Cin, Cout/2, kernel_size=2, stride=2)   network/module/vnet_upblock.py:11
// and their autograd adjoints.  The 2^3 cells do not overlap, so both are plain GEMMs with a strided gather
// (conv: M = output voxels, K = 8 taps x Cin) or a strided scatter (transposed conv: 8 GEMMs M = input voxels,
// K = Cin, one per tap, sharing the A tile).  At the top level they are HBM-bound (arithmetic intensity 13..21
// FLOP/B); below it they live in L2.  The VALU versions in conv_direct.hip stay as fallback.
//
//   gather  kernel (conv3d_k2s2_mfma_kernel):   y[v][b] = bias[b] + sum_{t,a} x[2v + t][a] W(a,b,t)
//       = Conv3d k2s2 forward, and the data-gradient of ConvTranspose3d k2s2
//   scatter kernel (convT3d_k2s2_mfma_kernel):  y[2i + t][b] = bias[b] + sum_a x[i][a] W(a,b,t)
//       = ConvTranspose3d k2s2 forward, and the data-gradient of Conv3d k2s2
//   pair-reduce kernel (k2_wgrad_mfma_kernel):  dW(t,a,b) = sum_v P[2v + t][a] Q[v][b]
//       = weight gradient of both (conv: P = x, Q = dy; transposed: P = dy, Q = x)
// Operand staging and the k = {r, 4 + r} pairing are those of conv_mfma.hip; weights come from pack_mfma_kernel
// with T = 8 (chunk image [8][half][32][4]).
#include "seg3d_common.h"
#include "seg3d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define K2_W_CHUNK 2048  // 8 * 2 * 32 * 4 floats
#define K2_MAXE 8        // float4 input loads per thread per chunk: 2 * NV <= 2048  (NV <= 1024 input voxels)

__device__ __forceinline__ int k2_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

struct K2Tile {
  int tz, ty, tx;
};

// tile of at most 128 "M" voxels (4 waves x one 32-row accumulator block)
static K2Tile k2_pick_tile(int D, int H, int W) {
  const int cz[] = {1, 2, 4, 8}, cy[] = {1, 2, 4, 8}, cx[] = {2, 4, 8, 16, 32};
  K2Tile best = {1, 1, 32};
  double best_cost = 1e30;
  for (int tz : cz)
    for (int ty : cy)
      for (int tx : cx) {
        const int mt = tz * ty * tx;
        if (mt > 128 || mt < 32) continue;
        const double tiles = (double)seg3d_cdiv(D, tz) * seg3d_cdiv(H, ty) * seg3d_cdiv(W, tx);
        double cost = tiles * 128.0 / ((double)D * H * W);  // MFMA rows issued / useful rows
        cost *= 1.0 + 0.02 * (8.0 / tx);                     // prefer long contiguous x runs
        if (cost < best_cost) {
          best_cost = cost;
          best = {tz, ty, tx};
        }
      }
  return best;
}

// ---------------------------------------------------------------------------------------------------------------
// gather: output tile TZ x TY x TX (<= 128 voxels), input tile 2TZ x 2TY x 2TX staged 8 channels at a time
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void conv3d_k2s2_mfma_kernel(const float* __restrict__ x,
                                                                    const float* __restrict__ wp,
                                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                                    float* __restrict__ stats, int N, int Do, int Ho,
                                                                    int Wo, int Cin, int Cout, int TZ, int TY, int TX,
                                                                    int ntz, int nty, int ntx) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HY = 2 * TY, HX = 2 * TX;
  const int NV = 8 * TZ * TY * TX;
  const int MT = TZ * TY * TX;
  float* xs = lds;                                       // [2][NV][4]
  float* ws = lds + 8 * NV;                              // [8][2][32][4]
  int* voff = reinterpret_cast<int*>(ws + K2_W_CHUNK);   // [MT]
  const int Di = 2 * Do, Hi = 2 * Ho, Wi = 2 * Wo;

  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int CIB = (Cin + 7) >> 3;
  const int cob = blockIdx.y;
  int b = blockIdx.x;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * TX;

  int goff[K2_MAXE];
  const int hh = tid & 1;
#pragma unroll
  for (int e = 0; e < K2_MAXE; ++e) {
    const int eidx = tid + e * 256;
    goff[e] = -1;
    if (eidx < 2 * NV) {
      const int v = eidx >> 1;
      const int hx = v % HX;
      const int t = v / HX;
      const int hy = t % HY;
      const int hz = t / HY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      if (gz < Di && gy < Hi && gx < Wi) goff[e] = (((n * Di + gz) * Hi + gy) * Wi + gx) * Cin + hh * 4;
    }
  }
  for (int idx = tid; idx < MT; idx += 256) {
    const int tx = idx % TX;
    const int t = idx / TX;
    const int ty = t % TY;
    const int tz = t / TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    voff[idx] = (gz < Do && gy < Ho && gx < Wo) ? ((n * Do + gz) * Ho + gy) * Wo + gx : -1;
  }
  int abase;
  {
    const int idx = wave * 32 + li;
    int vb = 0;
    if (idx < MT) {
      const int tx = idx % TX;
      const int t = idx / TX;
      const int ty = t % TY;
      const int tz = t / TY;
      vb = ((2 * tz) * HY + 2 * ty) * HX + 2 * tx;
    }
    abase = (lh * NV + vb) * 4;
  }
  const int bbase = (lh * 32 + li) * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  for (int cib = 0; cib < CIB; ++cib) {
    __syncthreads();
    const bool half_ok = cib * 8 + hh * 4 < Cin;
#pragma unroll
    for (int e = 0; e < K2_MAXE; ++e) {
      const int eidx = tid + e * 256;
      if (eidx < 2 * NV) {
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (goff[e] >= 0 && half_ok) val = *reinterpret_cast<const f32x4*>(x + (i64)goff[e] + cib * 8);
        *reinterpret_cast<f32x4*>(xs + (hh * NV + (eidx >> 1)) * 4) = val;
      }
    }
    {
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + ((i64)cob * CIB + cib) * K2_W_CHUNK);
      f32x4* wdst = reinterpret_cast<f32x4*>(ws);
#pragma unroll
      for (int k = 0; k < 2; ++k) wdst[tid + k * 256] = wsrc[tid + k * 256];
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
      const int tapoff = ((kz * HY + ky) * HX + kx) * 4;
      const f32x4 bw = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase);
      const f32x4 av = *reinterpret_cast<const f32x4*>(xs + abase + tapoff);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bw[r], acc, 0, 0, 0);
    }
  }

  const int co = cob * 32 + li;
  const bool co_ok = co < Cout;
  const float bv = (bias && co_ok) ? bias[co] : 0.f;
  float s[2] = {0.f, 0.f};
  // two passes: every store reads its own accumulator register (no shared temporary -> no vmcnt wait per store)
  int ooff[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int idx = wave * 32 + k2_row(r, lh);
    const int vo = idx < MT ? voff[idx] : -1;
    const bool ok = vo >= 0 && co_ok;
    ooff[r] = ok ? vo * Cout + co : -1;
    acc[r] += bv;
    const float val = ok ? acc[r] : 0.f;
    s[0] += val;
    s[1] += val * val;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r)
    if (ooff[r] >= 0) y[(i64)ooff[r]] = acc[r];
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, xs);
    if (tid == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + (((i64)n * tiles_per_sample + tile) * gridDim.y + cob) * 2;
      dst[0] = s[0];
      dst[1] = s[1];
    }
  }
}

extern "C" long long seg3d_conv3d_k2s2_mfma_stats_count(int Do, int Ho, int Wo, int Cout) {
  K2Tile t = k2_pick_tile(Do, Ho, Wo);
  return (long long)seg3d_cdiv(Do, t.tz) * seg3d_cdiv(Ho, t.ty) * seg3d_cdiv(Wo, t.tx) * ((Cout + 31) / 32);
}

// x [N][2Do][2Ho][2Wo][Cin] -> y [N][Do][Ho][Wo][Cout];  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 8)
extern "C" int seg3d_conv3d_k2s2_mfma_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int N,
                                          int Do, int Ho, int Wo, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k2s2_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && Do > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k2s2_mfma_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0, "seg3d_conv3d_k2s2_mfma_fwd: Cin must be a multiple of 4 (got %d)", Cin);
  SEG3D_REQUIRE((i64)N * Do * Ho * Wo * 8 * Cin < (1ll << 31) && (i64)N * Do * Ho * Wo * Cout < (1ll << 31),
                "seg3d_conv3d_k2s2_mfma_fwd: tensor exceeds 2^31 elements");
  K2Tile t = k2_pick_tile(Do, Ho, Wo);
  const int ntz = seg3d_cdiv(Do, t.tz), nty = seg3d_cdiv(Ho, t.ty), ntx = seg3d_cdiv(Wo, t.tx);
  const int mt = t.tz * t.ty * t.tx;
  const size_t lds = (size_t)(8 * 8 * mt + K2_W_CHUNK + ((mt + 3) & ~3)) * 4;
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  hipLaunchKernelGGL(conv3d_k2s2_mfma_kernel, grid, dim3(256), lds, (hipStream_t)stream, x, wp, bias, y, stats, N, Do, Ho,
                     Wo, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k2s2_mfma_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// scatter: input tile TZ x TY x TX (<= 128 voxels), one accumulator per tap, output cell 2^3 per input voxel
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void convT3d_k2s2_mfma_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ wp,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ stats,
                                                                     int N, int Di, int Hi, int Wi, int Cin, int Cout,
                                                                     int TZ, int TY, int TX, int ntz, int nty, int ntx) {
  __shared__ __attribute__((aligned(16))) float xs[2 * 128 * 4];     // [2][MT<=128][4]
  __shared__ __attribute__((aligned(16))) float ws[K2_W_CHUNK];      // [8][2][32][4]
  __shared__ int obase[128];                                         // output voxel index of tap (0,0,0) or -1
  const int MT = TZ * TY * TX;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int CIB = (Cin + 7) >> 3;
  const int cob = blockIdx.y;
  int b = blockIdx.x;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * TX;

  // one staged float4 per thread: voxel tid >> 1, half tid & 1
  const int sv = tid >> 1, hh = tid & 1;
  int goff = -1;
  if (sv < MT) {
    const int tx = sv % TX;
    const int t = sv / TX;
    const int ty = t % TY;
    const int tz = t / TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    if (gz < Di && gy < Hi && gx < Wi) {
      goff = (((n * Di + gz) * Hi + gy) * Wi + gx) * Cin + hh * 4;
      if (hh == 0) obase[sv] = (((n * 2 * Di + 2 * gz) * Ho + 2 * gy) * Wo + 2 * gx);
    } else if (hh == 0) {
      obase[sv] = -1;
    }
  }
  const int abase = (lh * MT + wave * 32 + li) * 4;  // rows >= MT read garbage inside xs (never stored)
  const int bbase = (lh * 32 + li) * 4;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  for (int cib = 0; cib < CIB; ++cib) {
    __syncthreads();
    if (sv < MT) {
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (goff >= 0 && cib * 8 + hh * 4 < Cin) val = *reinterpret_cast<const f32x4*>(x + (i64)goff + cib * 8);
      *reinterpret_cast<f32x4*>(xs + (hh * MT + sv) * 4) = val;
    }
    {
      const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + ((i64)cob * CIB + cib) * K2_W_CHUNK);
      f32x4* wdst = reinterpret_cast<f32x4*>(ws);
#pragma unroll
      for (int k = 0; k < 2; ++k) wdst[tid + k * 256] = wsrc[tid + k * 256];
    }
    __syncthreads();
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (wave * 32 + li < MT) av = *reinterpret_cast<const f32x4*>(xs + abase);
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const f32x4 bw = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase);
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[r], bw[r], acc[tap], 0, 0, 0);
    }
  }

  const int co = cob * 32 + li;
  const bool co_ok = co < Cout;
  const float bv = (bias && co_ok) ? bias[co] : 0.f;
  float s[2] = {0.f, 0.f};
  int ob[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int idx = wave * 32 + k2_row(r, lh);
    const int o = idx < MT ? obase[idx] : -1;
    const bool ok = o >= 0 && co_ok;
    ob[r] = ok ? o : -1;
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      acc[tap][r] += bv;
      const float val = ok ? acc[tap][r] : 0.f;
      s[0] += val;
      s[1] += val * val;
    }
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    if (ob[r] >= 0) {
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) {
        const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
        y[((i64)ob[r] + (kz * Ho + ky) * Wo + kx) * Cout + co] = acc[tap][r];
      }
    }
  }
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, xs);
    if (tid == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + (((i64)n * tiles_per_sample + tile) * gridDim.y + cob) * 2;
      dst[0] = s[0];
      dst[1] = s[1];
    }
  }
}

extern "C" long long seg3d_convT3d_k2s2_mfma_stats_count(int Di, int Hi, int Wi, int Cout) {
  K2Tile t = k2_pick_tile(Di, Hi, Wi);
  return (long long)seg3d_cdiv(Di, t.tz) * seg3d_cdiv(Hi, t.ty) * seg3d_cdiv(Wi, t.tx) * ((Cout + 31) / 32);
}

// x [N][Di][Hi][Wi][Cin] -> y [N][2Di][2Hi][2Wi][Cout];  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 8)
extern "C" int seg3d_convT3d_k2s2_mfma_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats,
                                           int N, int Di, int Hi, int Wi, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_convT3d_k2s2_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "seg3d_convT3d_k2s2_mfma_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0, "seg3d_convT3d_k2s2_mfma_fwd: Cin must be a multiple of 4 (got %d)", Cin);
  SEG3D_REQUIRE((i64)N * Di * Hi * Wi * 8 * Cout < (1ll << 31) && (i64)N * Di * Hi * Wi * Cin < (1ll << 31),
                "seg3d_convT3d_k2s2_mfma_fwd: tensor exceeds 2^31 elements");
  K2Tile t = k2_pick_tile(Di, Hi, Wi);
  const int ntz = seg3d_cdiv(Di, t.tz), nty = seg3d_cdiv(Hi, t.ty), ntx = seg3d_cdiv(Wi, t.tx);
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  hipLaunchKernelGGL(convT3d_k2s2_mfma_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, stats, N, Di, Hi,
                     Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx);
  SEG3D_LAUNCH_CHECK("seg3d_convT3d_k2s2_mfma_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// pair-reduce weight gradient: dW(t,a,b) = sum_v P[2v + t][a] Q[v][b]
// ---------------------------------------------------------------------------------------------------------------
#define K2W_TZ 2
#define K2W_TY 4
#define K2W_TX 8
#define K2W_MT (K2W_TZ * K2W_TY * K2W_TX)   // 64 Q voxels per tile
#define K2W_HY (2 * K2W_TY)
#define K2W_HX (2 * K2W_TX)
#define K2W_NV (8 * K2W_MT)                 // 512 P voxels per tile

__global__ __launch_bounds__(256, 2) void k2_wgrad_mfma_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                 float* __restrict__ part, int N, int Dq, int Hq, int Wq,
                                                                 int CA, int CB, int ntz, int nty, int ntx, int ntiles,
                                                                 int BB32) {
  __shared__ __attribute__((aligned(16))) float ps[K2W_NV * 32];
  __shared__ __attribute__((aligned(16))) float qs[K2W_MT * 32];
  const int Dp = 2 * Dq, Hp = 2 * Hq, Wp = 2 * Wq;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ab = blockIdx.y / BB32, bb = blockIdx.y % BB32;
  const int a0 = ab * 32, b0 = bb * 32;
  int tapoff[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tap = wave * 2 + j;
    const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
    tapoff[j] = ((kz * K2W_HY + ky) * K2W_HX + kx) * 32;
  }
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  const int q = tid & 7;
  const bool pq_ok = a0 + 4 * q < CA;
  const bool qq_ok = b0 + 4 * q < CB;

  // register-prefetch pipeline over the tile loop (see conv3d_k3_wgrad_mfma_kernel): this kernel is HBM-bound at the
  // top level, so keeping the next tile's 18 x 16-byte loads per thread in flight behind the MFMA block is what
  // keeps the memory system busy
  constexpr int PE = (K2W_NV * 8) / 256, QE = (K2W_MT * 8) / 256;
  f32x4 pst[PE], qst[QE];
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * K2W_TZ, y0 = tiy * K2W_TY, x0 = tix * K2W_TX;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const int v = (tid + e * 256) >> 3;
      const int hx = v % K2W_HX;
      const int t = v / K2W_HX;
      const int hy = t % K2W_HY;
      const int hz = t / K2W_HY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      const bool ok = pq_ok && gz < Dp && gy < Hp && gx < Wp;
      const f32x4 val = *reinterpret_cast<const f32x4*>(
          P + (ok ? ((((i64)n * Dp + gz) * Hp + gy) * Wp + gx) * CA + a0 + 4 * q : (i64)0));
      pst[e] = ok ? val : zero;
    }
#pragma unroll
    for (int e = 0; e < QE; ++e) {
      const int v = (tid + e * 256) >> 3;
      const int tx = v % K2W_TX;
      const int t = v / K2W_TX;
      const int ty = t % K2W_TY;
      const int tz = t / K2W_TY;
      const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
      const bool ok = qq_ok && gz < Dq && gy < Hq && gx < Wq;
      const f32x4 val = *reinterpret_cast<const f32x4*>(
          Q + (ok ? ((((i64)n * Dq + gz) * Hq + gy) * Wq + gx) * CB + b0 + 4 * q : (i64)0));
      qst[e] = ok ? val : zero;
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int e = 0; e < PE; ++e) *reinterpret_cast<f32x4*>(ps + ((tid + e * 256) >> 3) * 32 + 4 * q) = pst[e];
#pragma unroll
    for (int e = 0; e < QE; ++e) *reinterpret_cast<f32x4*>(qs + ((tid + e * 256) >> 3) * 32 + 4 * q) = qst[e];
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
#pragma unroll 4
    for (int kp = 0; kp < K2W_MT / 2; ++kp) {
      const int v = 2 * kp + lh;
      const int tx = v % K2W_TX;
      const int t = v / K2W_TX;
      const int ty = t % K2W_TY;
      const int tz = t / K2W_TY;
      const int base = (((2 * tz) * K2W_HY + 2 * ty) * K2W_HX + 2 * tx) * 32 + li;
      const float bvv = qs[v * 32 + li];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const float a = ps[base + tapoff[j]];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[j], 0, 0, 0);
      }
    }
  }
  float* dst = part + ((i64)blockIdx.x * gridDim.y + blockIdx.y) * 8 * 1024;
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tap = wave * 2 + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[tap * 1024 + k2_row(r, lh) * 32 + li] = acc[j][r];
  }
}

// dw[a*sa + b*sb + t] = sum_slab part[slab][a/32][b/32][t][a%32][b%32]   (a = reduction-side channel, b = output channel)
// 64 outputs per workgroup, 4 slab groups per output (coalesced reads of every slab), combined in a fixed order.
__global__ __launch_bounds__(256) void k2_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int slabs, int A,
             int B, int BB32, int npairs, i64 sa, i64 sb, int accumulate) {
  constexpr int T = 8;
  __shared__ float red[256];
  const i64 total = (i64)npairs * T * 1024;
  const i64 pidx = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  float s = 0.f;
  if (pidx < total) {
    const float* p = part + pidx;
    for (int k = g; k < slabs; k += 4) s += p[(i64)k * total];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && pidx < total) {
    const float v = (red[threadIdx.x] + red[64 + threadIdx.x]) + (red[128 + threadIdx.x] + red[192 + threadIdx.x]);
    const int b32 = (int)(pidx & 31), a32 = (int)((pidx >> 5) & 31);
    const i64 r = pidx >> 10;
    const int t = (int)(r % T);
    const int pair = (int)(r / T);
    const int a = (pair / BB32) * 32 + a32, b = (pair % BB32) * 32 + b32;
    if (a < A && b < B) {
      float* d = dw + a * sa + b * sb + t;
      *d = accumulate ? *d + v : v;
    }
  }
}

static int k2_wgrad_slabs(int N, int Dq, int Hq, int Wq, int npairs) {
  const int ntiles = N * seg3d_cdiv(Dq, K2W_TZ) * seg3d_cdiv(Hq, K2W_TY) * seg3d_cdiv(Wq, K2W_TX);
  int slabs = 512 / npairs;
  if (slabs > (ntiles + 3) / 4) slabs = (ntiles + 3) / 4;
  if (slabs < 1) slabs = 1;
  return slabs;
}

extern "C" long long seg3d_k2_mfma_wgrad_workspace_floats(int N, int Dq, int Hq, int Wq, int CA, int CB) {
  const int npairs = ((CA + 31) / 32) * ((CB + 31) / 32);
  return (long long)k2_wgrad_slabs(N, Dq, Hq, Wq, npairs) * npairs * 8 * 1024;
}

// P [N][2Dq][2Hq][2Wq][CA], Q [N][Dq][Hq][Wq][CB];  dw[a*sa + b*sb + t] (t < 8) receives the gradient
extern "C" int seg3d_k2_mfma_wgrad(const float* P, const float* Q, float* dw, float* workspace, int N, int Dq, int Hq,
                                   int Wq, int CA, int CB, long long sa, long long sb, int accumulate, void* stream) {
  SEG3D_REQUIRE(P && Q && dw && workspace, "seg3d_k2_mfma_wgrad: null pointer");
  SEG3D_REQUIRE(N > 0 && Dq > 0 && Hq > 0 && Wq > 0 && CA > 0 && CB > 0, "seg3d_k2_mfma_wgrad: bad dims");
  SEG3D_REQUIRE((CA % 4) == 0 && (CB % 4) == 0, "seg3d_k2_mfma_wgrad: channel counts must be multiples of 4 (got %d, %d)",
                CA, CB);
  const int AB32 = (CA + 31) / 32, BB32 = (CB + 31) / 32;
  const int npairs = AB32 * BB32;
  const int ntz = seg3d_cdiv(Dq, K2W_TZ), nty = seg3d_cdiv(Hq, K2W_TY), ntx = seg3d_cdiv(Wq, K2W_TX);
  const int ntiles = N * ntz * nty * ntx;
  const int slabs = k2_wgrad_slabs(N, Dq, Hq, Wq, npairs);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(k2_wgrad_mfma_kernel, dim3(slabs, npairs), dim3(256), 0, s, P, Q, workspace, N, Dq, Hq, Wq, CA, CB, ntz,
                     nty, ntx, ntiles, BB32);
  SEG3D_LAUNCH_CHECK("seg3d_k2_mfma_wgrad");
  const i64 total = (i64)npairs * 8 * 1024;
  hipLaunchKernelGGL(k2_wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace, dw, slabs, CA, CB,
                     BB32, npairs, (i64)sa, (i64)sb, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_k2_mfma_wgrad(reduce)");
  return SEG3D_OK;
}
