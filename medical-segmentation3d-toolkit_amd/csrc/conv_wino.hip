// conv_wino.hip -- C -> C nn.Conv3d(k3, p1) forward / data-gradient with Winograd F(2, 3) along x on the fp32 matrix
// cores (network/module/conv_gn_relu3.py:10; the data-gradient is the same kernel on tap-flipped, transposed weights).
//
// Why: these layers are 99 % of the V-Net's FLOPs and run on v_mfma_f32_32x32x2_f32 at 0.77-0.83 of its peak
// (conv_mfma.hip); the fp32 MFMA shares the VALU datapath, so what is left is overhead that cannot be
// hidden -- the remaining lever is the multiply count.  F(2, 3) computes two outputs that are neighbours in x from four
// inputs with 4 multiplies instead of 6:
//     d0..d3 = in[x0 - 1 .. x0 + 2]            D0 = d0 - d2, D1 = d1 + d2, D2 = d2 - d1, D3 = d1 - d3
//     g0..g2 = the kx taps of one (kz, ky)      U0 = g0, U1 = (g0 + g1 + g2) / 2, U2 = (g0 - g1 + g2) / 2, U3 = g2
//     M_p = sum_{kz, ky, ci} U_p D_p            y[x0] = M0 + M1 + M2,  y[x0 + 1] = M1 - M2 - M3
// i.e. four implicit GEMMs with K = 9 Cin instead of one with K = 27 Cin per output PAIR: 2/3 of the MFMAs.  The
// transforms are exact in fp32 up to one rounding each (coefficients 1, 1/2), the accumulation is the same fp32 MFMA
// chain; measured against float64 the error stays at the direct kernel's level (tests/test_gpu_kernels.py).
//
// Structure = the persistent kernel of conv_mfma.hip (one workgroup per CU walks (tile, column block) items, K chunks of 8
// input channels arrive by LDS-DMA, packed weight images are straight 1-KiB copies), with
//   * tile 8 x 8 x 8 voxels = 256 output pairs; eight waves (two per SIMD), wave w owns plane tz = w: 32 pairs, 4 point
//     accumulators of 16 registers;
//   * per chunk the RAW halo tile [half][10 x 10 x 10 voxels][4] lands by DMA while the previous chunk is multiplied; then
//     all waves transform it once into T [half][p][row][pair][4] (12.8 k adds per chunk against 144 MFMAs per wave: < 2 %),
//     two barriers per chunk instead of one; RAW needs only one buffer (it is dead once transformed), weights two;
//   * 36 MFMA steps per chunk: (kz, ky) x p, operands = one ds_read_b128 each, conflict-free (a wave's 32 pairs of one
//     (p, row block) are 512 contiguous bytes of T);
//   * LDS: RAW 32 KB + T 51 KB + 2 x 36 KB weights = 157.7 KB.
// Levels that are not multiples of the tile, channel counts that are not multiples of 8 / 32, and the spatially tiny
// levels (12^3, 6^3: split-K territory) stay on conv_mfma.hip.
#include "seg3d_common.h"
#include "seg3d_hip.h"
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define WN_TZ 8
#define WN_TY 8
#define WN_TX 8
#define WN_PX (WN_TX / 2)                         // output pairs per tile row
#define WN_HY (WN_TY + 2)
#define WN_HX (WN_TX + 2)
#define WN_NV ((WN_TZ + 2) * WN_HY * WN_HX)       // 1000 halo voxels
#define WN_NR ((WN_TZ + 2) * WN_HY)               // 100 halo rows
#define WN_RAW 8192                               // floats: [2][NV][4] padded to whole 1-KiB DMA pieces (32)
#define WN_T (2 * 4 * WN_NR * WN_PX * 4)          // floats of the transformed image: 12800
#define WN_W (36 * 256)                           // floats of one chunk's weight image: [36][2][32][4]
#define WN_NW 8
#define WN_LDS_FLOATS (WN_RAW + WN_T + 2 * WN_W)  // 39424 floats = 157696 bytes
#define WN_XPW 4                                  // raw pieces per wave (32 / 8)
#define WN_WPW 5                                  // weight pieces per wave (ceil(36 / 8))

__device__ __attribute__((aligned(16))) float wn_zero16[4];   // DMA source for zero padding

__device__ __forceinline__ void wn_glds16(const float* src, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds(src, lds_dst_wave_uniform, 16, 0, 0);
}
// The same LDS-DMA as inline assembly, for kernels with ONE wave per SIMD: while a builtin DMA is outstanding hipcc's
// wait-count pass treats it as a pending FLAT access and turns every LDS wait of the loop into lgkmcnt(0)
// (tools/ubench/waitcnt_dma.hip), a full drain of the operand reads issued for the following steps.  The compiler does not
// see this load: the kernel waits for it itself (wn_dma_wait) before the barrier that publishes the data, and uses M0 for
// nothing else.
typedef __attribute__((address_space(3))) float wn_lds_float;
__device__ __forceinline__ void wn_glds16_asm(const float* src, float* lds_dst_wave_uniform) {
  const unsigned off = (unsigned)(uintptr_t)(wn_lds_float*)lds_dst_wave_uniform;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(off) : "memory", "m0");
}
__device__ __forceinline__ void wn_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(512, 1) void conv3d_k3_wino_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                 const float* __restrict__ bias, float* __restrict__ y,
                                                                 float* __restrict__ stats, int N, int D, int H, int W, int Cin,
                                                                 int Cout, int ntz, int nty, int ntx, int ncog, int nitems,
                                                                 const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                       // [2][NV][4] (+ padding)
  float* timg = lds + WN_RAW;             // [2][4][NR][PX][4]
  float* wbuf = lds + WN_RAW + WN_T;      // [2][36][2][32][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int CIB = Cin >> 3;
  const int G = gridDim.x;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz, rNCOG = 1.0f / (float)ncog;

  // ---- per-lane constants ----
  // raw DMA pieces of this wave: piece p = wave + 8 j covers float4 entries e = 64 p + lane of [2][NV];
  // hpos[j] = halo coordinates (hz << 20 | hy << 10 | hx), bit 30 = upper channel half, -1 = padding entry
  int hpos[WN_XPW];
#pragma unroll
  for (int j = 0; j < WN_XPW; ++j) {
    const int e = (wave + WN_NW * j) * 64 + lane;
    hpos[j] = -1;
    if (e < 2 * WN_NV) {
      const int hh = e >= WN_NV;
      const int v = e - hh * WN_NV;
      const int t = v / WN_HX;
      const int hx = v - t * WN_HX;
      const int hz = t / WN_HY;
      const int hy = t - hz * WN_HY;
      hpos[j] = (hh << 30) | (hz << 20) | (hy << 10) | hx;
    }
  }
  // this lane's output pair: wave = plane tz, li = (ty, px); T offset of its (kz, ky) = (0, 0) row, point 0
  const int ty = li >> 2, px = li & 3;
  const int abase = (((lh * 4) * WN_NR + wave * WN_HY + ty) * WN_PX + px) * 4;
  const int bbase = (lh * 32 + li) * 4;
  // transform items of this thread: item i = tid + 512 k < 800 -> (half, row, pair)
  int t_src[2], t_dst[2];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    const int i = tid + 512 * k;
    t_src[k] = -1;
    t_dst[k] = 0;
    if (i < 2 * WN_NR * WN_PX) {
      const int hh = i / (WN_NR * WN_PX);
      const int r = i - hh * (WN_NR * WN_PX);
      const int row = r >> 2, p4 = r & 3;
      t_src[k] = (hh * WN_NV + row * WN_HX + 2 * p4) * 4;
      t_dst[k] = (((hh * 4) * WN_NR + row) * WN_PX + p4) * 4;
    }
  }

  // ---- work item state ----
  int it_n = 0, it_z0 = 0, it_y0 = 0, it_x0 = 0, it_cog = 0, it_tile = 0;
  const float* xsrc[WN_XPW];
  int xadv = 0;
  auto setup_item = [&](int item) {
    const int tile_all = fdiv(item, rNCOG);
    it_cog = item - tile_all * ncog;
    int b = tile_all;
    int q = fdiv(b, rNTX);
    const int tix = b - q * ntx;
    b = q;
    q = fdiv(b, rNTY);
    const int tiy = b - q * nty;
    b = q;
    q = fdiv(b, rNTZ);
    const int tiz = b - q * ntz;
    it_n = q;
    it_tile = (tiz * nty + tiy) * ntx + tix;
    it_z0 = tiz * WN_TZ, it_y0 = tiy * WN_TY, it_x0 = tix * WN_TX;
    xadv = 0;
#pragma unroll
    for (int j = 0; j < WN_XPW; ++j) {
      xsrc[j] = wn_zero16;
      const int hp = hpos[j];
      const int gz = it_z0 + ((hp >> 20) & 1023) - 1, gy = it_y0 + ((hp >> 10) & 1023) - 1, gx = it_x0 + (hp & 1023) - 1;
      if (hp >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        xsrc[j] = x + ((i64)(((it_n * D + gz) * H + gy) * W + gx) * Cin + ((hp >> 30) & 1) * 4);
        xadv |= 1 << j;
      }
    }
  };
  auto dma_x = [&](int j) {  // issues the piece into RAW, then steps its source to the next chunk
    wn_glds16(xsrc[j], raw + (wave + WN_NW * j) * 256);
    xsrc[j] += ((xadv >> j) & 1) * 8;
  };
  auto dma_w = [&](int j, const float* wchunk, float* wdst) {
    const int piece = wave + WN_NW * j;
    if (piece < 36) wn_glds16(wchunk + piece * 256 + lane * 4, wdst + piece * 256);
  };
  auto transform = [&]() {   // RAW -> T: D0 = d0 - d2, D1 = d1 + d2, D2 = d2 - d1, D3 = d1 - d3 per (half, row, pair)
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      if (t_src[k] >= 0) {
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(raw + t_src[k]);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(raw + t_src[k] + 4);
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(raw + t_src[k] + 8);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(raw + t_src[k] + 12);
        float* dst = timg + t_dst[k];
        *reinterpret_cast<f32x4*>(dst) = d0 - d2;
        *reinterpret_cast<f32x4*>(dst + WN_NR * WN_PX * 4) = d1 + d2;
        *reinterpret_cast<f32x4*>(dst + 2 * WN_NR * WN_PX * 4) = d2 - d1;
        *reinterpret_cast<f32x4*>(dst + 3 * WN_NR * WN_PX * 4) = d1 - d3;
      }
    }
  };

  // item walk: XCD-contiguous eighths of the item list (as conv_mfma.hip)
  int item = blockIdx.x, istride = G, ilimit = nitems;
  if ((G & 7) == 0) {
    const int per_xcd = (nitems + 7) >> 3, xcd = blockIdx.x & 7;
    item = xcd * per_xcd + (blockIdx.x >> 3);
    istride = G >> 3;
    ilimit = (xcd + 1) * per_xcd < nitems ? (xcd + 1) * per_xcd : nitems;
  }
  if (item >= ilimit) return;
  setup_item(item);
  {  // the only exposed DMA prologue of this workgroup: chunk 0 of its first item
    const float* w0 = wp + (i64)it_cog * CIB * WN_W;
#pragma unroll
    for (int j = 0; j < WN_XPW; ++j) dma_x(j);
#pragma unroll
    for (int j = 0; j < WN_WPW; ++j) dma_w(j, w0, wbuf);
  }
  __syncthreads();   // drains this wave's DMA count (vmcnt) and publishes RAW / weights
  transform();
  __syncthreads();

  int parity = 0;
  for (;;) {
    const int cur_n = it_n, cur_z0 = it_z0, cur_y0 = it_y0, cur_x0 = it_x0, cur_cog = it_cog, cur_tile = it_tile;
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;

    f32x16 acc[4];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    f32x4 bv[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co = cur_cog * 32 + 4 * lh + 8 * g4;
      bv[g4] = *reinterpret_cast<const f32x4*>((bias && co < Cout) ? bias + co : wn_zero16);
    }
    // the lane's two output voxels (x0 + 2 px, + 1) -- whole tiles only (host-checked): always inside the volume
    const int vo0 = ((cur_n * D + cur_z0 + wave) * H + cur_y0 + ty) * W + cur_x0 + 2 * px;
    f32x4 ad[2][4];  // fused addend values of this item (loaded at the start of its last chunk)
    for (int cib = 0; cib < CIB; ++cib) {
      const float* ws = wbuf + parity * WN_W;
      float* wnext_dst = wbuf + (parity ^ 1) * WN_W;
      const bool last = cib + 1 == CIB;
      const bool do_dma = !last || more_items;
      const float* wnext;
      if (last) {
        if (addend) {
#pragma unroll
          for (int m = 0; m < 2; ++m)
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
              const int co = cur_cog * 32 + 4 * lh + 8 * g4;
              ad[m][g4] = *reinterpret_cast<const f32x4*>(addend + (co < Cout ? (i64)(vo0 + m) * Cout + co : (i64)0));
            }
        }
        if (more_items) setup_item(next_item);  // DMA sources now belong to the next item
        wnext = wp + (i64)it_cog * CIB * WN_W;
      } else {
        wnext = wp + ((i64)cur_cog * CIB + cib + 1) * WN_W;
      }
      f32x4 bw = *reinterpret_cast<const f32x4*>(ws + bbase);
      f32x4 av = *reinterpret_cast<const f32x4*>(timg + abase);
#pragma unroll
      for (int st = 0; st < 36; ++st) {
        f32x4 bwn = bw, avn = av;
        if (st + 1 < 36) {   // operands of step st + 1 are read while step st is multiplied
          const int s1 = st + 1, t9 = s1 >> 2, p = s1 & 3;
          const int kz = t9 / 3, ky = t9 - 3 * kz;
          bwn = *reinterpret_cast<const f32x4*>(ws + s1 * 256 + bbase);
          avn = *reinterpret_cast<const f32x4*>(timg + abase + ((p * WN_NR + kz * WN_HY + ky) * WN_PX) * 4);
        }
        if (do_dma) {   // one DMA piece of the next chunk per step, behind the MFMAs
          if (st < WN_XPW) dma_x(st);
          else if (st - WN_XPW < WN_WPW) dma_w(st - WN_XPW, wnext, wnext_dst);
        }
        // A = weights, B = pairs: D[co][pair], a lane owns pair (lane & 31) and channels 8 g + 4 (lane >> 5) + c
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[st & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[r], av[r], acc[st & 3], 0, 0, 0);
        bw = bwn;
        av = avn;
      }
      if (do_dma) {
        __syncthreads();   // own DMAs landed (vmcnt(0)), everyone is done with T and this weight buffer
        transform();       // the next chunk's RAW -> T
        __syncthreads();
      }
      parity ^= 1;
    }

    // ---- output transform + epilogue: bias (+ addend), dwordx4 stores, per-wave GroupNorm partial sums ----
    float s0 = 0.f, s1 = 0.f;
    const int co_lane = cur_cog * 32 + 4 * lh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co = co_lane + 8 * g4;
      if (co < Cout) {   // Cout % 4 == 0 (host-checked)
        f32x4 v0, v1;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float m0 = acc[0][4 * g4 + c], m1 = acc[1][4 * g4 + c], m2 = acc[2][4 * g4 + c], m3 = acc[3][4 * g4 + c];
          v0[c] = ((m0 + m1) + m2) + bv[g4][c];
          v1[c] = ((m1 - m2) - m3) + bv[g4][c];
        }
        if (addend) {
          v0 += ad[0][g4];
          v1 += ad[1][g4];
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          s0 += v0[c] + v1[c];
          s1 += v0[c] * v0[c] + v1[c] * v1[c];
        }
        *reinterpret_cast<f32x4*>(y + (i64)vo0 * Cout + co) = v0;
        *reinterpret_cast<f32x4*>(y + (i64)(vo0 + 1) * Cout + co) = v1;
      }
    }
    if (stats) {
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      if (lane == 0) {
        const int tiles_per_sample = ntz * nty * ntx;
        float* dst = stats + ((((i64)cur_n * tiles_per_sample + cur_tile) * ncog + cur_cog) * WN_NW + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
    if (!more_items) break;
    item = next_item;
  }
}

// shapes this kernel takes: whole 8 x 8 x 8 tiles, channel blocks of 8 / 32
extern "C" int seg3d_conv3d_k3_wino_supported(int N, int D, int H, int W, int Cin, int Cout) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if ((D % WN_TZ) || (H % WN_TY) || (W % WN_TX) || (Cin & 7) || (Cout & 31)) return 0;
  const long long items = (long long)N * (D / WN_TZ) * (H / WN_TY) * (W / WN_TX) * (Cout / 32);
  if (items >= (1 << 20)) return 0;
  if ((long long)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
  return 1;
}

// ... and where it is the faster choice: enough (tile, column block) items to fill the 256 CUs (the spatially small levels
// -- 12^3, 6^3 -- are the split-K kernel's)
extern "C" int seg3d_conv3d_k3_wino_preferred(int N, int D, int H, int W, int Cin, int Cout) {
  if (!seg3d_conv3d_k3_wino_supported(N, D, H, W, Cin, Cout)) return 0;
  return (long long)N * (D / WN_TZ) * (H / WN_TY) * (W / WN_TX) * (Cout / 32) >= 192;
}

// GroupNorm partial (sum, sumsq) slots per sample
extern "C" long long seg3d_conv3d_k3_wino_stats_count(int N, int D, int H, int W, int Cin, int Cout) {
  (void)N; (void)Cin;
  return (long long)(D / WN_TZ) * (H / WN_TY) * (W / WN_TX) * (Cout / 32) * WN_NW;
}

// x [N][D][H][W][Cin], wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 36) (the Winograd image), y [N][D][H][W][Cout];
// bias, addend, stats as seg3d_conv3d_k3_mfma_fwd
extern "C" int seg3d_conv3d_k3_wino_fwd(const float* x, const float* wp, const float* bias, const float* addend, float* y,
                                        float* stats, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_wino_fwd: null pointer");
  SEG3D_REQUIRE(seg3d_conv3d_k3_wino_supported(N, D, H, W, Cin, Cout),
                "seg3d_conv3d_k3_wino_fwd: shape not supported (whole 8^3 tiles, Cin %% 8 == 0, Cout %% 32 == 0)");
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino_kernel), configured, "conv3d_k3_wino")) return rc;
  const int ntz = D / WN_TZ, nty = H / WN_TY, ntx = W / WN_TX, ncog = Cout / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid(seg3d_persistent_grid(nitems), 1, 1);
  hipLaunchKernelGGL(conv3d_k3_wino_kernel, grid, dim3(512), (size_t)WN_LDS_FLOATS * 4, (hipStream_t)stream, x, wp, bias, y,
                     stats, N, D, H, W, Cin, Cout, ntz, nty, ntx, ncog, nitems, addend);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino_fwd");
  return SEG3D_OK;
}

// ================================================================================================================
// Weight gradient of the same layers with Winograd F(3, 2) along x (conv3d_k3_wgrad_wino_kernel).
//   dW[kz][ky][kx][ci][co] = sum_v x[v + (kz, ky, kx) - 1][ci] dy[v][co].  For two output voxels that are neighbours in x
//   (an output PAIR, g0 = dy[x0], g1 = dy[x0 + 1]) and the four inputs d0..d3 = x[x0 - 1 .. x0 + 2] of one (kz, ky) row the
//   three kx taps are a 3-output correlation with a 2-tap filter: 4 multiplies instead of 6,
//       D0 = d0 - d2, D1 = d1 + d2, D2 = d2 - d1, D3 = d1 - d3      (the forward kernel's input transform)
//       E0 = g0,      E1 = g0 + g1, E2 = g0 - g1, E3 = g1           (the 1/2 of E1, E2 is applied once, after the sum)
//       M_p = sum_pairs D_p (x) E_p                                  (four rank-1 updates per pair: the MFMA work)
//       dW[kx=0] = M0 + (M1 + M2) / 2,  dW[1] = (M1 - M2) / 2,  dW[2] = (M1 + M2) / 2 - M3
//   i.e. 36 accumulators [9 (kz, ky)][4 points] of [32 ci][32 co] instead of 27 taps, each fed one voxel PAIR per K slot:
//   2/3 of the MFMAs of conv3d_k3_wgrad2_kernel (and 36 = 4 waves x 9: no idle accumulator slot, where 27 taps on 4 x 7
//   left one).
// Structure = conv3d_k3_wgrad2_kernel (conv_mfma.hip): one persistent workgroup per CU owns a 32 x 32 (ci, co) block pair
//   and one spatial slab of tiles; wave w owns point p = w and keeps its 9 accumulators in registers across all tiles; the
//   next tile's RAW x halo tile and dy tile arrive by LDS-DMA behind the MFMAs; between two tiles all waves transform RAW x
//   once into T [p][halo row][pair][32 ci] (two barriers per tile); E_p is formed from the raw dy pair in registers (one
//   FMA pair per 9 MFMAs).  Partial slabs [slab][pair][36][32][32] are reduced in fixed order, and turned into the three
//   kx taps, by conv3d_k3_wgrad_wino_reduce_kernel (bitwise reproducible).
//   LDS (4 x 4 x 8 tile): RAW x 45 KB + T 72 KB + 2 x 16 KB dy = 149 KB.
// ================================================================================================================
__device__ __forceinline__ int wn_mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int TZ, int TY, int TX>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_wino_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                        float* __restrict__ part, int N, int D, int H, int W,
                                                                        int Cin, int Cout, int ntz, int nty, int ntx, int ntiles,
                                                                        int slabs, int COB32) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HY = TY + 2, HX = TX + 2, PX = TX / 2;
  constexpr int NRH = (TZ + 2) * HY;                   // halo rows
  constexpr int NVH = NRH * HX;                        // halo voxels of the x tile
  constexpr int MTV = TZ * TY * TX;                    // voxels of the dy tile
  constexpr int KS = MTV / 4;                          // K steps per tile: two output pairs (four voxels) each
  constexpr int XPC = NVH / 8, YPC = MTV / 8;          // 1-KiB DMA pieces (8 voxels x 32 channels)
  static_assert(NVH % 8 == 0 && MTV % 8 == 0 && PX % 2 == 0, "tile shape");
  constexpr int XS = NVH * 32;                         // floats of RAW x
  constexpr int TS = 4 * NRH * PX * 32;                // floats of T
  constexpr int YS = MTV * 32;                         // floats of one dy buffer
  constexpr int GX = (XPC + 3) / 4, GY = (YPC + 3) / 4; // piece groups per wave per tile: x groups first, then dy groups
  constexpr int NG = GX + GY;
  static_assert(NG <= KS, "more DMA groups than K steps");
  float* rawx = lds;
  float* timg = lds + XS;
  float* rawy = lds + XS + TS;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = Winograd point p
  const int li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x % slabs;
  const int pg = blockIdx.x / slabs;                   // (ci block, co block)
  const int cib = pg / COB32, cob = pg % COB32;
  const int ci0 = cib * 32, co0 = cob * 32;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };  // exact for the small ranges used here
  // E_p = g0' + e1 g1:  p = 0: g0, 1: g0 + g1, 2: g0 - g1, 3: g1 (g0' read one voxel on, e1 = 0)
  const float e1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
  const int g0off = wave == 3 ? 32 : 0;

  f32x16 acc[9];
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // DMA pieces of this wave: group g < GX is x piece min(wave + 4 g, XPC - 1), group GX + g' is dy piece
  // min(wave + 4 g', YPC - 1) (a slot past the end repeats the last piece: same source, same destination -- no branch);
  // lane -> voxel 8 p + (lane >> 3), channels 4 (lane & 7)..+3.
  // A piece's source is  tile origin (uniform)  +  a tile-invariant per-lane offset, and it is zero padding exactly when its
  // halo face lies outside the volume (whole tiles only, host-checked).
  const int lv = lane >> 3, lq = lane & 7;
  int prel[NG], pflag[NG];  // float offset from the tile-origin voxel; face bits (bit 6: channel slice out of range)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    if (g < GX) {
      const int p = wave + 4 * g < XPC ? wave + 4 * g : XPC - 1;
      const int v = p * 8 + lv;
      const int t = fdiv(v, 1.0f / (float)HX);
      const int hx = v - t * HX;
      const int hz = fdiv(t, 1.0f / (float)HY);
      const int hy = t - hz * HY;
      prel[g] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 4 * lq;
      pflag[g] = (hz == 0 ? 1 : 0) | (hz == TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TY + 1 ? 8 : 0) |
                 (hx == 0 ? 16 : 0) | (hx == TX + 1 ? 32 : 0) | (ci0 + 4 * lq < Cin ? 0 : 64);
    } else {
      const int p = wave + 4 * (g - GX) < YPC ? wave + 4 * (g - GX) : YPC - 1;
      const int v = p * 8 + lv;
      const int co = co0 + 4 * lq;
      prel[g] = (((v / (TX * TY)) * H + (v / TX) % TY) * W + v % TX) * Cout + co;
      pflag[g] = co < Cout ? 0 : 64;
    }
  }
  int tn = 0, tz0 = 0, ty0 = 0, tx0 = 0;  // origin of the tile being fetched
  auto set_tile = [&](int tile) {
    int b = tile;
    int q = fdiv(b, rNTX);
    const int tix = b - q * ntx;
    b = q;
    q = fdiv(b, rNTY);
    const int tiy = b - q * nty;
    b = q;
    q = fdiv(b, rNTZ);
    const int tiz = b - q * ntz;
    tn = q;
    tz0 = tiz * TZ, ty0 = tiy * TY, tx0 = tix * TX;
  };
  auto issue_piece = [&](int g, float* ydst, const float* xbase, const float* ybase, int faces) {
    const float* base = g < GX ? xbase : ybase;            // compile-time choice (g is an unrolled loop index)
    float* dst;
    if (g < GX) dst = rawx + (wave + 4 * g < XPC ? wave + 4 * g : XPC - 1) * 256;
    else dst = ydst + (wave + 4 * (g - GX) < YPC ? wave + 4 * (g - GX) : YPC - 1) * 256;
    const float* src = (pflag[g] & faces) ? wn_zero16 : base + prel[g];
    wn_glds16_asm(src, dst);
  };
  auto transform = [&]() {   // RAW x -> T[p][row][pair][32]
    constexpr int ITEMS = NRH * PX * 8;
#pragma unroll
    for (int r = 0; r < (ITEMS + 255) / 256; ++r) {
      const int i = tid + 256 * r;
      if (i < ITEMS) {
        const int c4 = i & 7, t = i >> 3;
        const int row = t / PX, px = t - row * PX;
        const float* src = rawx + (row * HX + 2 * px) * 32 + 4 * c4;
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(src);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(src + 32);
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(src + 64);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(src + 96);
        float* dst = timg + (row * PX + px) * 32 + 4 * c4;
        *reinterpret_cast<f32x4*>(dst) = d0 - d2;
        *reinterpret_cast<f32x4*>(dst + NRH * PX * 32) = d1 + d2;
        *reinterpret_cast<f32x4*>(dst + 2 * NRH * PX * 32) = d2 - d1;
        *reinterpret_cast<f32x4*>(dst + 3 * NRH * PX * 32) = d1 - d3;
      }
    }
  };
  auto tile_faces = [&]() {
    return 64 | (tz0 == 0 ? 1 : 0) | (tz0 + TZ >= D ? 2 : 0) | (ty0 == 0 ? 4 : 0) | (ty0 + TY >= H ? 8 : 0) |
           (tx0 == 0 ? 16 : 0) | (tx0 + TX >= W ? 32 : 0);
  };

  // tile walk: XCD-contiguous when the slab count allows (as the forward kernels)
  int tile = slab, tstride = slabs, tlimit = ntiles;
  if ((slabs & 7) == 0) {
    const int per_xcd = (ntiles + 7) >> 3, xcd = slab & 7;
    tile = xcd * per_xcd + (slab >> 3);
    tstride = slabs >> 3;
    tlimit = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
  }
  int parity = 0;
  if (tile < tlimit) {
    set_tile(tile);
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    const int faces = tile_faces();
#pragma unroll
    for (int g = 0; g < NG; ++g) issue_piece(g, rawy, x + origin * Cin, dy + origin * Cout, faces);
    wn_dma_wait();
    __syncthreads();
    transform();
    __syncthreads();
  }
  for (; tile < tlimit; tile += tstride) {
    const float* ycur = rawy + parity * YS;
    float* ynxt = rawy + (parity ^ 1) * YS;
    const bool more = tile + tstride < tlimit;
    if (more) set_tile(tile + tstride);
    const int faces = tile_faces();
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    const float* xbase = x + origin * Cin;
    const float* ybase = dy + origin * Cout;
    // K step k: output pairs 2k (lane half 0) and 2k + 1 (half 1), neighbours in one tile row
    const float* ta = timg + wave * (NRH * PX * 32) + lane;       // + ((row + kz HY + ky) PX + px0) 32
    const float* yb = ycur + lh * 64 + li;                        // + (first voxel of pair 2k) 32; g1 one voxel on
    auto aoff = [](int k, int j) {
      const int q0 = 2 * k;
      const int z = q0 / (TY * PX), yy = (q0 / PX) % TY, px0 = q0 % PX;
      return (((z + j / 3) * HY + yy + j % 3) * PX + px0) * 32;
    };
    float a1[9], g1a, g1b;
#pragma unroll
    for (int j = 0; j < 9; ++j) a1[j] = ta[aoff(0, j)];
    g1a = yb[g0off];
    g1b = yb[32];
    // The next tile is fetched behind the first NG K steps, unconditionally: after its last tile a workgroup fetches that
    // tile once more into the idle buffers (no branch in the MFMA loop; the barrier below drains it).
#pragma unroll
    for (int k = 0; k < KS; ++k) {
      float a[9];
#pragma unroll
      for (int j = 0; j < 9; ++j) a[j] = a1[j];
      const float b = fmaf(e1, g1b, g1a);
      if (k + 1 < KS) {   // operands of step k + 1 are read while step k is multiplied
#pragma unroll
        for (int j = 0; j < 9; ++j) a1[j] = ta[aoff(k + 1, j)];
        g1a = yb[(k + 1) * 128 + g0off];
        g1b = yb[(k + 1) * 128 + 32];
      }
      if (k < NG) issue_piece(k, ynxt, xbase, ybase, faces);
      __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from sinking the reads above down to their first use
#pragma unroll
      for (int j = 0; j < 9; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b, acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    wn_dma_wait();       // own DMAs landed
    __syncthreads();     // everyone done with T
    if (more) {
      transform();       // the next tile's RAW x -> T
      __syncthreads();
    }
    parity ^= 1;
  }

  // part[slab][pair = cib * COB32 + cob][(kz, ky) * 4 + p][ci row][co col]
  float* dst = part + ((i64)slab * (COB32 * ((Cin + 31) / 32)) + cib * COB32 + cob) * 36 * 1024;
#pragma unroll
  for (int j = 0; j < 9; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) dst[(j * 4 + wave) * 1024 + wn_mfma_row(r, lh) * 32 + li] = acc[j][r];
}

// dw[a*sa + b*sb + (kz,ky)*3 + kx] from  M_p = sum_slab part[slab][a/32][b/32][(kz,ky)*4 + p][a%32][b%32]:
//   kx 0: M0 + (M1 + M2)/2,  kx 1: (M1 - M2)/2,  kx 2: (M1 + M2)/2 - M3.
// A lane owns four consecutive b of one (pair, (kz, ky), a); the G waves of a workgroup take the slabs k = g, g + G, ..;
// partial sums are combined through LDS in a fixed order.
template <int G>
__global__ __launch_bounds__(64 * G) void conv3d_k3_wgrad_wino_reduce_kernel(const float* __restrict__ part,
                                                                               float* __restrict__ dw, int slabs, int A, int B,
                                                                               int BB32, int npairs, i64 sa, i64 sb,
                                                                               int accumulate) {
  __shared__ f32x4 red[G * 64 * 4];
  const i64 totalq = (i64)npairs * 9 * 256;                     // (pair, (kz, ky), a, b quad)
  const i64 qidx = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  f32x4 s[4];
#pragma unroll
  for (int p = 0; p < 4; ++p) s[p] = f32x4{0.f, 0.f, 0.f, 0.f};
  const i64 slabq = (i64)npairs * 36 * 256;                     // float4 quads per slab
  if (qidx < totalq) {
    const i64 r9 = qidx >> 8;                                   // pair * 9 + (kz, ky)
    const f32x4* p0 = reinterpret_cast<const f32x4*>(part) + r9 * 4 * 256 + (qidx & 255);
    for (int k = g; k < slabs; k += G) {
      const f32x4* q = p0 + (i64)k * slabq;
#pragma unroll
      for (int p = 0; p < 4; ++p) s[p] += q[p * 256];
    }
  }
#pragma unroll
  for (int p = 0; p < 4; ++p) red[threadIdx.x * 4 + p] = s[p];
  __syncthreads();
  if (g == 0 && qidx < totalq) {
#pragma unroll
    for (int j = 1; j < G; ++j)
#pragma unroll
      for (int p = 0; p < 4; ++p) s[p] += red[(j * 64 + threadIdx.x) * 4 + p];
    const f32x4 hs = (s[1] + s[2]) * 0.5f, hd = (s[1] - s[2]) * 0.5f;
    const f32x4 w0 = s[0] + hs, w1 = hd, w2 = hs - s[3];
    const int b32 = (int)((qidx & 7) * 4), a32 = (int)((qidx >> 3) & 31);
    const i64 r9 = qidx >> 8;
    const int t9 = (int)(r9 % 9);
    const int pair = (int)(r9 / 9);
    const int a = (pair / BB32) * 32 + a32, b = (pair % BB32) * 32 + b32;
    if (a < A) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (b + j < B) {
          float* d = dw + a * sa + (b + j) * sb + t9 * 3;
          if (accumulate) {
            d[0] += w0[j];
            d[1] += w1[j];
            d[2] += w2[j];
          } else {
            d[0] = w0[j];
            d[1] = w1[j];
            d[2] = w2[j];
          }
        }
    }
  }
}

struct WnWgradPlan {
  int tx;      // tile 4 x 4 x tx, tx = 8 or 4
  int slabs;
};

static WnWgradPlan wn_wgrad_plan(int N, int D, int H, int W, int Cin, int Cout) {
  WnWgradPlan p;
  p.tx = (W % 8 == 0) ? 8 : 4;
  const i64 ntiles = (i64)N * (D / 4) * (H / 4) * (W / p.tx);
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  i64 slabs = 256 / npairs;  // one resident workgroup per CU over the whole grid
  if (slabs > (ntiles + 1) / 2) slabs = (ntiles + 1) / 2;  // small levels: >= 2 tiles per workgroup
  if (slabs < 1) slabs = 1;
  p.slabs = (int)slabs;
  return p;
}

// shapes the Winograd weight gradient takes: whole 4 x 4 x 4 tiles, channels in fours
extern "C" int seg3d_conv3d_k3_wino_wgrad_supported(int N, int D, int H, int W, int Cin, int Cout) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if ((D % 4) || (H % 4) || (W % 4) || (Cin & 3) || (Cout & 3)) return 0;
  if ((long long)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
  if ((long long)N * (D / 4) * (H / 4) * (W / 4) >= SEG3D_FDIV_MAX) return 0;
  return 1;
}

extern "C" long long seg3d_conv3d_k3_wino_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  return (long long)wn_wgrad_plan(N, D, H, W, Cin, Cout).slabs * npairs * 36 * 1024;
}

template <int TZ, int TY, int TX>
static int wn_launch_wgrad(const float* x, const float* dy, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                           int slabs, hipStream_t s) {
  const int ntz = D / TZ, nty = H / TY, ntx = W / TX;
  const int ntiles = N * ntz * nty * ntx;
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wgrad_wino_kernel<TZ, TY, TX>), configured, "conv3d_k3_wgrad_wino")) return rc;
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  constexpr int NRH = (TZ + 2) * (TY + 2);
  const size_t lds = (size_t)(NRH * (TX + 2) * 32 + 4 * NRH * (TX / 2) * 32 + 2 * TZ * TY * TX * 32) * 4;
  hipLaunchKernelGGL((conv3d_k3_wgrad_wino_kernel<TZ, TY, TX>), dim3((unsigned)(slabs * CIB32 * COB32)), dim3(256), lds, s, x, dy,
                     workspace, N, D, H, W, Cin, Cout, ntz, nty, ntx, ntiles, slabs, COB32);
  return SEG3D_OK;
}

// x [N][D][H][W][Cin], dy [N][D][H][W][Cout]; dw in the reference Conv3d layout [Cout][Cin][3][3][3] (written, or added to
// when accumulate != 0); workspace = seg3d_conv3d_k3_wino_wgrad_workspace_floats floats
extern "C" int seg3d_conv3d_k3_wino_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D, int H,
                                          int W, int Cin, int Cout, int accumulate, void* stream) {
  SEG3D_REQUIRE(x && dy && dw && workspace, "seg3d_conv3d_k3_wino_wgrad: null pointer");
  SEG3D_REQUIRE(seg3d_conv3d_k3_wino_wgrad_supported(N, D, H, W, Cin, Cout),
                "seg3d_conv3d_k3_wino_wgrad: shape not supported (whole 4^3 tiles, Cin %% 4 == 0, Cout %% 4 == 0)");
  const WnWgradPlan plan = wn_wgrad_plan(N, D, H, W, Cin, Cout);
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32, npairs = CIB32 * COB32;
  hipStream_t s = (hipStream_t)stream;
  const int rc = plan.tx == 8 ? wn_launch_wgrad<4, 4, 8>(x, dy, workspace, N, D, H, W, Cin, Cout, plan.slabs, s)
                              : wn_launch_wgrad<4, 4, 4>(x, dy, workspace, N, D, H, W, Cin, Cout, plan.slabs, s);
  if (rc != SEG3D_OK) return rc;
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino_wgrad");
  const i64 totalq = (i64)npairs * 9 * 256;
  const unsigned grid = (unsigned)((totalq + 63) / 64);
  if (plan.slabs >= 32)
    hipLaunchKernelGGL(conv3d_k3_wgrad_wino_reduce_kernel<8>, dim3(grid), dim3(512), 0, s, workspace, dw, plan.slabs, Cin, Cout,
                       COB32, npairs, (i64)27, (i64)Cin * 27, accumulate);
  else
    hipLaunchKernelGGL(conv3d_k3_wgrad_wino_reduce_kernel<4>, dim3(grid), dim3(256), 0, s, workspace, dw, plan.slabs, Cin, Cout,
                       COB32, npairs, (i64)27, (i64)Cin * 27, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino_wgrad(reduce)");
  return SEG3D_OK;
}
