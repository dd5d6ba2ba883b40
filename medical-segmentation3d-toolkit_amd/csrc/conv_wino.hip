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
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wino_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) {
      seg3d_set_error("conv3d_k3_wino: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SEG3D_ERR_LAUNCH;
    }
    configured = true;
  }
  const int ntz = D / WN_TZ, nty = H / WN_TY, ntx = W / WN_TX, ncog = Cout / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid((unsigned)(nitems < 256 ? nitems : 256), 1, 1);
  hipLaunchKernelGGL(conv3d_k3_wino_kernel, grid, dim3(512), (size_t)WN_LDS_FLOATS * 4, (hipStream_t)stream, x, wp, bias, y,
                     stats, N, D, H, W, Cin, Cout, ntz, nty, ntx, ncog, nitems, addend);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino_fwd");
  return SEG3D_OK;
}
