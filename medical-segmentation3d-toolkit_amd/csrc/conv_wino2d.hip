// conv_wino2d.hip -- C -> C nn.Conv3d(k3, p1) forward / data-gradient with Winograd F(2x2, 3x3) over (y, x) on the fp32
// matrix cores (network/module/conv_gn_relu3.py:10; the data-gradient is the same kernel on tap-flipped, transposed
// weights).
//
// conv_wino.hip removes a third of the multiplies with F(2, 3) along x.  Nesting the same transform along y computes a
// 2 x 2 output QUAD of one z plane from a 4 x 4 input patch with 16 multiplies instead of 36 per kz:
//     V = B^T d B      d = in[y0 - 1 .. y0 + 2][x0 - 1 .. x0 + 2],   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//     U = G g G^T      g = the (ky, kx) taps of one kz,               G   = [1 0 0; 1/2 1/2 1/2; 1/2 -1/2 1/2; 0 0 1]
//     M = sum_{kz, ci} U (.) V   (16 points)                          Y = A^T M A,  A^T = [1 1 1 0; 0 1 -1 -1]
// i.e. sixteen implicit GEMMs with K = 3 Cin per output quad: 4/9 of the direct kernel's MFMAs (2/3 of conv_wino.hip's).
// All coefficients are 1, 1/2, 1/4: the transforms are exact in fp32 up to one rounding per add, the accumulation is the
// same fp32 MFMA chain (error against float64: tests/test_gpu_kernels.py::test_conv3d_k3_winograd2d).
//
// Structure: the persistent skeleton of conv_wino.hip (one workgroup per CU walks (tile, column block) items; K chunks
// arrive by LDS-DMA; packed weight images are straight copies), with
//   * tile 8 x 8 x 8 voxels = 128 output quads; FOUR waves (one per SIMD, 512 registers): wave w owns the z planes 2w,
//     2w + 1 = 32 quads x 32 output channels, i.e. 16 point accumulators of 16 registers;
//   * K chunks of FOUR input channels (the weight image of a chunk is 48 x 512 B = 24 KB, two buffers; chunks of 8 would
//     need 210 KB): per chunk the RAW halo tile [10 x 10 x 10 voxels][4] lands by DMA while the previous chunk is
//     multiplied, then 160 threads transform it into T[p = 16][z = 10][quad = 16][4] (two barriers per chunk);
//   * 48 steps of 2 MFMAs per chunk: (kz, point), operands one ds_read_b64 each (a wave's 32 quads of one (p, z pair) are
//     512 contiguous bytes of T; the weight image is [t][co][4]);
//   * the weight image is the T = 48 pack of seg3d_pack_weights_mfma: t = kz * 16 + py * 4 + px, laid out per 8-channel chunk
//     [t][half][32 co][4] -- a 4-channel K chunk is one half;
//   * LDS: RAW 16 KB + T 40 KB + 2 x 24 KB weights = 104 KB.
#include "seg3d_common.h"
#include "seg3d_hip.h"
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define W2_TS 8                                   // tile edge
#define W2_H (W2_TS + 2)                          // halo edge: 10
#define W2_NV (W2_H * W2_H * W2_H)                // 1000 halo voxels
#define W2_NQ 16                                  // quads per z plane (4 x 4)
#define W2_RAW 4096                               // floats: [NV][4] padded to whole 1-KiB DMA pieces (16)
#define W2_T (16 * W2_H * W2_NQ * 4)              // floats of the transformed image: 10240
#define W2_W (48 * 128)                           // floats of one K chunk's weight image: [48][32][4]
#define W2_NW 4
#define W2_LDS_FLOATS (W2_RAW + W2_T + 2 * W2_W)  // 26624 floats = 106496 bytes
#define W2_XPW 4                                  // raw pieces per wave (16 / 4)
#define W2_WPW 6                                  // weight pieces per wave (24 / 4)

__device__ __attribute__((aligned(16))) float w2_zero16[4];   // DMA source for zero padding

// LDS-DMA of 16 bytes per lane, written as inline assembly ON PURPOSE: while a __builtin_amdgcn_global_load_lds is
// outstanding hipcc's wait-count pass treats it as a pending FLAT access and turns every LDS wait of the loop into
// lgkmcnt(0) (tools/ubench/waitcnt_dma.hip) -- a full drain of the operand reads issued for the following steps, which one
// wave per SIMD cannot hide.  The compiler does not see this load, so the kernel waits for it itself (w2_dma_wait) before
// the barrier that publishes the data.  Nothing else in this kernel uses M0.
typedef __attribute__((address_space(3))) float w2_lds_float;
__device__ __forceinline__ void w2_glds16(const float* src, float* lds_dst_wave_uniform) {
  const unsigned off = (unsigned)(uintptr_t)(w2_lds_float*)lds_dst_wave_uniform;
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(off) : "memory");
}
__device__ __forceinline__ void w2_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

__global__ __launch_bounds__(256, 1) void conv3d_k3_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   float* __restrict__ stats, int N, int D, int H, int W, int Cin,
                                                                   int Cout, int ntz, int nty, int ntx, int ncog, int nitems,
                                                                   const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                       // [NV][4] (+ padding)
  float* timg = lds + W2_RAW;             // [16][10][16][4]
  float* wbuf = lds + W2_RAW + W2_T;      // [2][48][32][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int NSC = Cin >> 2;               // K chunks of 4 channels
  const int AB = Cin >> 3;                // packed 8-channel chunks per column block
  const int G = gridDim.x;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz, rNCOG = 1.0f / (float)ncog;

  // ---- per-lane constants ----
  // raw DMA pieces of this wave: piece p = wave + 4 j covers halo voxels e = 64 p + lane;
  // hpos[j] = halo coordinates (hz << 20 | hy << 10 | hx), -1 = padding entry
  int hpos[W2_XPW];
#pragma unroll
  for (int j = 0; j < W2_XPW; ++j) {
    const int e = (wave + W2_NW * j) * 64 + lane;
    hpos[j] = -1;
    if (e < W2_NV) {
      const int t = e / W2_H;
      const int hx = e - t * W2_H;
      const int hz = t / W2_H;
      const int hy = t - hz * W2_H;
      hpos[j] = (hz << 20) | (hy << 10) | hx;
    }
  }
  // this lane's output quad: planes 2 wave + zz, quad (qy, qx)
  const int zz = li >> 4, qq = li & 15, qy = qq >> 2, qx = qq & 3;
  const int abase = li * 4 + 2 * lh;                                      // weights: [t][co = li][4], channels 2 lh, 2 lh + 1
  const int bbase = ((2 * wave + zz) * W2_NQ + qq) * 4 + 2 * lh;          // T: + (p * 10 + kz) * 64
  // transform item of this thread (tid < 160): (z, quad)
  const int t_z = tid >> 4, t_q = tid & 15;
  const int t_src = ((t_z * W2_H + 2 * (t_q >> 2)) * W2_H + 2 * (t_q & 3)) * 4;
  const int t_dst = (t_z * W2_NQ + t_q) * 4;

  // ---- work item state ----
  int it_n = 0, it_z0 = 0, it_y0 = 0, it_x0 = 0, it_cog = 0, it_tile = 0;
  const float* xsrc[W2_XPW];
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
    it_z0 = tiz * W2_TS, it_y0 = tiy * W2_TS, it_x0 = tix * W2_TS;
    xadv = 0;
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) {
      xsrc[j] = w2_zero16;
      const int hp = hpos[j];
      const int gz = it_z0 + ((hp >> 20) & 1023) - 1, gy = it_y0 + ((hp >> 10) & 1023) - 1, gx = it_x0 + (hp & 1023) - 1;
      if (hp >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        xsrc[j] = x + (i64)(((it_n * D + gz) * H + gy) * W + gx) * Cin;
        xadv |= 1 << j;
      }
    }
  };
  auto dma_x = [&](int j) {  // issues the piece into RAW, then steps its source to the next K chunk
    w2_glds16(xsrc[j], raw + (wave + W2_NW * j) * 256);
    xsrc[j] += ((xadv >> j) & 1) * 4;
  };
  // weight image of K chunk sc of column block cog: half (sc & 1) of the packed 8-channel chunk sc >> 1
  auto wsrc_of = [&](int cog, int sc) { return wp + ((i64)cog * AB + (sc >> 1)) * (48 * 256) + (sc & 1) * 128; };
  auto dma_w = [&](int j, const float* wsrc, float* wdst) {
    const int piece = wave + W2_NW * j;   // two images per piece
    w2_glds16(wsrc + (2 * piece + (lane >> 5)) * 256 + (lane & 31) * 4, wdst + piece * 256);
  };
  auto transform = [&]() {   // RAW -> T: V = B^T d B per (z, quad), 16 points
    if (tid < W2_H * W2_NQ) {
      f32x4 dx[4][4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* s = raw + t_src + r * (W2_H * 4);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(s);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(s + 4);
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(s + 8);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(s + 12);
        dx[r][0] = d0 - d2;
        dx[r][1] = d1 + d2;
        dx[r][2] = d2 - d1;
        dx[r][3] = d1 - d3;
      }
      float* dst = timg + t_dst;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        *reinterpret_cast<f32x4*>(dst + (0 * 4 + px) * (W2_H * W2_NQ * 4)) = dx[0][px] - dx[2][px];
        *reinterpret_cast<f32x4*>(dst + (1 * 4 + px) * (W2_H * W2_NQ * 4)) = dx[1][px] + dx[2][px];
        *reinterpret_cast<f32x4*>(dst + (2 * 4 + px) * (W2_H * W2_NQ * 4)) = dx[2][px] - dx[1][px];
        *reinterpret_cast<f32x4*>(dst + (3 * 4 + px) * (W2_H * W2_NQ * 4)) = dx[1][px] - dx[3][px];
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
  {  // the only exposed DMA prologue of this workgroup: K chunk 0 of its first item
    const float* w0 = wsrc_of(it_cog, 0);
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j);
#pragma unroll
    for (int j = 0; j < W2_WPW; ++j) dma_w(j, w0, wbuf);
  }
  w2_dma_wait();
  __syncthreads();   // publishes RAW / weights
  transform();
  __syncthreads();

  int parity = 0;
  for (;;) {
    const int cur_n = it_n, cur_z0 = it_z0, cur_y0 = it_y0, cur_x0 = it_x0, cur_cog = it_cog, cur_tile = it_tile;
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;

    f32x16 acc[16];
#pragma unroll
    for (int p = 0; p < 16; ++p)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[p][r] = 0.f;
    for (int sc = 0; sc < NSC; ++sc) {
      const float* ws = wbuf + parity * W2_W;
      float* wnext_dst = wbuf + (parity ^ 1) * W2_W;
      const bool last = sc + 1 == NSC;
      const float* wnext;
      if (last) {
        // DMA sources now belong to the next item; after the very last item they are reset to this one's chunk 0 (a
        // harmless refetch into the idle buffers, so that the unrolled loop below needs no branch)
        setup_item(more_items ? next_item : item);
        wnext = wsrc_of(it_cog, 0);
      } else {
        wnext = wsrc_of(cur_cog, sc + 1);
      }
      // operands of step st + 2 are read while step st is multiplied (one wave per SIMD: nothing else hides the LDS latency);
      // the scheduling barriers keep hipcc from sinking the reads down to their first use
      auto lda = [&](int s1) { return *reinterpret_cast<const f32x2*>(ws + s1 * 128 + abase); };
      auto ldb = [&](int s1) {
        return *reinterpret_cast<const f32x2*>(timg + ((s1 & 15) * W2_H + (s1 >> 4)) * (W2_NQ * 4) + bbase);
      };
      f32x2 aw0 = lda(0), bv0 = ldb(0), aw1 = lda(1), bv1 = ldb(1);
#pragma unroll
      for (int st = 0; st < 48; ++st) {
        f32x2 aw2 = aw1, bv2 = bv1;
        if (st + 2 < 48) {
          aw2 = lda(st + 2);
          bv2 = ldb(st + 2);
        }
        // the next K chunk (of this item, or chunk 0 of the next item; after the very last chunk: a harmless refetch)
        // arrives behind the first ten steps -- unconditionally, so that the unrolled loop has no branch
        if (st < W2_XPW) dma_x(st);
        else if (st - W2_XPW < W2_WPW) dma_w(st - W2_XPW, wnext, wnext_dst);
        __builtin_amdgcn_sched_barrier(0);
        // A = weights, B = quads: D[co][quad], a lane owns quad (lane & 31) and channels 8 g + 4 (lane >> 5) + c
        acc[st & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw0[0], bv0[0], acc[st & 15], 0, 0, 0);
        acc[st & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(aw0[1], bv0[1], acc[st & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        aw0 = aw1;
        bv0 = bv1;
        aw1 = aw2;
        bv1 = bv2;
      }
      w2_dma_wait();     // own DMAs landed
      __syncthreads();   // everyone is done with T and this weight buffer
      if (!last || more_items) {
        transform();     // the next chunk's RAW -> T
        __syncthreads();
      }
      parity ^= 1;
    }

    // ---- output transform Y = A^T M A + epilogue: bias (+ addend), dwordx4 stores, per-wave GroupNorm partial sums ----
    float s0 = 0.f, s1 = 0.f;
    const int co_lane = cur_cog * 32 + 4 * lh;
    // the quad's first voxel (whole tiles only, host-checked: always inside the volume)
    const int vo00 = ((cur_n * D + cur_z0 + 2 * wave + zz) * H + cur_y0 + 2 * qy) * W + cur_x0 + 2 * qx;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co = co_lane + 8 * g4;
      if (co < Cout) {   // Cout % 4 == 0 (host-checked)
        const f32x4 bvv = *reinterpret_cast<const f32x4*>(bias ? bias + co : w2_zero16);
        f32x4 v[2][2];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int r = 4 * g4 + c;
          float r0[4], r1[4];
#pragma unroll
          for (int px = 0; px < 4; ++px) {
            const float m0 = acc[px][r], m1 = acc[4 + px][r], m2 = acc[8 + px][r], m3 = acc[12 + px][r];
            r0[px] = (m0 + m1) + m2;
            r1[px] = (m1 - m2) - m3;
          }
          v[0][0][c] = ((r0[0] + r0[1]) + r0[2]) + bvv[c];
          v[0][1][c] = ((r0[1] - r0[2]) - r0[3]) + bvv[c];
          v[1][0][c] = ((r1[0] + r1[1]) + r1[2]) + bvv[c];
          v[1][1][c] = ((r1[1] - r1[2]) - r1[3]) + bvv[c];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const i64 off = (i64)(vo00 + i * W + j) * Cout + co;
            if (addend) v[i][j] += *reinterpret_cast<const f32x4*>(addend + off);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              s0 += v[i][j][c];
              s1 += v[i][j][c] * v[i][j][c];
            }
            *reinterpret_cast<f32x4*>(y + off) = v[i][j];
          }
      }
    }
    if (stats) {
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      if (lane == 0) {
        const int tiles_per_sample = ntz * nty * ntx;
        float* dst = stats + ((((i64)cur_n * tiles_per_sample + cur_tile) * ncog + cur_cog) * W2_NW + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
    if (!more_items) break;
    item = next_item;
  }
}

// shapes this kernel takes: whole 8 x 8 x 8 tiles, channel blocks of 8 / 32
extern "C" int seg3d_conv3d_k3_wino2d_supported(int N, int D, int H, int W, int Cin, int Cout) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if ((D % W2_TS) || (H % W2_TS) || (W % W2_TS) || (Cin & 7) || (Cout & 31)) return 0;
  const long long items = (long long)N * (D / W2_TS) * (H / W2_TS) * (W / W2_TS) * (Cout / 32);
  if (items >= (1 << 20)) return 0;
  if ((long long)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
  return 1;
}

// ... and where it is the faster choice: enough (tile, column block) items to fill the 256 CUs
extern "C" int seg3d_conv3d_k3_wino2d_preferred(int N, int D, int H, int W, int Cin, int Cout) {
  if (!seg3d_conv3d_k3_wino2d_supported(N, D, H, W, Cin, Cout)) return 0;
  return (long long)N * (D / W2_TS) * (H / W2_TS) * (W / W2_TS) * (Cout / 32) >= 192;
}

// GroupNorm partial (sum, sumsq) slots per sample
extern "C" long long seg3d_conv3d_k3_wino2d_stats_count(int N, int D, int H, int W, int Cin, int Cout) {
  (void)N; (void)Cin;
  return (long long)(D / W2_TS) * (H / W2_TS) * (W / W2_TS) * (Cout / 32) * W2_NW;
}

// x [N][D][H][W][Cin], wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 48) (the F(2x2, 3x3) image), y [N][D][H][W][Cout];
// bias, addend, stats as seg3d_conv3d_k3_mfma_fwd
extern "C" int seg3d_conv3d_k3_wino2d_fwd(const float* x, const float* wp, const float* bias, const float* addend, float* y,
                                          float* stats, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_wino2d_fwd: null pointer");
  SEG3D_REQUIRE(seg3d_conv3d_k3_wino2d_supported(N, D, H, W, Cin, Cout),
                "seg3d_conv3d_k3_wino2d_fwd: shape not supported (whole 8^3 tiles, Cin %% 8 == 0, Cout %% 32 == 0)");
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
    if (e != hipSuccess) {
      seg3d_set_error("conv3d_k3_wino2d: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
      return SEG3D_ERR_LAUNCH;
    }
    configured = true;
  }
  const int ntz = D / W2_TS, nty = H / W2_TS, ntx = W / W2_TS, ncog = Cout / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid((unsigned)(nitems < 256 ? nitems : 256), 1, 1);
  hipLaunchKernelGGL(conv3d_k3_wino2d_kernel, grid, dim3(256), (size_t)W2_LDS_FLOATS * 4, (hipStream_t)stream, x, wp, bias, y,
                     stats, N, D, H, W, Cin, Cout, ntz, nty, ntx, ncog, nitems, addend);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_fwd");
  return SEG3D_OK;
}
