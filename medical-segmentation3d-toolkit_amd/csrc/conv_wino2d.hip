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
//   * K chunks of FOUR input channels (the weight image of a chunk is 48 x 512 B = 24 KB; chunks of 8 would not fit);
//   * everything a chunk needs arrives or is made INSIDE the previous chunk's MFMA loop, one barrier per chunk: while chunk c
//     is multiplied, the weights of chunk c + 1 and the RAW halo tile [10 x 10 x 10 voxels][4] of chunk c + 2 land by
//     LDS-DMA (steps 0-9), and the RAW tile of chunk c + 1 is transformed into the second T buffer
//     T[p = 16][z = 10][quad = 16][4] in eight stages spread over steps 12-33 (40 (z, quad) items per wave, packed fp32
//     adds: 64 vector instructions per wave and chunk beside 96 MFMAs).  The first version transformed between two barriers
//     with the matrix cores idle: 1 800 of 12 000 cycles per chunk (tools/ubench/wino2d_stamp.hip);
//   * 48 steps of 2 MFMAs per chunk: (kz, point), operands one ds_read_b64 each, read two steps ahead (a wave's 32 quads
//     of one (p, z pair) are 512 contiguous bytes of T; the weight image is [t][co][4]);
//   * the weight image is the T = 48 pack of seg3d_pack_weights_mfma: t = kz * 16 + py * 4 + px, laid out per 8-channel chunk
//     [t][half][32 co][4] -- a 4-channel K chunk is one half;
//   * LDS: 2 x 16 KB RAW + 2 x 40 KB T + 2 x 24 KB weights = 160 KB, all of it.
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
#define W2_LDS_FLOATS (2 * W2_RAW + 2 * W2_T + 2 * W2_W)  // 40960 floats = 163840 bytes: all of the CU's LDS
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
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(off) : "memory", "m0");
}
__device__ __forceinline__ void w2_dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// a - b on two packed floats in one instruction (hipcc selects v_pk_add_f32 for additions but two v_sub_f32 for this)
__device__ __forceinline__ f32x2 w2_pk_add(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
  return d;
}
__device__ __forceinline__ f32x2 w2_pk_sub(f32x2 a, f32x2 b) {
  f32x2 d;
  asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
  return d;
}

#ifdef W2_STAMPS   // diagnostic build only (tools/ubench/wino2d_stamp.hip): s_memtime stamps of the first chunks of a few workgroups
__device__ long long* w2_stamp_buf;
#define W2_STAMP(chunk, k)                                                                                        \
  do {                                                                                                            \
    if (blockIdx.x < 8 && (chunk) < 64 && (threadIdx.x & 63) == 0)                                                \
      w2_stamp_buf[((blockIdx.x * 4 + (threadIdx.x >> 6)) * 64 + (chunk)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define W2_STAMP(chunk, k)
#endif

// BIAS / ADD: is there a bias (forward launches) / a fused addend (data-gradient launches that take the residual-path gradient)?
// Compile-time, not run-time: the epilogue must not contain a load it does not need -- on gfx9 the wait for a load issued
// after a group's stores also waits for those stores (one in-order vmcnt); without any load the epilogue of an item takes
// 5 700 instead of 7 900 cycles (tools/ubench/wino2d_stamp.hip, mode 1).  The bias therefore enters at accumulator
// initialisation (below), not in the epilogue.
template <bool BIAS, bool ADD>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                   const float* __restrict__ bias, float* __restrict__ y,
                                                                   float* __restrict__ stats, int N, int D, int H, int W, int Cin,
                                                                   int Cout, int ntz, int nty, int ntx, int ncog, int nitems,
                                                                   const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                               // [2][NV][4] (+ padding)
  float* timg = lds + 2 * W2_RAW;                 // [2][16][10][16][4]
  float* wbuf = lds + 2 * W2_RAW + 2 * W2_T;      // [2][48][32][4]
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
  // transform item of this lane: 160 (z, quad) items, 40 per wave (lanes >= 40 repeat item 39 of their wave: same values to
  // the same addresses -- no branch)
  const int t_i = wave * 40 + (lane < 40 ? lane : 39);
  const int t_z = t_i >> 4, t_q = t_i & 15;
  const int t_src = ((t_z * W2_H + 2 * (t_q >> 2)) * W2_H + 2 * (t_q & 3)) * 4;
  const int t_dst = (t_z * W2_NQ + t_q) * 4;

  auto decode = [&](int item, int& n, int& z0, int& y0, int& x0, int& cog, int& tile) {
    const int tile_all = fdiv(item, rNCOG);
    cog = item - tile_all * ncog;
    int b = tile_all;
    int q = fdiv(b, rNTX);
    const int tix = b - q * ntx;
    b = q;
    q = fdiv(b, rNTY);
    const int tiy = b - q * nty;
    b = q;
    q = fdiv(b, rNTZ);
    const int tiz = b - q * ntz;
    n = q;
    tile = (tiz * nty + tiy) * ntx + tix;
    z0 = tiz * W2_TS, y0 = tiy * W2_TS, x0 = tix * W2_TS;
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

  // ---- fetch cursors ----
  // While K chunk c is multiplied, the weights of chunk c + 1 and the RAW tile of chunk c + 2 arrive by DMA and the RAW tile
  // of chunk c + 1 (landed during chunk c - 1) is transformed into the second T buffer -- all inside the MFMA loop, one
  // barrier per chunk.  Each stream has its own cursor over the (item, chunk) sequence of this workgroup; past the last
  // chunk of the last item a cursor wraps to that item's chunk 0 (harmless refetch: the loop stays uniform, no branch).
  int fx_item = item, fx_sc = 0;   // RAW cursor
  const float* xsrc[W2_XPW];
  int xadv = 0;
  auto fx_setup = [&](int it) {   // DMA sources of chunk 0 of item `it`
    int n, z0, y0, x0, cog, tile;
    decode(it, n, z0, y0, x0, cog, tile);
    xadv = 0;
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) {
      xsrc[j] = w2_zero16;
      const int hp = hpos[j];
      const int gz = z0 + ((hp >> 20) & 1023) - 1, gy = y0 + ((hp >> 10) & 1023) - 1, gx = x0 + (hp & 1023) - 1;
      if (hp >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        xsrc[j] = x + (i64)(((n * D + gz) * H + gy) * W + gx) * Cin;
        xadv |= 1 << j;
      }
    }
  };
  auto fx_advance = [&]() {
    ++fx_sc;
    if (fx_sc == NSC) {
      fx_sc = 0;
      if (fx_item + istride < ilimit) fx_item += istride;
      fx_setup(fx_item);
    }
  };
  auto dma_x = [&](int j, float* rdst) {  // issues the piece into RAW, then steps its source to the next K chunk
    w2_glds16(xsrc[j], rdst + (wave + W2_NW * j) * 256);
    xsrc[j] += ((xadv >> j) & 1) * 4;
  };
  int fw_item = item, fw_sc = 0;   // weight cursor
  // weight image of K chunk sc of column block cog: half (sc & 1) of the packed 8-channel chunk sc >> 1
  auto fw_src = [&]() {
    const int cog = fw_item - fdiv(fw_item, rNCOG) * ncog;
    return wp + ((i64)cog * AB + (fw_sc >> 1)) * (48 * 256) + (fw_sc & 1) * 128;
  };
  auto fw_advance = [&]() {
    ++fw_sc;
    if (fw_sc == NSC) {
      fw_sc = 0;
      if (fw_item + istride < ilimit) fw_item += istride;
    }
  };
  auto dma_w = [&](int j, const float* wsrc, float* wdst) {
    const int piece = wave + W2_NW * j;   // two images per piece
    w2_glds16(wsrc + (2 * piece + (lane >> 5)) * 256 + (lane & 31) * 4, wdst + piece * 256);
  };
  // RAW -> T: V = B^T d B per (z, quad), 16 points, in stages that the MFMA loop interleaves: four row stages (read one row
  // of the 4 x 4 patch, x pass) and four column stages (y pass of one px, four stores); packed fp32 math
  f32x4 rd[2][4];
  f32x2 dxl[4][4], dxh[4][4];   // [row][px], channel pairs (0, 1) and (2, 3)
  auto tr_read = [&](const float* rw, int r) {
    const float* sp = rw + t_src + r * (W2_H * 4);
#pragma unroll
    for (int k = 0; k < 4; ++k) rd[r & 1][k] = *reinterpret_cast<const f32x4*>(sp + 4 * k);
  };
  auto tr_x = [&](int r) {
    const f32x4* d = rd[r & 1];
    const f32x2 d0l = {d[0][0], d[0][1]}, d0h = {d[0][2], d[0][3]}, d1l = {d[1][0], d[1][1]}, d1h = {d[1][2], d[1][3]};
    const f32x2 d2l = {d[2][0], d[2][1]}, d2h = {d[2][2], d[2][3]}, d3l = {d[3][0], d[3][1]}, d3h = {d[3][2], d[3][3]};
    dxl[r][0] = w2_pk_sub(d0l, d2l), dxh[r][0] = w2_pk_sub(d0h, d2h);
    dxl[r][1] = w2_pk_add(d1l, d2l), dxh[r][1] = w2_pk_add(d1h, d2h);
    dxl[r][2] = w2_pk_sub(d2l, d1l), dxh[r][2] = w2_pk_sub(d2h, d1h);
    dxl[r][3] = w2_pk_sub(d1l, d3l), dxh[r][3] = w2_pk_sub(d1h, d3h);
  };
  auto tr_y = [&](float* tdst, int px) {
    float* dst = tdst + t_dst + px * (W2_H * W2_NQ * 4);
    const f32x2 v0l = w2_pk_sub(dxl[0][px], dxl[2][px]), v0h = w2_pk_sub(dxh[0][px], dxh[2][px]);
    const f32x2 v1l = w2_pk_add(dxl[1][px], dxl[2][px]), v1h = w2_pk_add(dxh[1][px], dxh[2][px]);
    const f32x2 v2l = w2_pk_sub(dxl[2][px], dxl[1][px]), v2h = w2_pk_sub(dxh[2][px], dxh[1][px]);
    const f32x2 v3l = w2_pk_sub(dxl[1][px], dxl[3][px]), v3h = w2_pk_sub(dxh[1][px], dxh[3][px]);
    *reinterpret_cast<f32x4*>(dst + 0 * 4 * (W2_H * W2_NQ * 4)) = f32x4{v0l[0], v0l[1], v0h[0], v0h[1]};
    *reinterpret_cast<f32x4*>(dst + 1 * 4 * (W2_H * W2_NQ * 4)) = f32x4{v1l[0], v1l[1], v1h[0], v1h[1]};
    *reinterpret_cast<f32x4*>(dst + 2 * 4 * (W2_H * W2_NQ * 4)) = f32x4{v2l[0], v2l[1], v2h[0], v2h[1]};
    *reinterpret_cast<f32x4*>(dst + 3 * 4 * (W2_H * W2_NQ * 4)) = f32x4{v3l[0], v3l[1], v3h[0], v3h[1]};
  };

  fx_setup(item);
  {  // the exposed prologue of this workgroup: RAW of its chunks 0 and 1, weights of chunk 0; chunk 0 transformed stand-alone
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, raw);
    fx_advance();
    const float* w0 = fw_src();
#pragma unroll
    for (int j = 0; j < W2_WPW; ++j) dma_w(j, w0, wbuf);
    fw_advance();
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, raw + W2_RAW);
    fx_advance();
  }
  w2_dma_wait();
  __syncthreads();   // publishes RAW / weights
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    tr_read(raw, r);
    tr_x(r);
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) tr_y(timg, px);
  __syncthreads();

  int ci_ = 0;   // parity of the current chunk: T, W buffers ci_; RAW of the next chunk in raw[ci_ ^ 1]
#ifdef W2_STAMPS
  int gchunk = 0;
#endif
  for (;;) {
    int cur_n, cur_z0, cur_y0, cur_x0, cur_cog, cur_tile;
    decode(item, cur_n, cur_z0, cur_y0, cur_x0, cur_cog, cur_tile);
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;

    // The bias enters through the accumulator of point (py, px) = (1, 1): the output transform Y = A^T M A adds M[1][1] to each of
    // the quad's four outputs once (r0[1] and r1[1] both contain it with weight +1, and each output takes row entry 1 with
    // weight +1), so starting that accumulator at b instead of 0 yields + b on every output.  The lane's 16 channels (8 g + 4 lh
    // + c) come through SCALAR loads of the column block's 32 biases (wave-uniform address, selected per lane half): loaded
    // in the epilogue instead, each group's vector load followed the previous group's stores, and on gfx9 the wait for a load
    // also waits for every store issued before it -- 7 200 cycles per item against 5 700 for an item without bias
    // (tools/ubench/wino2d_stamp.hip).  The scalar loads count on lgkmcnt and land behind the 240 register writes below.
    f32x16 acc[16];
    {
      float binit[16];
      if (BIAS) {
        const int cb = __builtin_amdgcn_readfirstlane(cur_cog * 32);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float b0 = bias[cb + 8 * (r >> 2) + (r & 3)], b1 = bias[cb + 8 * (r >> 2) + 4 + (r & 3)];
          binit[r] = lh ? b1 : b0;
        }
      }
#pragma unroll
      for (int p = 0; p < 16; ++p)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[p][r] = (BIAS && p == 5) ? binit[r] : 0.f;
    }
    for (int sc = 0; sc < NSC; ++sc) {
      const float* ws = wbuf + ci_ * W2_W;
      const float* tcur = timg + ci_ * W2_T;
      float* wdst1 = wbuf + (ci_ ^ 1) * W2_W;           // weights of the next chunk
      float* tdst1 = timg + (ci_ ^ 1) * W2_T;           // T of the next chunk
      const float* rsrc1 = raw + (ci_ ^ 1) * W2_RAW;    // RAW of the next chunk (landed)
      float* rdst2 = raw + ci_ * W2_RAW;                // RAW of the chunk after next (RAW of this chunk is dead)
      const float* wsrc1 = fw_src();
      // operands of step st + 2 are read while step st is multiplied (one wave per SIMD: nothing else hides the LDS latency);
      // the scheduling barriers keep hipcc from moving the reads, the DMA issue and the transform stages
      auto lda = [&](int s1) { return *reinterpret_cast<const f32x2*>(ws + s1 * 128 + abase); };
      auto ldb = [&](int s1) {
        return *reinterpret_cast<const f32x2*>(tcur + ((s1 & 15) * W2_H + (s1 >> 4)) * (W2_NQ * 4) + bbase);
      };
      W2_STAMP(gchunk, 0);
      f32x2 aw0 = lda(0), bv0 = ldb(0), aw1 = lda(1), bv1 = ldb(1);
#pragma unroll
      for (int st = 0; st < 48; ++st) {
        f32x2 aw2 = aw1, bv2 = bv1;
        if (st + 2 < 48) {
          aw2 = lda(st + 2);
          bv2 = ldb(st + 2);
        }
        // steps 0-9: DMA issue (weights first, they are needed first); steps 12-21: row stages; 24-33: column stages
#ifndef W2_EXP_NODMA   // (diagnostic builds of tools/ubench/wino2d_stamp.hip drop one ingredient of the loop to price it)
        if (st < W2_WPW) dma_w(st, wsrc1, wdst1);
        else if (st - W2_WPW < W2_XPW) dma_x(st - W2_WPW, rdst2);
#endif
#ifndef W2_EXP_NOTR
        if (st >= 12 && st <= 18 && (st & 1) == 0) tr_read(rsrc1, (st - 12) >> 1);
        if (st >= 15 && st <= 21 && (st & 1) == 1) tr_x((st - 15) >> 1);
        if (st >= 24 && st <= 33 && (st - 24) % 3 == 0) tr_y(tdst1, (st - 24) / 3);
#endif
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
      W2_STAMP(gchunk, 1);
      fw_advance();
      fx_advance();
      w2_dma_wait();     // own DMAs landed
      W2_STAMP(gchunk, 2);
      __syncthreads();   // next T complete, everyone is done with this T / weight buffer
      W2_STAMP(gchunk, 3);
      ci_ ^= 1;
#ifdef W2_STAMPS
      ++gchunk;
#endif
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
        // fused addend: the group's four quads are requested BEFORE its accumulators are read and transformed, so their
        // latency -- and the drain of the previous group's stores that the wait implies -- passes behind ~180 vector
        // instructions.  Loaded next to each store (first version) the epilogue was 16 store + load round trips: 22 000 cycles
        // per item against 13 000 now (mode 2 of the stamp harness; 96^3 32 -> 32: 177 -> 191 TFLOP/s algorithmic).  Fetching a
        // group ahead of the previous group's stores needs 16 more registers and spills (54): measured slower.
        f32x4 adq[4];
        if (ADD) {
#pragma unroll
          for (int k = 0; k < 4; ++k)
            adq[k] = *reinterpret_cast<const f32x4*>(addend + (i64)(vo00 + (k >> 1) * W + (k & 1)) * Cout + co);
        }
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
          v[0][0][c] = (r0[0] + r0[1]) + r0[2];
          v[0][1][c] = (r0[1] - r0[2]) - r0[3];
          v[1][0][c] = (r1[0] + r1[1]) + r1[2];
          v[1][1][c] = (r1[1] - r1[2]) - r1[3];
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const i64 off = (i64)(vo00 + i * W + j) * Cout + co;
            if (ADD) v[i][j] += adq[2 * i + j];
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
    W2_STAMP(gchunk - 1, 6);
    if (!more_items) break;
    item = next_item;
  }
  w2_dma_wait();   // nothing in flight when the workgroup's LDS is released
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
  static Seg3dOncePerDevice configured[4];
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<false, false>), configured[0], "conv3d_k3_wino2d")) return rc;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<true, false>), configured[1], "conv3d_k3_wino2d")) return rc;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<false, true>), configured[2], "conv3d_k3_wino2d")) return rc;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<true, true>), configured[3], "conv3d_k3_wino2d")) return rc;
  const int ntz = D / W2_TS, nty = H / W2_TS, ntx = W / W2_TS, ncog = Cout / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid((unsigned)(nitems < 256 ? nitems : 256), 1, 1);
#define W2_LAUNCH(B_, A_)                                                                                                       \
  hipLaunchKernelGGL((conv3d_k3_wino2d_kernel<B_, A_>), grid, dim3(256), (size_t)W2_LDS_FLOATS * 4, (hipStream_t)stream, x, wp, \
                     bias, y, stats, N, D, H, W, Cin, Cout, ntz, nty, ntx, ncog, nitems, addend)
  if (bias) {
    if (addend) W2_LAUNCH(true, true); else W2_LAUNCH(true, false);
  } else {
    if (addend) W2_LAUNCH(false, true); else W2_LAUNCH(false, false);
  }
#undef W2_LAUNCH
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_fwd");
  return SEG3D_OK;
}

// ================================================================================================================
// Weight gradient of the same layers with Winograd F(3x3, 2x2) over (y, x) (conv3d_k3_wgrad_wino2d_kernel), the transpose
// of the kernel above.  For an output QUAD (g = dy at (y0..y0+1, x0..x0+1) of plane z) and the 4 x 4 input patch d of plane
// z + kz - 1 the nine (ky, kx) taps of one kz are a 3 x 3-output correlation with a 2 x 2 filter: 16 rank-1 updates
// instead of 36,
//     V = B^T d B  (as above)        E = G' g G'^T,  G' = [1 0; 1 1; 1 -1; 0 1]  (the 1/2 factors move to the output)
//     M_p = sum_quads V_p (x) E_p    dW[kz] = A'^T M A',  A'^T = [1 1/2 1/2 0; 0 1/2 -1/2 0; 0 1/2 1/2 -1]
// i.e. 48 accumulators [3 kz][16 points] of [32 ci][32 co] instead of 27 taps, fed one QUAD per K slot: 4/9 of the MFMAs
// of conv3d_k3_wgrad2_kernel (2/3 of conv3d_k3_wgrad_wino_kernel's).
// Structure = conv3d_k3_wgrad_wino_kernel: one persistent workgroup per CU owns a 32 x 32 (ci, co) block pair and one slab
// of 4 x 4 x 4 tiles; wave w owns the point row py = w: 12 accumulators [kz][px] in registers across all tiles; the next
// tile's RAW x halo tile (6^3 voxels) and dy tile arrive by (inline-assembly) LDS-DMA behind the first K steps, branch-free;
// between two tiles 192 threads transform RAW x into T[p][z][quad][32 ci] (two barriers per tile); E is formed from the raw
// dy quad in registers (4 FMAs / adds per 12 MFMAs).  Partial slabs [slab][pair][48][32][32] are reduced in fixed order,
// and turned into the 27 taps, by conv3d_k3_wgrad_wino2d_reduce16_kernel (bitwise reproducible).
// A tile is only 96 MFMAs per wave (2.6 us): too short to cover a DMA issued in the same tile, so tiles are fetched TWO
// ahead (RAW x double-, dy triple-buffered; the wait before the barrier is vmcnt(pieces of one tile), not 0).
// LDS: 2 x 27 KB RAW x + T 48 KB + 3 x 8 KB dy = 126 KB.
// ================================================================================================================
__device__ __forceinline__ int w2_mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

#define G2_XV 216                                 // halo voxels of the x tile (6^3)
#define G2_XS (G2_XV * 32)                        // floats of RAW x
#define G2_TS (16 * 24 * 32)                      // floats of T: [16 p][6 z x 4 quads][32 ci]
#define G2_YS (64 * 32)                           // floats of one dy tile
#define G2_LDS_FLOATS (2 * G2_XS + G2_TS + 3 * G2_YS)   // RAW x twice, T, three dy tiles: 129 KB
#define G2_XPC 27                                 // 1-KiB DMA pieces of RAW x
#define G2_YPC 8
#define G2_GX 7                                   // piece groups per wave: x
#define G2_GY 2                                   // ... dy
#define G2_NG (G2_GX + G2_GY)
#define G2_KS 8                                   // K steps per tile: 16 quads, two per step

__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                         float* __restrict__ part, int N, int D, int H, int W,
                                                                         int Cin, int Cout, int ntz, int nty, int ntx, int ntiles,
                                                                         int slabs, int COB32) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rawx = lds;                        // [2][216][32]
  float* timg = lds + 2 * G2_XS;
  float* rawy = lds + 2 * G2_XS + G2_TS;    // [3][64][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = point row py
  const int li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x % slabs;
  const int pg = blockIdx.x / slabs;                   // (ci block, co block)
  const int cib = pg / COB32, cob = pg % COB32;
  const int ci0 = cib * 32, co0 = cob * 32;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };  // exact for the small ranges used here
  // row combination of the dy quad for py = wave:  r_j = g0j' + e1 g1j   (py 0: g0j, 1: g0j + g1j, 2: g0j - g1j, 3: g1j --
  // there g0j' is read one row down and e1 = 0)
  const float e1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);
  const int g0off = wave == 3 ? 4 * 32 : 0;            // one y row of the 4 x 4 x 4 dy tile = 4 voxels

  f32x16 acc[12];   // [kz][px]
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // DMA pieces of this wave (as conv3d_k3_wgrad_wino_kernel): group g < GX is x piece min(wave + 4 g, XPC - 1), group
  // GX + g' is dy piece min(wave + 4 g', YPC - 1); lane -> voxel 8 p + (lane >> 3), channels 4 (lane & 7)..+3
  const int lv = lane >> 3, lq = lane & 7;
  int prel[G2_NG], pflag[G2_NG];
#pragma unroll
  for (int g = 0; g < G2_NG; ++g) {
    if (g < G2_GX) {
      const int p = wave + 4 * g < G2_XPC ? wave + 4 * g : G2_XPC - 1;
      const int v = p * 8 + lv;
      const int t = fdiv(v, 1.0f / 6.0f);
      const int hx = v - t * 6;
      const int hz = fdiv(t, 1.0f / 6.0f);
      const int hy = t - hz * 6;
      prel[g] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 4 * lq;
      pflag[g] = (hz == 0 ? 1 : 0) | (hz == 5 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == 5 ? 8 : 0) | (hx == 0 ? 16 : 0) |
                 (hx == 5 ? 32 : 0) | (ci0 + 4 * lq < Cin ? 0 : 64);
    } else {
      const int p = wave + 4 * (g - G2_GX) < G2_YPC ? wave + 4 * (g - G2_GX) : G2_YPC - 1;
      const int v = p * 8 + lv;
      const int co = co0 + 4 * lq;
      prel[g] = (((v >> 4) * H + ((v >> 2) & 3)) * W + (v & 3)) * Cout + co;
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
    tz0 = tiz * 4, ty0 = tiy * 4, tx0 = tix * 4;
  };
  auto issue_piece = [&](int g, float* xdst, float* ydst, const float* xbase, const float* ybase, int faces) {
    const float* base = g < G2_GX ? xbase : ybase;            // compile-time choice (g is an unrolled loop index)
    float* dst;
    if (g < G2_GX) dst = xdst + (wave + 4 * g < G2_XPC ? wave + 4 * g : G2_XPC - 1) * 256;
    else dst = ydst + (wave + 4 * (g - G2_GX) < G2_YPC ? wave + 4 * (g - G2_GX) : G2_YPC - 1) * 256;
    const float* src = (pflag[g] & faces) ? w2_zero16 : base + prel[g];
    w2_glds16(src, dst);
  };
  // transform item of this thread (tid < 192): (z, quad, channel quad)
  const int t_c4 = tid & 7, t_q = (tid >> 3) & 3, t_z = tid >> 5;
  const int t_src = ((t_z * 6 + 2 * (t_q >> 1)) * 6 + 2 * (t_q & 1)) * 32 + 4 * t_c4;
  const int t_dst = (t_z * 4 + t_q) * 32 + 4 * t_c4;
  auto transform = [&](const float* rx) {   // RAW x -> T: V = B^T d B per (z, quad), 16 points; packed fp32 adds
    if (tid < 192) {
      f32x2 dxl[4][4], dxh[4][4];   // [row][px], channel pairs (0, 1) and (2, 3)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float* s = rx + t_src + r * (6 * 32);
        const f32x4 d0 = *reinterpret_cast<const f32x4*>(s);
        const f32x4 d1 = *reinterpret_cast<const f32x4*>(s + 32);
        const f32x4 d2 = *reinterpret_cast<const f32x4*>(s + 64);
        const f32x4 d3 = *reinterpret_cast<const f32x4*>(s + 96);
        const f32x2 d0l = {d0[0], d0[1]}, d0h = {d0[2], d0[3]}, d1l = {d1[0], d1[1]}, d1h = {d1[2], d1[3]};
        const f32x2 d2l = {d2[0], d2[1]}, d2h = {d2[2], d2[3]}, d3l = {d3[0], d3[1]}, d3h = {d3[2], d3[3]};
        dxl[r][0] = w2_pk_sub(d0l, d2l), dxh[r][0] = w2_pk_sub(d0h, d2h);
        dxl[r][1] = w2_pk_add(d1l, d2l), dxh[r][1] = w2_pk_add(d1h, d2h);
        dxl[r][2] = w2_pk_sub(d2l, d1l), dxh[r][2] = w2_pk_sub(d2h, d1h);
        dxl[r][3] = w2_pk_sub(d1l, d3l), dxh[r][3] = w2_pk_sub(d1h, d3h);
      }
      float* dst = timg + t_dst;
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const f32x2 v0l = w2_pk_sub(dxl[0][px], dxl[2][px]), v0h = w2_pk_sub(dxh[0][px], dxh[2][px]);
        const f32x2 v1l = w2_pk_add(dxl[1][px], dxl[2][px]), v1h = w2_pk_add(dxh[1][px], dxh[2][px]);
        const f32x2 v2l = w2_pk_sub(dxl[2][px], dxl[1][px]), v2h = w2_pk_sub(dxh[2][px], dxh[1][px]);
        const f32x2 v3l = w2_pk_sub(dxl[1][px], dxl[3][px]), v3h = w2_pk_sub(dxh[1][px], dxh[3][px]);
        *reinterpret_cast<f32x4*>(dst + (0 * 4 + px) * (24 * 32)) = f32x4{v0l[0], v0l[1], v0h[0], v0h[1]};
        *reinterpret_cast<f32x4*>(dst + (1 * 4 + px) * (24 * 32)) = f32x4{v1l[0], v1l[1], v1h[0], v1h[1]};
        *reinterpret_cast<f32x4*>(dst + (2 * 4 + px) * (24 * 32)) = f32x4{v2l[0], v2l[1], v2h[0], v2h[1]};
        *reinterpret_cast<f32x4*>(dst + (3 * 4 + px) * (24 * 32)) = f32x4{v3l[0], v3l[1], v3h[0], v3h[1]};
      }
    }
  };
  auto tile_faces = [&]() {
    return 64 | (tz0 == 0 ? 1 : 0) | (tz0 + 4 >= D ? 2 : 0) | (ty0 == 0 ? 4 : 0) | (ty0 + 4 >= H ? 8 : 0) |
           (tx0 == 0 ? 16 : 0) | (tx0 + 4 >= W ? 32 : 0);
  };

  // tile walk: XCD-contiguous when the slab count allows (as the forward kernels)
  int tile = slab, tstride = slabs, tlimit = ntiles;
  if ((slabs & 7) == 0) {
    const int per_xcd = (ntiles + 7) >> 3, xcd = slab & 7;
    tile = xcd * per_xcd + (slab >> 3);
    tstride = slabs >> 3;
    tlimit = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
  }
  auto fetch = [&](int t, float* xdst, float* ydst) {   // all pieces of tile t at once (prologue)
    set_tile(t);
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    const int faces = tile_faces();
#pragma unroll
    for (int g = 0; g < G2_NG; ++g) issue_piece(g, xdst, ydst, x + origin * Cin, dy + origin * Cout, faces);
  };
  int xi = 0, yi = 0;   // buffers of the current tile: RAW x (already transformed) xi, dy yi
  if (tile < tlimit) {
    fetch(tile, rawx, rawy);
    fetch(tile + tstride < tlimit ? tile + tstride : tile, rawx + G2_XS, rawy + G2_YS);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G2_NG) : "memory");   // the first tile landed, the second may be in flight
    __syncthreads();
    transform(rawx);
    __syncthreads();
  }
  for (; tile < tlimit; tile += tstride) {
    const float* ycur = rawy + yi * G2_YS;
    const int y2 = yi == 0 ? 2 : yi - 1;                 // (yi + 2) % 3
    float* ynxt2 = rawy + y2 * G2_YS;                    // dy buffer of the tile after next
    float* xnxt2 = rawx + xi * G2_XS;                    // RAW x of the current tile is dead (transformed): reuse it
    const bool more = tile + tstride < tlimit;
    // the tile after next -- past the end: this tile once more, into idle buffers (keeps the loop and the DMA count uniform)
    set_tile(tile + 2 * tstride < tlimit ? tile + 2 * tstride : tile);
    const int faces = tile_faces();
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    const float* xbase = x + origin * Cin;
    const float* ybase = dy + origin * Cout;
    // K step k: quads 2k (lane half 0) and 2k + 1 (half 1) = (z, qy) = (k >> 1, k & 1), qx = lane half
    const float* ta = timg + (wave * 4) * (24 * 32) + lane;        // + (px * 24 + (z + kz) * 4 + 2 qy) * 32
    const float* yb = ycur + lh * 64 + li;                         // + ((z * 4 + 2 qy) * 4) * 32; j: + 32, i: + 128
    auto aoff = [](int k, int kz, int px) { return (px * 24 + ((k >> 1) + kz) * 4 + 2 * (k & 1)) * 32; };
    auto yoff = [](int k) { return (((k >> 1) * 4 + 2 * (k & 1)) * 4) * 32; };
    float a1[12], g00, g01, g10, g11;
#pragma unroll
    for (int j = 0; j < 12; ++j) a1[j] = ta[aoff(0, j >> 2, j & 3)];
    g00 = yb[yoff(0) + g0off];
    g01 = yb[yoff(0) + g0off + 32];
    g10 = yb[yoff(0) + 128];
    g11 = yb[yoff(0) + 128 + 32];
#pragma unroll
    for (int k = 0; k < G2_KS; ++k) {
      float a[12], e[4];
#pragma unroll
      for (int j = 0; j < 12; ++j) a[j] = a1[j];
      const float r0 = fmaf(e1, g10, g00), r1 = fmaf(e1, g11, g01);
      e[0] = r0;
      e[1] = r0 + r1;
      e[2] = r0 - r1;
      e[3] = r1;
      if (k + 1 < G2_KS) {   // operands of step k + 1 are read while step k is multiplied
#pragma unroll
        for (int j = 0; j < 12; ++j) a1[j] = ta[aoff(k + 1, j >> 2, j & 3)];
        g00 = yb[yoff(k + 1) + g0off];
        g01 = yb[yoff(k + 1) + g0off + 32];
        g10 = yb[yoff(k + 1) + 128];
        g11 = yb[yoff(k + 1) + 128 + 32];
      }
#pragma unroll
      for (int g = 0; g < G2_NG; ++g)
        if (g % G2_KS == k) issue_piece(g, xnxt2, ynxt2, xbase, ybase, faces);
      __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from sinking the reads above down to their first use
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], e[j & 3], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G2_NG) : "memory");   // the NEXT tile landed (fetched one tile ago)
    __syncthreads();     // everyone done with T
    xi ^= 1;
    yi = yi == 2 ? 0 : yi + 1;
    if (more) {
#ifndef G2_EXP_NOTR    // (diagnostic builds: what do the exposed transform and its barrier cost?)
      transform(rawx + xi * G2_XS);   // the next tile's RAW x -> T
#endif
#ifndef G2_EXP_NOBAR2
      __syncthreads();
#endif
    }
  }
  w2_dma_wait();   // nothing in flight when the workgroup's LDS is released

  // part[slab][pair = cib * COB32 + cob][kz * 16 + py * 4 + px][ci row][co col]
  float* dst = part + ((i64)slab * (COB32 * ((Cin + 31) / 32)) + cib * COB32 + cob) * 48 * 1024;
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dst[((j >> 2) * 16 + wave * 4 + (j & 3)) * 1024 + w2_mfma_row(r, lh) * 32 + li] = acc[j][r];
}

// dw[a*sa + b*sb + kz*9 + ky*3 + kx] from  M[py][px] = sum_slab part[slab][a/32][b/32][kz*16 + py*4 + px][a%32][b%32]:
//   dW[kz] = A'^T M A',  A'^T = [1 1/2 1/2 0; 0 1/2 -1/2 0; 0 1/2 1/2 -1].
// A lane owns ONE point row (py: four 16-byte loads per slab) of one position (pair, kz, a, b quad); a workgroup takes 16
// positions with its G waves striding over the slabs (G = 16 from 64 slabs up: 48 workgroups per block pair and 64 KB of
// loads in flight each; G = 4 below); per-wave sums meet in LDS in a fixed order (slabs g, g + G, .. per wave, then waves
// 0..G-1: bitwise reproducible), the four rows of a position meet in LDS for the output transform, and lanes py < 3 write
// the output row ky = py.  (The first form -- one lane per position, 16 loads per slab, 12 workgroups per block pair, 36
// scattered scalar stores per lane -- ran 50 MB of slabs at 1.7 TB/s: 29 us where this one takes 15.)
template <int G>
__global__ __launch_bounds__(64 * G) void conv3d_k3_wgrad_wino2d_reduce16_kernel(const float* __restrict__ part,
                                                                                 float* __restrict__ dw, int slabs, int A, int B,
                                                                                 int BB32, int npairs, i64 sa, i64 sb,
                                                                                 int accumulate) {
  __shared__ f32x4 red[G * 64 * 4];                             // [wave][lane][px]: 64 KB at G = 16
  const int lane = threadIdx.x & 63, g = threadIdx.x >> 6;
  const int ql = lane & 15, py = lane >> 4;
  const i64 qidx = (i64)blockIdx.x * 16 + ql;                   // position (pair, kz, a, b quad); their count is a multiple of 16
  const i64 slabq = (i64)npairs * 48 * 256;                     // float4 quads per slab
  f32x4 s[4];
#pragma unroll
  for (int px = 0; px < 4; ++px) s[px] = f32x4{0.f, 0.f, 0.f, 0.f};
  {
    const i64 r3 = qidx >> 8;                                   // pair * 3 + kz
    const f32x4* p0 = reinterpret_cast<const f32x4*>(part) + (r3 * 16 + py * 4) * 256 + (qidx & 255);
    for (int k = g; k < slabs; k += G) {
      const f32x4* q = p0 + (i64)k * slabq;
#pragma unroll
      for (int px = 0; px < 4; ++px) s[px] += q[px * 256];
    }
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) red[(g * 64 + lane) * 4 + px] = s[px];
  __syncthreads();
  if (g == 0) {
    // this lane's point row over all 16 waves, back into LDS (slot of wave 0), then every lane reads the four rows of its
    // position: lanes with py < 3 produce the output row ky = py
#pragma unroll
    for (int j = 1; j < G; ++j)
#pragma unroll
      for (int px = 0; px < 4; ++px) s[px] += red[(j * 64 + lane) * 4 + px];
#pragma unroll
    for (int px = 0; px < 4; ++px) red[lane * 4 + px] = s[px];
  }
  __syncthreads();
  if (g == 0 && py < 3) {
    f32x4 m[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int px = 0; px < 4; ++px) m[r][px] = red[((r * 16 + ql)) * 4 + px];
    f32x4 rw[4];   // output row ky = py of A'^T M
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      const f32x4 hs = (m[1][px] + m[2][px]) * 0.5f, hd = (m[1][px] - m[2][px]) * 0.5f;
      rw[px] = py == 0 ? m[0][px] + hs : (py == 1 ? hd : hs - m[3][px]);
    }
    const f32x4 hs = (rw[1] + rw[2]) * 0.5f, hd = (rw[1] - rw[2]) * 0.5f;
    const f32x4 w0 = rw[0] + hs, w1 = hd, w2 = hs - rw[3];
    const int b32 = (int)((qidx & 7) * 4), a32 = (int)((qidx >> 3) & 31);
    const i64 r3 = qidx >> 8;
    const int kz = (int)(r3 % 3);
    const int pair = (int)(r3 / 3);
    const int a = (pair / BB32) * 32 + a32, b = (pair % BB32) * 32 + b32;
    if (a < A) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (b + j < B) {
          float* d = dw + a * sa + (b + j) * sb + kz * 9 + py * 3;
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

static int g2_slabs(int N, int D, int H, int W, int Cin, int Cout) {
  const i64 ntiles = (i64)N * (D / 4) * (H / 4) * (W / 4);
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  i64 slabs = 256 / npairs;  // one resident workgroup per CU over the whole grid
  if (slabs > (ntiles + 1) / 2) slabs = (ntiles + 1) / 2;  // small levels: >= 2 tiles per workgroup
  if (slabs < 1) slabs = 1;
  return (int)slabs;
}

// shapes the F(3x3, 2x2) weight gradient takes: whole 4 x 4 x 4 tiles, channels in fours
extern "C" int seg3d_conv3d_k3_wino2d_wgrad_supported(int N, int D, int H, int W, int Cin, int Cout) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if ((D % 4) || (H % 4) || (W % 4) || (Cin & 3) || (Cout & 3)) return 0;
  if ((long long)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
  if ((long long)N * (D / 4) * (H / 4) * (W / 4) >= SEG3D_FDIV_MAX) return 0;
  return 1;
}

// ... and where it is the faster choice (else F(3, 2) along x, conv_wino.hip): enough tiles per workgroup (>= 10) to amortise
// its 48 partial accumulators
extern "C" int seg3d_conv3d_k3_wino2d_wgrad_preferred(int N, int D, int H, int W, int Cin, int Cout) {
  if (!seg3d_conv3d_k3_wino2d_wgrad_supported(N, D, H, W, Cin, Cout)) return 0;
  const int slabs = g2_slabs(N, D, H, W, Cin, Cout);
  const long long ntiles = (long long)N * (D / 4) * (H / 4) * (W / 4);
  return ntiles >= 10ll * slabs;
}

extern "C" long long seg3d_conv3d_k3_wino2d_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  return (long long)g2_slabs(N, D, H, W, Cin, Cout) * npairs * 48 * 1024;
}

// x [N][D][H][W][Cin], dy [N][D][H][W][Cout]; dw in the reference Conv3d layout [Cout][Cin][3][3][3] (written, or added to
// when accumulate != 0); workspace = seg3d_conv3d_k3_wino2d_wgrad_workspace_floats floats
extern "C" int seg3d_conv3d_k3_wino2d_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D, int H,
                                            int W, int Cin, int Cout, int accumulate, void* stream) {
  SEG3D_REQUIRE(x && dy && dw && workspace, "seg3d_conv3d_k3_wino2d_wgrad: null pointer");
  SEG3D_REQUIRE(seg3d_conv3d_k3_wino2d_wgrad_supported(N, D, H, W, Cin, Cout),
                "seg3d_conv3d_k3_wino2d_wgrad: shape not supported (whole 4^3 tiles, Cin %% 4 == 0, Cout %% 4 == 0)");
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wgrad_wino2d_kernel), configured, "conv3d_k3_wgrad_wino2d")) return rc;
  const int slabs = g2_slabs(N, D, H, W, Cin, Cout);
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32, npairs = CIB32 * COB32;
  const int ntz = D / 4, nty = H / 4, ntx = W / 4;
  const int ntiles = N * ntz * nty * ntx;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3d_k3_wgrad_wino2d_kernel, dim3((unsigned)(slabs * npairs)), dim3(256), (size_t)G2_LDS_FLOATS * 4, s, x,
                     dy, workspace, N, D, H, W, Cin, Cout, ntz, nty, ntx, ntiles, slabs, COB32);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_wgrad");
  const i64 totalq = (i64)npairs * 3 * 256;
  // one point row per lane; 16 waves over the slabs where there are many of them, 4 otherwise
  if (slabs >= 64)
    hipLaunchKernelGGL(conv3d_k3_wgrad_wino2d_reduce16_kernel<16>, dim3((unsigned)(totalq / 16)), dim3(1024), 0, s, workspace, dw,
                       slabs, Cin, Cout, COB32, npairs, (i64)27, (i64)Cin * 27, accumulate);
  else
    hipLaunchKernelGGL(conv3d_k3_wgrad_wino2d_reduce16_kernel<4>, dim3((unsigned)(totalq / 16)), dim3(256), 0, s, workspace, dw,
                       slabs, Cin, Cout, COB32, npairs, (i64)27, (i64)Cin * 27, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_wgrad(reduce)");
  return SEG3D_OK;
}
