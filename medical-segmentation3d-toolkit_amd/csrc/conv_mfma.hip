// conv_mfma.hip -- 3x3x3 stride-1 pad-1 convolution (forward, data-gradient, weight-gradient) as an
// implicit GEMM on the CDNA4 matrix cores, exact fp32 (v_mfma_f32_32x32x2_f32).
//
// These are the layers that hold 99 % of the V-Net FLOPs (every C->C k3 conv with
// C >= 16, reference call sites network/module/conv_gn_relu3.py:10, residual_block3.py:13-16). In fp32 they
// are FLOP-bound (arithmetic intensity 216..1152 FLOP/B vs. a ridge of ~20), so the roof is the fp32 MFMA
// peak of 157.3 TFLOP/s, not HBM.
//
// Forward / dgrad  (conv3d_k3_mfma_kernel)
//   GEMM view: M = voxels, N = Cout, K = 27 taps x Cin.  A workgroup (4 waves) owns one spatial tile
//   TZ x TY x TX of output voxels and one block of 32 output channels.  The input tile with its 1-voxel halo
//   is staged in LDS 8 input channels at a time as [half][halo voxel][4 floats] so that a lane's A operands
//   for 4 consecutive MFMAs come from ONE ds_read_b128; the matching weight chunk [27][half][32][4] is a
//   straight copy of the packed weights (layout.hip: pack_mfma_kernel).  MFMA r of a group takes
//   k = {r, 4 + r} of the 8-channel chunk (lanes 0-31 / 32-63), which both operands agree on.
//   Each wave keeps MA accumulators (32 voxels x 32 channels each).  Zero padding = zeros written to LDS.
//   dgrad is the same kernel run on dy with flipped / transposed weights (pack with flip = 1).
//   The epilogue adds the bias and emits the per-workgroup (sum, sum of squares) that GroupNorm(1, C)
//   needs (conv_gn_relu3.py:11), so the statistics cost no extra pass over y.
//
// Weight gradient (conv3d_k3_wgrad_mfma_kernel)
//   GEMM view: M = Cin (32 block), N = Cout (32 block), K = voxels; one accumulator per tap.  A workgroup owns
//   one (ci block, co block) pair and walks a strided share of the spatial tiles; wave w owns taps 7w..7w+6.
//   A = x[v + tap][ci] and B = dy[v][co] are single ds_read_b32 per lane (lanes 0-31: voxel 2k, 32-63: 2k+1).
//   Partial slabs are reduced in fixed order by conv3d_k3_wgrad_reduce_kernel (bitwise reproducible).
#include <type_traits>
#include "seg3d_common.h"
#include "seg3d_hip.h"
#include <math.h>
#include <stdlib.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define SEG3D_MAXE 10          // float4 input-tile loads per thread per K-chunk  (2 * NV <= 2560)
#define SEG3D_W_CHUNK 6912     // 27 * 2 * 32 * 4 floats

__device__ __forceinline__ int mfma_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

template <int MA>
__global__ __launch_bounds__(256, 2) void conv3d_k3_mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, float* __restrict__ kpart, int cpk, const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HY = TY + 2, HX = TX + 2;
  const int NV = (TZ + 2) * HY * HX;
  const int MT = TZ * TY * TX;
  float* xs = lds;                                   // [2][NV][4]
  float* ws = lds + 8 * NV;                          // [27][2][32][4]
  int* voff = reinterpret_cast<int*>(ws + SEG3D_W_CHUNK);  // [MT] global voxel index or -1

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

  // global offsets (in floats) of this thread's staged float4s; the voxel set is the same for every K-chunk
  int goff[SEG3D_MAXE];
  const int hh = tid & 1;
#pragma unroll
  for (int e = 0; e < SEG3D_MAXE; ++e) {
    const int eidx = tid + e * 256;
    goff[e] = -1;
    if (eidx < 2 * NV) {
      const int v = eidx >> 1;
      const int hx = v % HX;
      const int t = v / HX;
      const int hy = t % HY;
      const int hz = t / HY;
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
        goff[e] = (((n * D + gz) * H + gy) * W + gx) * Cin + hh * 4;
    }
  }
  for (int idx = tid; idx < MT; idx += 256) {
    const int tx = idx % TX;
    const int t = idx / TX;
    const int ty = t % TY;
    const int tz = t / TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    voff[idx] = (gz < D && gy < H && gx < W) ? ((n * D + gz) * H + gy) * W + gx : -1;
  }
  // LDS base (in floats) of each of this lane's A rows
  int abase[MA];
#pragma unroll
  for (int m = 0; m < MA; ++m) {
    const int idx = (wave + 4 * m) * 32 + li;
    int vb = 0;
    if (idx < MT) {
      const int tx = idx % TX;
      const int t = idx / TX;
      const int ty = t % TY;
      const int tz = t / TY;
      vb = (tz * HY + ty) * HX + tx;
    }
    abase[m] = (lh * NV + vb) * 4;
  }
  const int bbase = (lh * 32 + li) * 4;

  f32x16 acc[MA];
#pragma unroll
  for (int m = 0; m < MA; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;

  // Software pipeline (register prefetch, one LDS buffer): the global loads of chunk c+1 (input tile + weights) are
  // issued BEFORE the MFMA block of chunk c and stay in flight behind it (nothing in the block waits on vmcnt: both
  // operands come from LDS); after the block a barrier retires the LDS reads, the parked registers are written to
  // LDS, and a second barrier publishes them.  Global latency is hidden inside the workgroup instead of relying on
  // the co-resident workgroup being out of phase.
  f32x4 xstage[SEG3D_MAXE];
  f32x4 wstage[7];
  auto load_chunk = [&](int cib) {
    // branch-free: out-of-range lanes read a valid dummy address and are zeroed by a select, so the compiler keeps
    // all loads in flight instead of waiting inside exec-masked branches
    // (the select itself happens in store_chunk: consuming a loaded value here would make the compiler wait for
    // each load before issuing the next)
    const bool half_ok = cib * 8 + hh * 4 < Cin;
#pragma unroll
    for (int e = 0; e < SEG3D_MAXE; ++e) {
      const bool ok = goff[e] >= 0 && half_ok;
      xstage[e] = *reinterpret_cast<const f32x4*>(x + (ok ? (i64)goff[e] + cib * 8 : (i64)0));
    }
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + ((i64)cob * CIB + cib) * SEG3D_W_CHUNK);
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int idx = tid + k * 256;
      wstage[k] = wsrc[idx < SEG3D_W_CHUNK / 4 ? idx : 0];
    }
  };
  auto store_chunk = [&](int cib) {
    const bool half_ok = cib * 8 + hh * 4 < Cin;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < SEG3D_MAXE; ++e) {
      const int eidx = tid + e * 256;
      const bool ok = goff[e] >= 0 && half_ok;
      if (eidx < 2 * NV) *reinterpret_cast<f32x4*>(xs + (hh * NV + (eidx >> 1)) * 4) = ok ? xstage[e] : zero;
    }
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      const int idx = tid + k * 256;
      if (idx < SEG3D_W_CHUNK / 4) reinterpret_cast<f32x4*>(ws)[idx] = wstage[k];
    }
  };
  // split-K (deep, spatially tiny levels): blockIdx.z owns the K-chunks [cib0, cib1) and writes a raw partial slab
  const int cib0 = kpart ? blockIdx.z * cpk : 0;
  const int cib1 = kpart ? (cib0 + cpk < CIB ? cib0 + cpk : CIB) : CIB;
  load_chunk(cib0);
  store_chunk(cib0);
  __syncthreads();
  for (int cib = cib0; cib < cib1; ++cib) {
    if (cib + 1 < cib1) load_chunk(cib + 1);
    {
      // LDS operands are double-buffered in registers: tap t+1 is read before the 4*MA MFMAs of tap t are issued
      f32x4 bw = *reinterpret_cast<const f32x4*>(ws + bbase);
      f32x4 av[MA];
#pragma unroll
      for (int m = 0; m < MA; ++m) av[m] = *reinterpret_cast<const f32x4*>(xs + abase[m]);
#pragma unroll
      for (int tap = 0; tap < 27; ++tap) {
        f32x4 bwn = bw;
        f32x4 avn[MA];
#pragma unroll
        for (int m = 0; m < MA; ++m) avn[m] = av[m];
        if (tap + 1 < 27) {
          const int t1 = tap + 1;
          const int kz = t1 / 9, ky = (t1 / 3) % 3, kx = t1 % 3;
          const int tapoff = ((kz * HY + ky) * HX + kx) * 4;
          bwn = *reinterpret_cast<const f32x4*>(ws + t1 * 256 + bbase);
#pragma unroll
          for (int m = 0; m < MA; ++m) avn[m] = *reinterpret_cast<const f32x4*>(xs + abase[m] + tapoff);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int m = 0; m < MA; ++m)
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[m][r], bw[r], acc[m], 0, 0, 0);
        bw = bwn;
#pragma unroll
        for (int m = 0; m < MA; ++m) av[m] = avn[m];
      }
    }
    if (cib + 1 < cib1) {
      __syncthreads();  // every wave is done reading chunk c
      store_chunk(cib + 1);
      __syncthreads();  // chunk c+1 visible
    }
  }

  // ---- epilogue: bias, store, GroupNorm partial statistics ----
  // Two passes so that every store reads its own accumulator register: a shared temporary would force the compiler
  // to wait (vmcnt) for the previous store before reusing it, serialising 16*MA stores per lane.
  const int co = cob * 32 + li;
  const bool co_ok = co < Cout;
  if (kpart) {  // raw partial sums; bias, store and statistics happen in conv3d_splitk_finish_kernel
    float* dstp = kpart + (i64)blockIdx.z * N * D * H * W * Cout;
#pragma unroll
    for (int m = 0; m < MA; ++m) {
      const int sub = wave + 4 * m;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int idx = sub * 32 + mfma_row(r, lh);
        const int vo = idx < MT ? voff[idx] : -1;
        if (vo >= 0 && co_ok) dstp[(i64)vo * Cout + co] = acc[m][r];
      }
    }
    return;
  }
  const float bv = (bias && co_ok) ? bias[co] : 0.f;
  float s[2] = {0.f, 0.f};
  int ooff[MA][16];
#pragma unroll
  for (int m = 0; m < MA; ++m) {
    const int sub = wave + 4 * m;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int idx = sub * 32 + mfma_row(r, lh);
      const int vo = idx < MT ? voff[idx] : -1;
      const bool ok = vo >= 0 && co_ok;
      ooff[m][r] = ok ? vo * Cout + co : -1;
      acc[m][r] += bv;
      if (addend && ok) acc[m][r] += addend[(i64)ooff[m][r]];  // fused "+ residual-path gradient" (dgrad use)
      const float val = ok ? acc[m][r] : 0.f;
      s[0] += val;
      s[1] += val * val;
    }
  }
#pragma unroll
  for (int m = 0; m < MA; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if (ooff[m][r] >= 0) y[(i64)ooff[m][r]] = acc[m][r];
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

// ----------------------------------------------------------------------------------------------------------------
// Forward / dgrad, second generation (conv3d_k3_mfma2_kernel): ONE persistent workgroup per CU, one wave per SIMD.
//   Measured on MI355X (tools/ubench/mfma_shape.hip): an LDS-fed 32x32x2 loop sustains 154 TFLOP/s, so the matrix
//   pipe is only kept waiting by what surrounds the loop; in-kernel stamps (tools/ubench/conv_stamp.hip) priced that
//   at ~30 % of a short-K workgroup (prologue DMA burst, index math, dword store tail).  Here:
//   * the next K-chunk (input halo tile + NB weight blocks) is brought in by LDS-DMA (global_load_lds_dwordx4: no
//     staging registers, no ds_write pass) into the second of two LDS buffers while the MFMAs of the current chunk
//     run, one 1-KiB piece per tap; halo voxels outside the volume read a 16-byte zero source instead;
//   * one barrier per chunk (after the wave's own DMA count has drained) publishes the new buffer and frees the old;
//   * the workgroup is persistent and walks work items (tile, column group) with a stride of gridDim.x: the LAST
//     chunk of an item already fetches chunk 0 of the next item, so no item but the first pays a DMA prologue, and
//     the epilogue's stores drain behind the next item's MFMAs;
//   * the MFMA operands are swapped (A = weights, B = voxels): a lane then owns ONE voxel and 4 x 4 consecutive
//     output channels of each accumulator, so the epilogue is 4 dwordx4 stores per accumulator instead of 16 dword
//     stores and needs no LDS voxel table; bias / fused addend are float4 loads;
//   * GroupNorm partial sums are written per wave (no workgroup reduction, no barrier);
//   * a workgroup owns MA x 4 row blocks (voxels) x NB column blocks (32 output channels each): the input tile is
//     staged once for NB*32 channels instead of once per 32.
// LDS: 2 x { xs [2][NV][4] padded to 1 KiB, ws [NB][27][2][32][4] }  (<= 160 KB).  Needs Cin % 8 == 0, Cout % 4 == 0.
// ----------------------------------------------------------------------------------------------------------------
__device__ __attribute__((aligned(16))) float seg3d_zero16[4];  // DMA source for zero padding

#define SEG3D_V2_MAXPX 10  // input-tile DMA pieces per wave per chunk (40 pieces = 2560 float4 = 2 halves x 1280 voxels)

// Diagnostic build only (tools/ubench/conv_stamp.hip defines SEG3D_STAMPS): wave 0 records s_memtime at phase
// boundaries into a buffer nothing else reads.  The product library is built without it (no stamp executes).
#ifdef SEG3D_STAMPS
__device__ long long* seg3d_stamp_buf;
#define SEG3D_STAMP(slot, k)                                                                                  \
  do {                                                                                                        \
    if (threadIdx.x == 0) seg3d_stamp_buf[(size_t)(slot) * 16 + (k)] = __builtin_amdgcn_s_memtime();           \
  } while (0)
#else
#define SEG3D_STAMP(slot, k) do { } while (0)
#endif

#ifndef SEG3D_XCD_WALK   // XCD-contiguous item walk of the persistent forward kernels
#define SEG3D_XCD_WALK 1
#endif
#ifndef SEG3D_LANE_SLOTS   // 0: tile-linear lane -> voxel order (measurement builds)
#define SEG3D_LANE_SLOTS 1
#endif
#ifndef SEG3D_BF16_PD   // taps of LDS operands kept in flight by the bf16 forward kernel
#define SEG3D_BF16_PD 3
#endif
#ifndef SEG3D_EXP_FWD_NODMA   // measurement builds: 1 = the forward kernels skip their steady-state LDS-DMA (results wrong)
#define SEG3D_EXP_FWD_NODMA 0
#endif

__device__ __forceinline__ void seg3d_glds16(const float* src, float* lds_dst_wave_uniform) {
  __builtin_amdgcn_global_load_lds(src, lds_dst_wave_uniform, 16, 0, 0);
}

// BF16 = true: x and wp hold bf16 (viewed here as 32-bit words, `Cin` = words per voxel = channels / 2).  The byte
// geometry is the same as in fp32 -- a 16-byte piece per (voxel, half), 1-KiB weight pieces per tap -- only a chunk is
// 16 channels instead of 8 and ONE v_mfma_f32_32x32x16_bf16 (a lane supplies the 8 channels of its half) replaces the
// four 32x32x2 fp32 MFMAs per tap.  Accumulators, bias, addend, output and statistics stay fp32.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// NW = waves per workgroup: 4 (one per SIMD) or 8 (two per SIMD, each with half the row blocks: while one wave is stuck
// issuing an LDS-DMA piece -- ~150 cycles, during which its single in-flight MFMA runs out -- the other one keeps the
// matrix pipe fed; needs <= 256 registers per wave, i.e. MA * NB <= 2).  The tile has 32 * NW * MA voxels either way.
template <int MA, int NB, bool SPLITK, bool BF16 = false, bool OUT_BF = false, int NW = 4>
__device__ __forceinline__ void conv3d_k3_mfma2_body(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, int ncog, int nitems, const float* __restrict__ addend, float* __restrict__ kpart, int ksplit, int cpk_arg) {
  // SPLITK is a compile-time switch so that the whole-K instantiation keeps exactly the code it had without it
  // split-K (kpart != nullptr): a work item is (tile, column group, K range of cpk chunks); its raw partial sums go to
  // slab `ks` of kpart and conv3d_splitk_finish_kernel adds the slabs, the bias, the addend and takes the statistics.
  // Used where whole-K items cannot fill 256 CUs evenly (12^3 and 6^3 levels): finer items remove the idle tail.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int HY = TY + 2, HX = TX + 2;
  const int NV = (TZ + 2) * HY * HX;
  const int MT = TZ * TY * TX;
  const int XS = (8 * NV + 255) & ~255;  // floats; whole 1-KiB DMA pieces
  const int BUF = XS + NB * SEG3D_W_CHUNK;

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int CIB = Cin >> 3;
  const int G = gridDim.x;

  // Index math without integer division: for 0 <= v < 2^20 and d <= 2^10, v / d == (int)((v + 0.5f) * (1.0f / d))
  // exactly (v + 0.5 is at least 0.5/d away from a multiple of d; the float error stays below that).  With one wave
  // per SIMD nothing hides such instructions, and a 32-bit division is ~35 of them.
  const float rHX = 1.0f / (float)HX, rHY = 1.0f / (float)HY, rTX = 1.0f / (float)TX, rTY = 1.0f / (float)TY;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz, rNCOG = 1.0f / (float)ncog;
  const float rKS = 1.0f / (float)ksplit;
  const int cpk = SPLITK ? cpk_arg : CIB;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };

  // ---- per-lane constants that do not depend on the work item ----
  // input-tile DMA pieces of this wave: piece p = wave + 4 j covers float4 entries e = 64 p + lane of [2][NV];
  // hpos[j] = halo coordinates (hz << 20 | hy << 10 | hx), bit 30 = upper channel half, -1 = padding entry
  const int NPX = XS >> 8;
  const int nx = NPX > wave ? (NPX - wave + NW - 1) / NW : 0;
  int hpos[SEG3D_V2_MAXPX];
#pragma unroll
  for (int j = 0; j < SEG3D_V2_MAXPX; ++j) {
    const int e = (wave + NW * j) * 64 + lane;
    hpos[j] = -1;
    if (e < 2 * NV) {
      const int hh = e >= NV;
      const int v = e - hh * NV;
      const int t = fdiv(v, rHX);
      const int hx = v - t * HX;
      const int hz = fdiv(t, rHY);
      const int hy = t - hz * HY;
      hpos[j] = (hh << 30) | (hz << 20) | (hy << 10) | hx;
    }
  }
  // this lane's MA voxels (one per accumulator row block): LDS base of the A rows and tile-local coordinates.
  // Which of the 32 voxels of a row block a lane takes is free (the lane feeds and owns MFMA column `li` either way); it
  // is chosen so that each 16-lane group the hardware serves a ds_read_b128 in -- {0-3, 12-15, 20-27}, {4-11, 16-19,
  // 28-31} and the same + 32 -- reads 16 CONSECUTIVE voxels of the tile: with tile rows of 16 voxels that is one contiguous
  // 256-byte run = all 64 banks once.  (In tile-linear lane order a group gathered 4 + 4 + 8 voxels from four rows 160
  // bytes apart: SQ_LDS_BANK_CONFLICT was 60 % of the LDS cycles of the bf16 kernel.)
  // NW = 8 with a 384-voxel tile: waves 4..7 own only one of their two row blocks (12 blocks over 8 waves = 3 per SIMD);
  // the MFMAs of the missing block are skipped (wave-uniform)
  const bool act1 = NW != 8 || MA < 2 || (wave + NW) * 32 < MT;
  const int lslot = li < 4 ? li : li < 12 ? li + 12 : li < 16 ? li - 8 : li < 20 ? li + 8 : li < 28 ? li - 12 : li;
  int abase[MA], vpos[MA];
#pragma unroll
  for (int m = 0; m < MA; ++m) {
    const int idx = (wave + NW * m) * 32 + (SEG3D_LANE_SLOTS ? lslot : li);
    int vb = 0;
    vpos[m] = -1;
    if (idx < MT) {
      const int t = fdiv(idx, rTX);
      const int tx = idx - t * TX;
      const int tz = fdiv(t, rTY);
      const int ty = t - tz * TY;
      vb = (tz * HY + ty) * HX + tx;
      vpos[m] = (tz << 20) | (ty << 10) | tx;
    }
    abase[m] = (lh * NV + vb) * 4;
  }
  const int bbase = (lh * 32 + li) * 4;

  // ---- work item state ----
  int it_n = 0, it_z0 = 0, it_y0 = 0, it_x0 = 0, it_cog = 0, it_tile = 0;  // item whose DMA sources are set up
  int it_ks = 0, it_c0 = 0, it_c1 = CIB;                                    // its K slab and chunk range
  const float* xsrc[SEG3D_V2_MAXPX];
  int xadv = 0;  // bit j: piece j advances by 8 channels per chunk (0 for zero-padding sources)
  auto setup_item = [&](int item) {
    int item_k = item;
    if (SPLITK) {
      item_k = fdiv(item, rKS);                      // K slab index varies fastest
      it_ks = item - item_k * ksplit;
      it_c0 = it_ks * cpk;
      it_c1 = it_c0 + cpk < CIB ? it_c0 + cpk : CIB;
    }
    const int tile_all = fdiv(item_k, rNCOG);
    it_cog = item_k - tile_all * ncog;
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
    it_z0 = tiz * TZ, it_y0 = tiy * TY, it_x0 = tix * TX;
    xadv = 0;
#pragma unroll
    for (int j = 0; j < SEG3D_V2_MAXPX; ++j) {
      xsrc[j] = seg3d_zero16;
      const int hp = hpos[j];
      const int gz = it_z0 + ((hp >> 20) & 1023) - 1, gy = it_y0 + ((hp >> 10) & 1023) - 1, gx = it_x0 + (hp & 1023) - 1;
      if (hp >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W) {
        xsrc[j] = x + ((i64)(((it_n * D + gz) * H + gy) * W + gx) * Cin + ((hp >> 30) & 1) * 4 + (SPLITK ? it_c0 * 8 : 0));
        xadv |= 1 << j;
      }
    }
  };
  auto dma_x = [&](int j, float* buf) {  // j compile-time after unrolling; issues the piece, then steps to the next chunk
    seg3d_glds16(xsrc[j], buf + (wave + NW * j) * 256);
    xsrc[j] += ((xadv >> j) & 1) * 8;
  };
  constexpr int NPW = 27 * NB;  // weight pieces (1 KiB = one tap of one column block)
  constexpr int MAXPW = (NPW + NW - 1) / NW;
  auto dma_w = [&](int j, const float* wchunk, float* buf) {  // wchunk: packed weights of (first column block, chunk)
    const int piece = wave + NW * j;
    if (piece < NPW) {
      const int nb = piece / 27, tap = piece - nb * 27;
      seg3d_glds16(wchunk + (i64)nb * CIB * SEG3D_W_CHUNK + tap * 256 + lane * 4, buf + XS + piece * 256);
    }
  };

  // Item walk.  SEG3D_XCD_WALK: workgroups go to the 8 XCDs round-robin (blockIdx % 8); XCD j then owns the contiguous
  // eighth j of the item list and its G / 8 workgroups walk it side by side, so that neighbouring tiles (overlapping halos)
  // are fetched through the same L2 at about the same time.
  int item = blockIdx.x, istride = G, ilimit = nitems;
  if (SEG3D_XCD_WALK && (G & 7) == 0) {
    const int per_xcd = (nitems + 7) >> 3, xcd = blockIdx.x & 7;
    item = xcd * per_xcd + (blockIdx.x >> 3);
    istride = G >> 3;
    ilimit = (xcd + 1) * per_xcd < nitems ? (xcd + 1) * per_xcd : nitems;
  }
  if (item >= ilimit) return;
  setup_item(item);
  SEG3D_STAMP(blockIdx.x, 0);
  {  // the only exposed DMA prologue of this workgroup: chunk 0 of its first item into buffer 0
    const float* w0 = wp + ((i64)(it_cog * NB) * CIB + (SPLITK ? it_c0 : 0)) * SEG3D_W_CHUNK;
#pragma unroll
    for (int j = 0; j < SEG3D_V2_MAXPX; ++j)
      if (j < nx) dma_x(j, lds);
#pragma unroll
    for (int j = 0; j < MAXPW; ++j) dma_w(j, w0, lds);
  }
  __syncthreads();  // drains this wave's DMA count (vmcnt) and publishes buffer 0
  SEG3D_STAMP(blockIdx.x, 1);

  int parity = 0;
  for (;;) {
    // the item being multiplied (its identity is needed again in the epilogue, after setup_item moved on)
    const int cur_n = it_n, cur_z0 = it_z0, cur_y0 = it_y0, cur_x0 = it_x0, cur_cog = it_cog, cur_tile = it_tile;
    const int cur_ks = it_ks, cur_c0 = SPLITK ? it_c0 : 0, cur_c1 = SPLITK ? it_c1 : CIB;
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;

    f32x16 acc[MA][NB];
#pragma unroll
    for (int m = 0; m < MA; ++m)
#pragma unroll
      for (int q = 0; q < NB; ++q)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;
    // this lane's 16 bias values per column block, fetched now (no DMA is in flight here) and used in the epilogue
    f32x4 bv[NB][4];
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int co = (cur_cog * NB + q) * 32 + 4 * lh + 8 * g4;
        const f32x4 val = *reinterpret_cast<const f32x4*>((bias && co < Cout) ? bias + co : seg3d_zero16);
        bv[q][g4] = val;
      }

    f32x4 ad[MA][NB][4];  // fused addend values of this item (loaded at the start of its last chunk)
    for (int cib = cur_c0; cib < cur_c1; ++cib) {
      float* cur = lds + parity * BUF;
      float* nxt = lds + (parity ^ 1) * BUF;
      const bool last = cib + 1 == cur_c1;
      const bool do_dma = !last || more_items;
      const float* wnext;
      if (last) {
        if (!SPLITK && addend) {
          // fused addend (dgrad of a residual block's first conv): fetched NOW, before this chunk's DMAs are issued,
          // so the loads complete behind the MFMAs of the last chunk instead of stalling the epilogue
#pragma unroll
          for (int m = 0; m < MA; ++m) {
            const int vp = vpos[m];
            const int gz = cur_z0 + ((vp >> 20) & 1023), gy = cur_y0 + ((vp >> 10) & 1023), gx = cur_x0 + (vp & 1023);
            const int vom = (vp >= 0 && gz < D && gy < H && gx < W) ? ((cur_n * D + gz) * H + gy) * W + gx : -1;
#pragma unroll
            for (int q = 0; q < NB; ++q)
#pragma unroll
              for (int g4 = 0; g4 < 4; ++g4) {
                const int co = cur_cog * NB * 32 + 4 * lh + 32 * q + 8 * g4;
                const bool ok = vom >= 0 && co < Cout;
                ad[m][q][g4] = *reinterpret_cast<const f32x4*>(addend + (ok ? (i64)vom * Cout + co : (i64)0));
              }
          }
        }
        if (more_items) setup_item(next_item);  // DMA sources now belong to the next item
        wnext = wp + ((i64)(it_cog * NB) * CIB + (SPLITK ? it_c0 : 0)) * SEG3D_W_CHUNK;
      } else {
        wnext = wp + ((i64)(cur_cog * NB) * CIB + cib + 1) * SEG3D_W_CHUNK;
      }
      const float* xs = cur;
      const float* ws = cur + XS;
      if constexpr (BF16) {
        // bf16: a tap is 4 * MA * NB / 4 MFMAs of 32 cycles -- 8x shorter than in fp32 -- so operands read one tap ahead
        // arrive too late (LDS latency > one tap).  A ring of SEG3D_BF16_PD taps of operands is kept in flight instead;
        // the compiler places the counted lgkmcnt waits itself (plain LDS loads).
        constexpr int PD = NW == 8 ? 1 : SEG3D_BF16_PD;   // (two waves per SIMD: 256 registers, and the other wave covers)
        f32x4 bwq[PD][NB], avq[PD][MA];
        auto tap_off = [&](int t) {
          const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
          return ((kz * HY + ky) * HX + kx) * 4;
        };
#pragma unroll
        for (int p = 0; p < PD; ++p) {
#pragma unroll
          for (int q = 0; q < NB; ++q) bwq[p][q] = *reinterpret_cast<const f32x4*>(ws + q * SEG3D_W_CHUNK + p * 256 + bbase);
#pragma unroll
          for (int m = 0; m < MA; ++m) avq[p][m] = *reinterpret_cast<const f32x4*>(xs + abase[m] + tap_off(p));
        }
#pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          const int slot = tap % PD;
          f32x4 bwc[NB], avc[MA];
#pragma unroll
          for (int q = 0; q < NB; ++q) bwc[q] = bwq[slot][q];
#pragma unroll
          for (int m = 0; m < MA; ++m) avc[m] = avq[slot][m];
          if (tap + PD < 27) {
            const int t1 = tap + PD;
#pragma unroll
            for (int q = 0; q < NB; ++q)
              bwq[slot][q] = *reinterpret_cast<const f32x4*>(ws + q * SEG3D_W_CHUNK + t1 * 256 + bbase);
#pragma unroll
            for (int m = 0; m < MA; ++m) avq[slot][m] = *reinterpret_cast<const f32x4*>(xs + abase[m] + tap_off(t1));
          }
          // keep the reads up here: left alone, the scheduler sinks every read to just in front of the MFMA that consumes
          // it (register pressure heuristics) and puts an lgkmcnt(0) wait before nearly every MFMA
          __builtin_amdgcn_sched_barrier(0);
          if (do_dma && !SEG3D_EXP_FWD_NODMA) {  // one DMA piece of the next chunk per tap, behind the MFMAs
            if (tap < SEG3D_V2_MAXPX) {
              if (tap < nx) dma_x(tap, nxt);
            } else if (tap - SEG3D_V2_MAXPX < MAXPW) {
              dma_w(tap - SEG3D_V2_MAXPX, wnext, nxt);
            }
          }
#pragma unroll
          for (int m = 0; m < MA; ++m) {
            if (NW == 8 && m == 1 && !act1) continue;
#pragma unroll
            for (int q = 0; q < NB; ++q)
              acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bwc[q]),
                                                                  __builtin_bit_cast(bf16x8, avc[m]), acc[m][q], 0, 0, 0);
          }
        }
      } else {
        f32x4 bw[NB], av[MA];
  #pragma unroll
        for (int q = 0; q < NB; ++q) bw[q] = *reinterpret_cast<const f32x4*>(ws + q * SEG3D_W_CHUNK + bbase);
  #pragma unroll
        for (int m = 0; m < MA; ++m) av[m] = *reinterpret_cast<const f32x4*>(xs + abase[m]);
  #pragma unroll
        for (int tap = 0; tap < 27; ++tap) {
          f32x4 bwn[NB], avn[MA];
  #pragma unroll
          for (int q = 0; q < NB; ++q) bwn[q] = bw[q];
  #pragma unroll
          for (int m = 0; m < MA; ++m) avn[m] = av[m];
          if (tap + 1 < 27) {  // operands of tap t+1 are read while tap t is multiplied
            const int t1 = tap + 1;
            const int kz = t1 / 9, ky = (t1 / 3) % 3, kx = t1 % 3;
            const int tapoff = ((kz * HY + ky) * HX + kx) * 4;
  #pragma unroll
            for (int q = 0; q < NB; ++q) bwn[q] = *reinterpret_cast<const f32x4*>(ws + q * SEG3D_W_CHUNK + t1 * 256 + bbase);
  #pragma unroll
            for (int m = 0; m < MA; ++m) avn[m] = *reinterpret_cast<const f32x4*>(xs + abase[m] + tapoff);
          }
          if (do_dma && !SEG3D_EXP_FWD_NODMA) {  // one DMA piece of the next chunk per tap, behind the MFMAs
            if (tap < SEG3D_V2_MAXPX) {
              if (tap < nx) dma_x(tap, nxt);
            } else if (tap - SEG3D_V2_MAXPX < MAXPW) {
              dma_w(tap - SEG3D_V2_MAXPX, wnext, nxt);
            }
          }
          // A = weights, B = voxels: D[co][voxel], a lane owns voxel (lane & 31) and channels 8 g + 4 (lane >> 5) + c
          if constexpr (BF16) {
  #pragma unroll
            for (int m = 0; m < MA; ++m)
  #pragma unroll
              for (int q = 0; q < NB; ++q)
                acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, bw[q]),
                                                                    __builtin_bit_cast(bf16x8, av[m]), acc[m][q], 0, 0, 0);
          } else {
  #pragma unroll
            for (int r = 0; r < 4; ++r)
  #pragma unroll
              for (int m = 0; m < MA; ++m) {
                if (NW == 8 && m == 1 && !act1) continue;
  #pragma unroll
                for (int q = 0; q < NB; ++q)
                  acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[q][r], av[m][r], acc[m][q], 0, 0, 0);
              }
          }
  #pragma unroll
          for (int q = 0; q < NB; ++q) bw[q] = bwn[q];
  #pragma unroll
          for (int m = 0; m < MA; ++m) av[m] = avn[m];
        }
      }
      if (do_dma) __syncthreads();  // own DMAs landed (vmcnt(0)), everyone done with `cur`, `nxt` visible
      parity ^= 1;
    }
    SEG3D_STAMP(item, 2);

    if (SPLITK) {
      // split-K: raw partial sums of this K range -> slab cur_ks (bias, addend and statistics belong to the finish pass)
      float* slab = kpart + (i64)cur_ks * N * D * H * W * Cout;
#pragma unroll
      for (int m = 0; m < MA; ++m) {
        const int vp = vpos[m];
        const int gz = cur_z0 + ((vp >> 20) & 1023), gy = cur_y0 + ((vp >> 10) & 1023), gx = cur_x0 + (vp & 1023);
        const bool vok = vp >= 0 && gz < D && gy < H && gx < W;
        const i64 vo = vok ? ((i64)(cur_n * D + gz) * H + gy) * W + gx : 0;
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int co = cur_cog * NB * 32 + 4 * lh + 32 * q + 8 * g4;
            if (vok && co < Cout) {
              f32x4 v;
#pragma unroll
              for (int c = 0; c < 4; ++c) v[c] = acc[m][q][4 * g4 + c];
              *reinterpret_cast<f32x4*>(slab + vo * Cout + co) = v;
            }
          }
      }
      if (!more_items) break;
      item = next_item;
      continue;
    }

    // ---- epilogue: bias (+ addend), dwordx4 stores, per-wave GroupNorm partial sums ----
    // Loads first, stores last, nothing in between: on gfx9 stores count in vmcnt too, so a load placed after a store
    // makes the compiler wait for that store's full round trip before the next one is issued.
    float s0 = 0.f, s1 = 0.f;
    int vo[MA];
#pragma unroll
    for (int m = 0; m < MA; ++m) {
      const int vp = vpos[m];
      const int gz = cur_z0 + ((vp >> 20) & 1023), gy = cur_y0 + ((vp >> 10) & 1023), gx = cur_x0 + (vp & 1023);
      vo[m] = (vp >= 0 && gz < D && gy < H && gx < W) ? ((cur_n * D + gz) * H + gy) * W + gx : -1;
    }
    const int co_lane = cur_cog * NB * 32 + 4 * lh;  // + 32 q + 8 g4
    if (addend) {
#pragma unroll
      for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[m][q][4 * g4 + c] += ad[m][q][g4][c];
    }
    // interior items (whole tile inside the volume, all 32-channel blocks complete) store unconditionally: straight-
    // line code, 4 * MA * NB dwordx4 stores back to back; the per-lane test of edge items costs exec-mask branches
    const bool whole = MT == 32 * NW * MA && cur_z0 + TZ <= D && cur_y0 + TY <= H && cur_x0 + TX <= W &&
                       (cur_cog + 1) * NB * 32 <= Cout;
    if (whole) {
#pragma unroll
      for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            f32x4 v;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[m][q][4 * g4 + c] + bv[q][g4][c];
              s0 += v[c];
              s1 += v[c] * v[c];
            }
#ifndef SEG3D_EXP_NOSTORE   // measurement builds only: skip the interior stores to price them
            Seg3dQuad<OUT_BF>::store(y, (i64)vo[m] * Cout + (co_lane + 32 * q + 8 * g4), v);   // y: bf16 when OUT_BF
#endif
          }
    } else {
#pragma unroll
      for (int m = 0; m < MA; ++m)
#pragma unroll
        for (int q = 0; q < NB; ++q)
#pragma unroll
          for (int g4 = 0; g4 < 4; ++g4) {
            const int co = co_lane + 32 * q + 8 * g4;
            if (vo[m] >= 0 && co < Cout) {
              f32x4 v;
#pragma unroll
              for (int c = 0; c < 4; ++c) {
                v[c] = acc[m][q][4 * g4 + c] + bv[q][g4][c];
                s0 += v[c];
                s1 += v[c] * v[c];
              }
              Seg3dQuad<OUT_BF>::store(y, (i64)vo[m] * Cout + co, v);
            }
          }
    }
    if (stats) {
      s0 = wave_sum(s0);
      s1 = wave_sum(s1);
      if (lane == 0) {
        const int tiles_per_sample = ntz * nty * ntx;
        float* dst = stats + ((((i64)cur_n * tiles_per_sample + cur_tile) * ncog + cur_cog) * NW + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
    SEG3D_STAMP(item, 3);
    if (!more_items) break;
    item = next_item;
  }
}

struct Seg3dTile {
  int tz, ty, tx;
};

#define SPLITK_CHUNK 4096  // elements per workgroup of the finish pass

template <int MA, int NB>
__global__ __launch_bounds__(256, 1) void conv3d_k3_mfma2_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, int ncog, int nitems, const float* __restrict__ addend) {
  conv3d_k3_mfma2_body<MA, NB, false>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, TZ, TY, TX, ntz, nty, ntx, ncog, nitems,
                                      addend, nullptr, 1, 0);
}

// two waves per SIMD (NW = 8): MA counts the row blocks PER WAVE, the tile has 256 * MA voxels
template <int MA, int NB>
__global__ __launch_bounds__(512, 1) void conv3d_k3_mfma2w8_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, int ncog, int nitems, const float* __restrict__ addend) {
  conv3d_k3_mfma2_body<MA, NB, false, false, false, 8>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, TZ, TY, TX, ntz, nty,
                                                       ntx, ncog, nitems, addend, nullptr, 1, 0);
}

// same loop over (tile, column group, K range) items; writes raw partial slabs for conv3d_splitk_finish_kernel
template <int MA, int NB>
__global__ __launch_bounds__(256, 1) void conv3d_k3_mfma2_splitk_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ kpart, int N, int D, int H, int W,
    int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty, int ntx, int ncog, int nitems, int ksplit, int cpk) {
  conv3d_k3_mfma2_body<MA, NB, true>(x, wp, nullptr, nullptr, nullptr, N, D, H, W, Cin, Cout, TZ, TY, TX, ntz, nty, ntx,
                                     ncog, nitems, nullptr, kpart, ksplit, cpk);
}

// bf16-input variants (x, wp: bf16 viewed as words; Cw = Cin / 2 words per voxel)
// OUT_BF: the output tensor is bf16 (a data-gradient that autograd hands on as the gradient of a bf16 activation); the
// statistics, when asked for, are still those of the fp32 values
template <int MA, int NB, bool OUT_BF>
__global__ __launch_bounds__(256, 1) void conv3d_k3_mfma2_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cw, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, int ncog, int nitems, const float* __restrict__ addend) {
  conv3d_k3_mfma2_body<MA, NB, false, true, OUT_BF>(x, wp, bias, y, stats, N, D, H, W, Cw, Cout, TZ, TY, TX, ntz, nty, ntx,
                                                    ncog, nitems, addend, nullptr, 1, 0);
}

template <int MA, int NB, bool OUT_BF>
__global__ __launch_bounds__(512, 1) void conv3d_k3_mfma2w8_bf16_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cw, int Cout, int TZ, int TY, int TX, int ntz, int nty,
    int ntx, int ncog, int nitems, const float* __restrict__ addend) {
  conv3d_k3_mfma2_body<MA, NB, false, true, OUT_BF, 8>(x, wp, bias, y, stats, N, D, H, W, Cw, Cout, TZ, TY, TX, ntz, nty,
                                                       ntx, ncog, nitems, addend, nullptr, 1, 0);
}

template <int MA, int NB>
__global__ __launch_bounds__(256, 1) void conv3d_k3_mfma2_bf16_splitk_kernel(
    const float* __restrict__ x, const float* __restrict__ wp, float* __restrict__ kpart, int N, int D, int H, int W,
    int Cw, int Cout, int TZ, int TY, int TX, int ntz, int nty, int ntx, int ncog, int nitems, int ksplit, int cpk) {
  conv3d_k3_mfma2_body<MA, NB, true, true>(x, wp, nullptr, nullptr, nullptr, N, D, H, W, Cw, Cout, TZ, TY, TX, ntz, nty,
                                           ntx, ncog, nitems, nullptr, kpart, ksplit, cpk);
}

// y[e] = bias[c] + sum_ks part[ks][e]; emits GroupNorm (sum, sumsq) partials per workgroup.  HBM-bound, float4.
template <bool OUT_BF>
__device__ __forceinline__ void conv3d_splitk_finish_body(const float* __restrict__ part, const float* __restrict__ bias,
                                                          const float* __restrict__ addend, float* __restrict__ y,
                                                          float* __restrict__ stats, int KS, i64 M, i64 total, int Cout,
                                                          int nblk) {
  __shared__ float red[8];
  const int n = blockIdx.y;
  const i64 e0 = (i64)blockIdx.x * SPLITK_CHUNK;
  i64 e1 = e0 + SPLITK_CHUNK;
  if (e1 > M) e1 = M;
  float s[2] = {0.f, 0.f};
  for (i64 e = e0 + 4 * threadIdx.x; e < e1; e += 1024) {
    const i64 g = (i64)n * M + e;
    float4 acc = *reinterpret_cast<const float4*>(part + g);
    for (int k = 1; k < KS; ++k) {
      const float4 p = *reinterpret_cast<const float4*>(part + (i64)k * total + g);
      acc.x += p.x; acc.y += p.y; acc.z += p.z; acc.w += p.w;
    }
    if (bias) {
      const int c = (int)(e % Cout);
      acc.x += bias[c]; acc.y += bias[c + 1]; acc.z += bias[c + 2]; acc.w += bias[c + 3];
    }
    if (addend) {
      const float4 a = *reinterpret_cast<const float4*>(addend + g);
      acc.x += a.x; acc.y += a.y; acc.z += a.z; acc.w += a.w;
    }
    const seg3d_f32x4 accq = {acc.x, acc.y, acc.z, acc.w};
    Seg3dQuad<OUT_BF>::store(y, g, accq);
    s[0] += (acc.x + acc.y) + (acc.z + acc.w);
    s[1] += (acc.x * acc.x + acc.y * acc.y) + (acc.z * acc.z + acc.w * acc.w);
  }
  block_sum_256<2>(s, red);
  if (stats && threadIdx.x == 0) {
    stats[((i64)n * nblk + blockIdx.x) * 2 + 0] = s[0];
    stats[((i64)n * nblk + blockIdx.x) * 2 + 1] = s[1];
  }
}

__global__ __launch_bounds__(256) void conv3d_splitk_finish_kernel(const float* __restrict__ part,
                                                                     const float* __restrict__ bias,
                                                                     const float* __restrict__ addend, float* __restrict__ y,
                                                                     float* __restrict__ stats, int KS, i64 M, i64 total,
                                                                     int Cout, int nblk) {
  conv3d_splitk_finish_body<false>(part, bias, addend, y, stats, KS, M, total, Cout, nblk);
}

__global__ __launch_bounds__(256) void conv3d_splitk_finish_bf16out_kernel(const float* __restrict__ part,
                                                                             const float* __restrict__ bias,
                                                                             const float* __restrict__ addend,
                                                                             float* __restrict__ y, float* __restrict__ stats,
                                                                             int KS, i64 M, i64 total, int Cout, int nblk) {
  conv3d_splitk_finish_body<true>(part, bias, addend, y, stats, KS, M, total, Cout, nblk);
}

// Pick the output tile for one level by a small time model: a workgroup's duration is proportional to its MFMA row
// blocks per wave (MA) plus a mild halo-staging term; 512 workgroups are resident at once (2 per CU), a partial last
// round runs somewhat faster than a full one (its workgroups have their SIMDs to themselves).  Constraints: tile <= 512
// voxels (4 waves x 4 accumulators), halo tile <= 1280 voxels (staging registers).
static Seg3dTile seg3d_pick_tile(int N, int D, int H, int W, int cout_blocks) {
  const int cand_z[] = {1, 2, 3, 4, 6, 8};
  const int cand_y[] = {2, 3, 4, 6, 8, 12, 16};
  const int cand_x[] = {4, 6, 8, 12, 16, 24, 32};
  Seg3dTile best = {1, 2, 4};
  double best_cost = 1e30;
  for (int tz : cand_z)
    for (int ty : cand_y)
      for (int tx : cand_x) {
        if (tz > D && tz != 1) continue;
        const int mt = tz * ty * tx;
        if (mt > 512 || mt < 32) continue;
        const int nv = (tz + 2) * (ty + 2) * (tx + 2);
        if (2 * nv > SEG3D_MAXE * 256) continue;
        const int ntz = seg3d_cdiv(D, tz), nty = seg3d_cdiv(H, ty), ntx = seg3d_cdiv(W, tx);
        const double wgs = (double)N * ntz * nty * ntx * cout_blocks;
        const int subs = (mt + 31) / 32;
        const int ma = (subs + 3) / 4;
        const double full = floor(wgs / 512.0);
        const double rem = wgs - 512.0 * full;
        const double rounds = full + (rem > 0.0 ? 0.55 + 0.45 * rem / 512.0 : 0.0);
        const double per_wg = (double)ma * (1.0 + 0.06 * ((double)nv / mt - 1.0)) + 0.35;  // + fixed prologue/epilogue
        const double cost = rounds * per_wg;
        if (cost < best_cost) {
          best_cost = cost;
          best = {tz, ty, tx};
        }
      }
  return best;
}

static size_t seg3d_fwd_lds_bytes(const Seg3dTile& t) {
  const int nv = (t.tz + 2) * (t.ty + 2) * (t.tx + 2);
  const int mt = t.tz * t.ty * t.tx;
  return (size_t)(8 * nv + SEG3D_W_CHUNK + ((mt + 3) & ~3)) * 4;
}

// number of K splits: only when the (tile, cout-block) grid cannot fill the chip and K is long enough to cut
static int seg3d_fwd_ksplit(int N, int D, int H, int W, int Cin, int Cout) {
  const int cob = (Cout + 31) / 32, cib = (Cin + 7) / 8;
  Seg3dTile t = seg3d_pick_tile(N, D, H, W, cob);
  const i64 wgs = (i64)N * seg3d_cdiv(D, t.tz) * seg3d_cdiv(H, t.ty) * seg3d_cdiv(W, t.tx) * cob;
  if (wgs >= 192 || cib < 4 || (Cout & 3)) return 1;
  int ks = (int)((512 + wgs - 1) / wgs);
  if (ks > cib / 2) ks = cib / 2;
  if (ks > 16) ks = 16;
  return ks < 2 ? 1 : ks;
}

// ---- second-generation kernel: tile / column-block choice for ONE workgroup per CU -------------------------------
static size_t seg3d_fwd2_lds_bytes(const Seg3dTile& t, int nb) {
  const int nv = (t.tz + 2) * (t.ty + 2) * (t.tx + 2);
  const int mt = t.tz * t.ty * t.tx;
  const int xs = (8 * nv + 255) & ~255;
  (void)mt;
  return (size_t)(2 * (xs + nb * SEG3D_W_CHUNK)) * 4;
}

struct Seg3dFwdPlan {
  int version;  // 1: two workgroups per CU, register-staged (also the split-K path); 2: one per CU, LDS-DMA
  Seg3dTile t;
  int ma, nb, ks;
  int nw;       // waves per workgroup of the version-2 kernel: 4, or 8 (two per SIMD, ma / 2 row blocks each)
};

// time model (cycles): 256 workgroups run at once, every round costs one workgroup's duration =
// K-chunks x 27 taps x 4 x MA x NB MFMAs of 64 cycles + DMA issue + a fixed prologue/epilogue
// With ks > 1 an item covers only ceil(cib / ks) chunks and a finish pass (read ks slabs, write y) is added; that pays
// on the spatially small levels where whole-K items leave most of a round idle.
static bool seg3d_pick_tile_v2(int N, int D, int H, int W, int Cin, int Cout, Seg3dTile* tile, int* ma_out, int* nb_out,
                               int* ks_out, bool bf16 = false) {
  const int cand_ks[] = {1, 2, 3, 4, 6, 8, 12, 16};
  const double out_bytes = (double)N * D * H * W * Cout * 4.0;
  const int cand_z[] = {1, 2, 3, 4, 6, 8};
  const int cand_y[] = {2, 3, 4, 6, 8, 12, 16};
  const int cand_x[] = {4, 6, 8, 12, 16, 24, 32};
  const int cobs = (Cout + 31) / 32, cib = bf16 ? Cin / 16 : (Cin + 7) / 8;
  // cycles of the 27 x 4 (fp32: K = 2 each) or 27 x 1 (bf16: K = 16) MFMAs of one accumulator per chunk
  const double mfma_chunk = bf16 ? 27.0 * 32.0 : 6912.0;
  double best_cost = 1e30;
  bool found = false;
  for (int nb = 1; nb <= 2; ++nb) {
    if (cobs % nb) continue;
    for (int tz : cand_z)
      for (int ty : cand_y)
        for (int tx : cand_x) {
          if (tz > D && tz != 1) continue;
          const int mt = tz * ty * tx;
          if (mt > 128 * (4 / nb) || mt < 32) continue;  // MA * NB <= 4 accumulators per wave (16 MFMAs per tap)
          const int nv = (tz + 2) * (ty + 2) * (tx + 2);
          if (2 * nv > SEG3D_V2_MAXPX * 256) continue;
          Seg3dTile t = {tz, ty, tx};
          if (seg3d_fwd2_lds_bytes(t, nb) > 160 * 1024) continue;
          const int ma = ((mt + 31) / 32 + 3) / 4;
          const double wgs = (double)N * seg3d_cdiv(D, tz) * seg3d_cdiv(H, ty) * seg3d_cdiv(W, tx) * (cobs / nb);
          const double pieces = (((8 * nv + 255) >> 8) + 27 * nb) / 4.0;
          for (int ks : cand_ks) {
            if (ks > 1 && 2 * ks > cib) break;
            const int cpk = (cib + ks - 1) / ks;
            const int slabs = (cib + cpk - 1) / cpk;
            if (slabs != ks) continue;  // this ks leaves an empty slab; a smaller one covers the same split
            const double rounds = ceil(wgs * ks / 256.0);
            // bf16: the chunk is as long as the slower of its MFMAs and its DMA traffic (4 x pieces KiB per workgroup
            // at ~16 B/clk per CU out of L2)
            // bf16: LDS-read bound unless the tile rows are 16 voxels long (bank-conflict-free operand reads, see the
            // kernel): price the operand reads of a chunk (27 taps, 4 waves x (ma + nb) KiB) at 256 B/clk, x 2.5 with conflicts
            const double lds_chunk = 27.0 * 4.0 * (ma + nb) * 1024.0 / 256.0 * ((tx % 16) ? 2.5 : 1.0);
            const double body = bf16 ? fmax(fmax(mfma_chunk * ma * nb, lds_chunk), 4.0 * pieces * 1024.0 / 16.0)
                                     : mfma_chunk * ma * nb;
            const double per_wg = cpk * (body + 60.0 * pieces + 400.0) + 9000.0;
            // finish pass: (ks + 2) x output bytes through ~4 TB/s (1700 B/cycle) + launch and pipeline latency
            const double finish = ks > 1 ? (ks + 2) * out_bytes / 1700.0 + 16000.0 : 0.0;
            const double cost = rounds * per_wg + finish;
            if (cost < best_cost) {
              best_cost = cost;
              *tile = t;
              *ma_out = ma;
              *nb_out = nb;
              *ks_out = ks;
              found = true;
            }
          }
        }
  }
  return found;
}

static Seg3dFwdPlan seg3d_fwd_plan(int N, int D, int H, int W, int Cin, int Cout) {
  Seg3dFwdPlan p;
  p.ks = seg3d_fwd_ksplit(N, D, H, W, Cin, Cout);
  p.version = 1;
  p.nw = 4;
  p.nb = 1;
  p.t = seg3d_pick_tile(N, D, H, W, (Cout + 31) / 32);
  p.ma = ((p.t.tz * p.t.ty * p.t.tx + 31) / 32 + 3) / 4;
  if ((Cin & 7) == 0 && (Cout & 3) == 0) {
    Seg3dTile t2;
    int ma2, nb2, ks2 = 1;
    if (seg3d_pick_tile_v2(N, D, H, W, Cin, Cout, &t2, &ma2, &nb2, &ks2) &&
        (i64)N * seg3d_cdiv(D, t2.tz) * seg3d_cdiv(H, t2.ty) * seg3d_cdiv(W, t2.tx) * ((Cout + 31) / 32 / nb2) * ks2 <
            (1 << 20)) {
      p.version = 2;
      p.t = t2;
      p.ma = ma2;
      p.nb = nb2;
      p.ks = ks2;
      // (fp32: the 384-voxel tiles -- 12 row blocks over 8 waves -- measured 0..2 % slower with 8 waves, bf16 4 % faster)
      if (ks2 == 1 && nb2 == 1 && (ma2 == 2 || ma2 == 4)) p.nw = 8;
    }
  }
  return p;
}

// bf16-input plan: always the second-generation kernel (Cin % 16 == 0, Cout % 4 == 0 checked by the entry point)
static Seg3dFwdPlan seg3d_fwd_plan_bf16(int N, int D, int H, int W, int Cin, int Cout) {
  Seg3dFwdPlan p;
  p.version = 0;
  p.nw = 4;
  p.ks = 1;
  p.ma = p.nb = 1;
  p.t = {1, 1, 1};
  Seg3dTile t2;
  int ma2, nb2, ks2 = 1;
  if ((Cin & 15) == 0 && (Cout & 3) == 0 && seg3d_pick_tile_v2(N, D, H, W, Cin, Cout, &t2, &ma2, &nb2, &ks2, true) &&
      (i64)N * seg3d_cdiv(D, t2.tz) * seg3d_cdiv(H, t2.ty) * seg3d_cdiv(W, t2.tx) * ((Cout + 31) / 32 / nb2) * ks2 <
          (1 << 20)) {
    p.version = 2;
    p.t = t2;
    p.ma = ma2;
    p.nb = nb2;
    p.ks = ks2;
    if (ks2 == 1 && nb2 == 1 && ma2 >= 2) p.nw = 8;
  }
  return p;
}

extern "C" long long seg3d_conv3d_k3_bf16_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int ks = seg3d_fwd_plan_bf16(N, D, H, W, Cin, Cout).ks;
  return ks > 1 ? (long long)ks * N * D * H * W * Cout : 0;
}

extern "C" long long seg3d_conv3d_k3_bf16_stats_count(int N, int D, int H, int W, int Cin, int Cout) {
  const Seg3dFwdPlan p = seg3d_fwd_plan_bf16(N, D, H, W, Cin, Cout);
  if (p.version != 2) return 0;
  if (p.ks > 1) return ((long long)D * H * W * Cout + SPLITK_CHUNK - 1) / SPLITK_CHUNK;
  const long long tiles = (long long)seg3d_cdiv(D, p.t.tz) * seg3d_cdiv(H, p.t.ty) * seg3d_cdiv(W, p.t.tx);
  return tiles * ((Cout + 31) / 32 / p.nb) * p.nw;
}

// 200 + 10 MA + NB of conv3d_k3_mfma2_bf16_kernel<MA, NB> (0: shape not supported)
extern "C" int seg3d_conv3d_k3_bf16_variant(int N, int D, int H, int W, int Cin, int Cout) {
  const Seg3dFwdPlan p = seg3d_fwd_plan_bf16(N, D, H, W, Cin, Cout);
  if (p.version == 2 && p.nw == 8) return 400 + 10 * ((p.ma + 1) / 2) + p.nb;   // conv3d_k3_mfma2w8_bf16_kernel<ceil(MA / 2), NB, .>
  return p.version == 2 ? 200 + 10 * p.ma + p.nb : 0;
}

extern "C" long long seg3d_conv3d_k3_mfma_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int ks = seg3d_fwd_plan(N, D, H, W, Cin, Cout).ks;
  return ks > 1 ? (long long)ks * N * D * H * W * Cout : 0;
}

// GroupNorm partial (sum, sumsq) slots per sample that seg3d_conv3d_k3_mfma_fwd writes for this problem
extern "C" long long seg3d_conv3d_k3_mfma_stats_count(int N, int D, int H, int W, int Cin, int Cout) {
  const Seg3dFwdPlan p = seg3d_fwd_plan(N, D, H, W, Cin, Cout);
  if (p.ks > 1) return ((long long)D * H * W * Cout + SPLITK_CHUNK - 1) / SPLITK_CHUNK;
  const int cob = (Cout + 31) / 32;
  const long long tiles = (long long)seg3d_cdiv(D, p.t.tz) * seg3d_cdiv(H, p.t.ty) * seg3d_cdiv(W, p.t.tx);
  return p.version == 2 ? tiles * (cob / p.nb) * p.nw /* one slot per wave */ : tiles * cob;
}

// which template instantiation a given problem runs (lets profilers attribute time): MA (row blocks per wave, 1..4)
// for the first-generation kernel conv3d_k3_mfma_kernel<MA>; 100 + 10 MA + NB for conv3d_k3_mfma2_kernel<MA, NB>
extern "C" int seg3d_conv3d_k3_mfma_variant(int N, int D, int H, int W, int Cin, int Cout) {
  const Seg3dFwdPlan p = seg3d_fwd_plan(N, D, H, W, Cin, Cout);
  if (p.version == 2 && p.nw == 8) return 300 + 10 * ((p.ma + 1) / 2) + p.nb;   // conv3d_k3_mfma2w8_kernel<ceil(MA / 2), NB>
  return p.version == 2 ? 100 + 10 * p.ma + p.nb : p.ma;
}

template <int MA>
static int launch_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                      int W, int Cin, int Cout, const Seg3dTile& t, hipStream_t s, float* kpart, int ks,
                      const float* addend) {
  const int ntz = seg3d_cdiv(D, t.tz), nty = seg3d_cdiv(H, t.ty), ntx = seg3d_cdiv(W, t.tx);
  const size_t lds = seg3d_fwd_lds_bytes(t);
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma_kernel<MA>), configured, "conv3d_k3_mfma")) return rc;
  const int cib = (Cin + 7) / 8;
  const int cpk = (cib + ks - 1) / ks;
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32), (unsigned)(ks > 1 ? (cib + cpk - 1) / cpk : 1));
  hipLaunchKernelGGL((conv3d_k3_mfma_kernel<MA>), grid, dim3(256), lds, s, x, wp, bias, y, stats, N, D, H, W, Cin, Cout,
                     t.tz, t.ty, t.tx, ntz, nty, ntx, ks > 1 ? kpart : nullptr, cpk, ks > 1 ? nullptr : addend);
  return SEG3D_OK;
}

template <int MA, int NB, bool BF16 = false, bool OUT_BF = false>
static int launch_fwd2(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                       int W, int Cin, int Cout, const Seg3dTile& t, hipStream_t s, const float* addend, float* kpart,
                       int ks) {
  if constexpr (BF16) {  // Cin counts bf16 channels; the kernel sees Cin / 2 words per voxel and 16-channel chunks
    const int ntz = seg3d_cdiv(D, t.tz), nty = seg3d_cdiv(H, t.ty), ntx = seg3d_cdiv(W, t.tx);
    const size_t lds = seg3d_fwd2_lds_bytes(t, NB);
    static Seg3dOncePerDevice configured16;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2_bf16_kernel<MA, NB, OUT_BF>), configured16, "conv3d_k3_mfma2_bf16")) return rc;
    static Seg3dOncePerDevice configured16_b;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2_bf16_splitk_kernel<MA, NB>), configured16_b, "conv3d_k3_mfma2_bf16")) return rc;
    const int ncog = (Cout + 31) / 32 / NB;
    const int nitems = N * ntz * nty * ntx * ncog * ks;
    const int cib = Cin / 16, cpk = (cib + ks - 1) / ks;
    dim3 grid(seg3d_persistent_grid(nitems), 1, 1);
    if (ks > 1)
      hipLaunchKernelGGL((conv3d_k3_mfma2_bf16_splitk_kernel<MA, NB>), grid, dim3(256), lds, s, x, wp, kpart, N, D, H, W,
                         Cin / 2, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, ks, cpk);
    else
      hipLaunchKernelGGL((conv3d_k3_mfma2_bf16_kernel<MA, NB, OUT_BF>), grid, dim3(256), lds, s, x, wp, bias, y, stats, N,
                         D, H, W, Cin / 2, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, addend);
    return SEG3D_OK;
  }
  const int ntz = seg3d_cdiv(D, t.tz), nty = seg3d_cdiv(H, t.ty), ntx = seg3d_cdiv(W, t.tx);
  const size_t lds = seg3d_fwd2_lds_bytes(t, NB);
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2_kernel<MA, NB>), configured, "conv3d_k3_mfma2")) return rc;
  static Seg3dOncePerDevice configured_b;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2_splitk_kernel<MA, NB>), configured_b, "conv3d_k3_mfma2")) return rc;
  const int ncog = (Cout + 31) / 32 / NB;
  const int nitems = N * ntz * nty * ntx * ncog * ks;
  const int cib = Cin / 8, cpk = (cib + ks - 1) / ks;
  dim3 grid(seg3d_persistent_grid(nitems), 1, 1);  // persistent: one workgroup per CU walks the items
  if (ks > 1)
    hipLaunchKernelGGL((conv3d_k3_mfma2_splitk_kernel<MA, NB>), grid, dim3(256), lds, s, x, wp, kpart, N, D, H, W, Cin,
                       Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, ks, cpk);
  else
    hipLaunchKernelGGL((conv3d_k3_mfma2_kernel<MA, NB>), grid, dim3(256), lds, s, x, wp, bias, y, stats, N, D, H, W, Cin,
                       Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, addend);
  return SEG3D_OK;
}

template <int MA, bool OUT_BF>
static int launch_fwd2_w8_bf16(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                               int W, int Cin, int Cout, const Seg3dTile& t, hipStream_t s, const float* addend) {
  const int ntz = seg3d_cdiv(D, t.tz), nty = seg3d_cdiv(H, t.ty), ntx = seg3d_cdiv(W, t.tx);
  const size_t lds = seg3d_fwd2_lds_bytes(t, 1);
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2w8_bf16_kernel<MA, 1, OUT_BF>), configured, "conv3d_k3_mfma2w8_bf16")) return rc;
  const int ncog = (Cout + 31) / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid(seg3d_persistent_grid(nitems), 1, 1);
  hipLaunchKernelGGL((conv3d_k3_mfma2w8_bf16_kernel<MA, 1, OUT_BF>), grid, dim3(512), lds, s, x, wp, bias, y, stats, N, D, H,
                     W, Cin / 2, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, addend);
  return SEG3D_OK;
}

template <int MA>
static int launch_fwd2_w8(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                          int W, int Cin, int Cout, const Seg3dTile& t, hipStream_t s, const float* addend) {
  const int ntz = seg3d_cdiv(D, t.tz), nty = seg3d_cdiv(H, t.ty), ntx = seg3d_cdiv(W, t.tx);
  const size_t lds = seg3d_fwd2_lds_bytes(t, 1);
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_mfma2w8_kernel<MA, 1>), configured, "conv3d_k3_mfma2w8")) return rc;
  const int ncog = (Cout + 31) / 32;
  const int nitems = N * ntz * nty * ntx * ncog;
  dim3 grid(seg3d_persistent_grid(nitems), 1, 1);
  hipLaunchKernelGGL((conv3d_k3_mfma2w8_kernel<MA, 1>), grid, dim3(512), lds, s, x, wp, bias, y, stats, N, D, H, W, Cin,
                     Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ncog, nitems, addend);
  return SEG3D_OK;
}

// x [N][D][H][W][Cin], wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 27), y [N][D][H][W][Cout];
// stats (optional): [N][seg3d_conv3d_k3_mfma_stats_count][2] partial (sum, sumsq) of y per sample.
// addend (optional, same shape as y): y = conv(x) + bias + addend -- used by the data-gradient of the first conv of a
// residual block to fold in the gradient that arrives through the identity path.
extern "C" int seg3d_conv3d_k3_mfma_fwd(const float* x, const float* wp, const float* bias, const float* addend, float* y,
                                        float* stats, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                                        void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k3_mfma_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0, "seg3d_conv3d_k3_mfma_fwd: Cin must be a multiple of 4 (got %d); use the direct kernel", Cin);
  SEG3D_REQUIRE((i64)N * D * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31),
                "seg3d_conv3d_k3_mfma_fwd: tensor exceeds 2^31 elements");
  const Seg3dFwdPlan plan = seg3d_fwd_plan(N, D, H, W, Cin, Cout);
  const Seg3dTile t = plan.t;
  const int ma = plan.ma;
  hipStream_t s = (hipStream_t)stream;
  const int ks = plan.ks;
  SEG3D_REQUIRE(ks == 1 || workspace, "seg3d_conv3d_k3_mfma_fwd: this shape runs split-K and needs the workspace "
                "(seg3d_conv3d_k3_mfma_fwd_workspace_floats)");
  if (plan.version == 2 && plan.nw == 8) {
    const int rc8 = plan.ma >= 3 ? launch_fwd2_w8<2>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend)
                                 : launch_fwd2_w8<1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend);
    if (rc8 != SEG3D_OK) return rc8;
    SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_fwd(v2, 8 waves)");
    return SEG3D_OK;
  }
  if (plan.version == 2) {
    int rc2;
    switch (plan.ma * 10 + plan.nb) {
      case 11: rc2 = launch_fwd2<1, 1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      case 21: rc2 = launch_fwd2<2, 1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      case 31: rc2 = launch_fwd2<3, 1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      case 41: rc2 = launch_fwd2<4, 1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      case 12: rc2 = launch_fwd2<1, 2>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      case 22: rc2 = launch_fwd2<2, 2>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, ks); break;
      default:
        SEG3D_UNSUPPORTED("seg3d_conv3d_k3_mfma_fwd: internal plan error (ma=%d nb=%d)", plan.ma, plan.nb);
    }
    if (rc2 != SEG3D_OK) return rc2;
    SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_fwd(v2)");
    if (ks > 1) {
      const i64 M = (i64)D * H * W * Cout;
      const int nblk = (int)((M + SPLITK_CHUNK - 1) / SPLITK_CHUNK);
      hipLaunchKernelGGL(conv3d_splitk_finish_kernel, dim3(nblk, N), dim3(256), 0, s, workspace, bias, addend, y, stats,
                         ks, M, (i64)N * M, Cout, nblk);
      SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_fwd(v2 split-K finish)");
    }
    return SEG3D_OK;
  }
  int rc;
  switch (ma) {
    case 1: rc = launch_fwd<1>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, workspace, ks, addend); break;
    case 2: rc = launch_fwd<2>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, workspace, ks, addend); break;
    case 3: rc = launch_fwd<3>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, workspace, ks, addend); break;
    case 4: rc = launch_fwd<4>(x, wp, bias, y, stats, N, D, H, W, Cin, Cout, t, s, workspace, ks, addend); break;
    default:
      SEG3D_UNSUPPORTED("seg3d_conv3d_k3_mfma_fwd: internal tile error (ma=%d)", ma);
  }
  if (rc != SEG3D_OK) return rc;
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_fwd");
  if (ks > 1) {
    const i64 M = (i64)D * H * W * Cout;
    const int nblk = (int)((M + SPLITK_CHUNK - 1) / SPLITK_CHUNK);
    const int cib = (Cin + 7) / 8, cpk = (cib + ks - 1) / ks;
    hipLaunchKernelGGL(conv3d_splitk_finish_kernel, dim3(nblk, N), dim3(256), 0, s, workspace, bias, addend, y, stats,
                       (cib + cpk - 1) / cpk, M, (i64)N * M, Cout, nblk);
    SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_fwd(split-K finish)");
  }
  return SEG3D_OK;
}

// bf16 inputs: x [N][D][H][W][Cin] bf16, wp = seg3d_pack_weights_mfma_bf16(A = Cin, B = Cout, T = 27); bias, addend,
// y, stats, workspace fp32 exactly as in seg3d_conv3d_k3_mfma_fwd.  Needs Cin % 16 == 0 and Cout % 4 == 0.
extern "C" int seg3d_conv3d_k3_bf16_fwd(const void* x, const void* wp, const float* bias, const float* addend, void* yv,
                                        float* stats, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                                        int out_bf16, void* stream) {
  float* y = reinterpret_cast<float*>(yv);   // bf16 storage when out_bf16 (the kernels cast back)
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_bf16_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k3_bf16_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 16) == 0 && (Cout % 4) == 0,
                "seg3d_conv3d_k3_bf16_fwd: needs Cin %% 16 == 0 and Cout %% 4 == 0 (got %d, %d)", Cin, Cout);
  SEG3D_REQUIRE((i64)N * D * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31),
                "seg3d_conv3d_k3_bf16_fwd: tensor exceeds 2^31 elements");
  const Seg3dFwdPlan plan = seg3d_fwd_plan_bf16(N, D, H, W, Cin, Cout);
  if (plan.version != 2) SEG3D_UNSUPPORTED("seg3d_conv3d_k3_bf16_fwd: no tile fits this shape");
  const int ks = plan.ks;
  SEG3D_REQUIRE(ks == 1 || workspace, "seg3d_conv3d_k3_bf16_fwd: this shape runs split-K and needs the workspace "
                "(seg3d_conv3d_k3_bf16_fwd_workspace_floats)");
  const float* xw = reinterpret_cast<const float*>(x);
  const float* ww = reinterpret_cast<const float*>(wp);
  const Seg3dTile t = plan.t;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (plan.nw == 8) {
    int rc8;
    if (plan.ma >= 3)
      rc8 = out_bf16 ? launch_fwd2_w8_bf16<2, true>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend)
                     : launch_fwd2_w8_bf16<2, false>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend);
    else
      rc8 = out_bf16 ? launch_fwd2_w8_bf16<1, true>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend)
                     : launch_fwd2_w8_bf16<1, false>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend);
    if (rc8 != SEG3D_OK) return rc8;
    SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_bf16_fwd(8 waves)");
    return SEG3D_OK;
  }
#define SEG3D_BF16_LAUNCH(MA_, NB_)                                                                                    \
  rc = out_bf16 ? launch_fwd2<MA_, NB_, true, true>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend, workspace, \
                                                    ks)                                                               \
                : launch_fwd2<MA_, NB_, true, false>(xw, ww, bias, y, stats, N, D, H, W, Cin, Cout, t, s, addend,      \
                                                     workspace, ks)
  switch (plan.ma * 10 + plan.nb) {
    case 11: SEG3D_BF16_LAUNCH(1, 1); break;
    case 21: SEG3D_BF16_LAUNCH(2, 1); break;
    case 31: SEG3D_BF16_LAUNCH(3, 1); break;
    case 41: SEG3D_BF16_LAUNCH(4, 1); break;
    case 12: SEG3D_BF16_LAUNCH(1, 2); break;
    case 22: SEG3D_BF16_LAUNCH(2, 2); break;
    default:
      SEG3D_UNSUPPORTED("seg3d_conv3d_k3_bf16_fwd: internal plan error (ma=%d nb=%d)", plan.ma, plan.nb);
  }
#undef SEG3D_BF16_LAUNCH
  if (rc != SEG3D_OK) return rc;
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_bf16_fwd");
  if (ks > 1) {
    const i64 M = (i64)D * H * W * Cout;
    const int nblk = (int)((M + SPLITK_CHUNK - 1) / SPLITK_CHUNK);
    if (out_bf16)
      hipLaunchKernelGGL(conv3d_splitk_finish_bf16out_kernel, dim3(nblk, N), dim3(256), 0, s, workspace, bias, addend, y,
                         stats, ks, M, (i64)N * M, Cout, nblk);
    else
      hipLaunchKernelGGL(conv3d_splitk_finish_kernel, dim3(nblk, N), dim3(256), 0, s, workspace, bias, addend, y, stats, ks,
                         M, (i64)N * M, Cout, nblk);
    SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_bf16_fwd(split-K finish)");
  }
  return SEG3D_OK;
}

// ================================================================================================================
// weight gradient
// ================================================================================================================
#define SEG3D_WG_TZ 4
#define SEG3D_WG_TY 4
#define SEG3D_WG_TX 8
#define SEG3D_WG_MT (SEG3D_WG_TZ * SEG3D_WG_TY * SEG3D_WG_TX)                           // 128 voxels
#define SEG3D_WG_HY (SEG3D_WG_TY + 2)
#define SEG3D_WG_HX (SEG3D_WG_TX + 2)
#define SEG3D_WG_NV ((SEG3D_WG_TZ + 2) * SEG3D_WG_HY * SEG3D_WG_HX)                     // 360 halo voxels
#define SEG3D_WG_XE ((SEG3D_WG_NV * 8 + 255) / 256)                                     // float4 per thread: 12
#define SEG3D_WG_YE ((SEG3D_WG_MT * 8) / 256)                                           // 4

// BF (bf16 mode): x and dy are bf16, widened to fp32 when the staged tile is written to LDS (fp32 MFMA, fp32 partials)
template <bool BF>
__device__ __forceinline__ void conv3d_k3_wgrad_mfma_body(const void* __restrict__ x, const void* __restrict__ dy,
                                                          float* __restrict__ part, int N, int D, int H, int W, int Cin,
                                                          int Cout, int ntz, int nty, int ntx, int ntiles, int COB32) {
  __shared__ __attribute__((aligned(16))) float xs[SEG3D_WG_NV * 32];
  __shared__ __attribute__((aligned(16))) float dys[SEG3D_WG_MT * 32];
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int cib = blockIdx.y / COB32, cob = blockIdx.y % COB32;
  const int ci0 = cib * 32, co0 = cob * 32;

  int tapoff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    int tap = wave * 7 + j;
    if (tap > 26) tap = 26;  // idle slot of wave 3 recomputes tap 26 into a discarded accumulator
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    tapoff[j] = ((kz * SEG3D_WG_HY + ky) * SEG3D_WG_HX + kx) * 32;
  }
  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  const int q = tid & 7;  // float4 index inside the 32-channel row
  const bool xq_ok = ci0 + 4 * q < Cin;
  const bool yq_ok = co0 + 4 * q < Cout;

  // Register-prefetch pipeline over the tile loop: the global loads of tile t+1 are issued before the MFMA block of
  // tile t (branch-free, clamped addresses) and written to LDS after it, so HBM/L2 latency hides behind the matrix work.
  // (the zero-select of out-of-range entries happens in store_tile from a bit mask: consuming a loaded value right
  // here would make the compiler wait for each load before issuing the next)
  typename Seg3dQuad<BF>::raw xst[SEG3D_WG_XE], yst[SEG3D_WG_YE];
  unsigned okmask = 0;
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * SEG3D_WG_TZ, y0 = tiy * SEG3D_WG_TY, x0 = tix * SEG3D_WG_TX;
    okmask = 0;
#pragma unroll
    for (int e = 0; e < SEG3D_WG_XE; ++e) {
      const int eidx = tid + e * 256;
      const int v = eidx >> 3;
      const int hx = v % SEG3D_WG_HX;
      const int t = v / SEG3D_WG_HX;
      const int hy = t % SEG3D_WG_HY;
      const int hz = t / SEG3D_WG_HY;
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool ok = eidx < SEG3D_WG_NV * 8 && xq_ok && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      xst[e] = Seg3dQuad<BF>::load(x, ok ? ((((i64)n * D + gz) * H + gy) * W + gx) * Cin + ci0 + 4 * q : (i64)0);
      okmask |= (ok ? 1u : 0u) << e;
    }
#pragma unroll
    for (int e = 0; e < SEG3D_WG_YE; ++e) {
      const int eidx = tid + e * 256;
      const int v = eidx >> 3;
      const int tx = v % SEG3D_WG_TX;
      const int t = v / SEG3D_WG_TX;
      const int ty = t % SEG3D_WG_TY;
      const int tz = t / SEG3D_WG_TY;
      const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
      const bool ok = yq_ok && gz < D && gy < H && gx < W;
      yst[e] = Seg3dQuad<BF>::load(dy, ok ? ((((i64)n * D + gz) * H + gy) * W + gx) * Cout + co0 + 4 * q : (i64)0);
      okmask |= (ok ? 1u : 0u) << (16 + e);
    }
  };
  auto store_tile = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < SEG3D_WG_XE; ++e) {
      const int eidx = tid + e * 256;
      if (eidx < SEG3D_WG_NV * 8)
        *reinterpret_cast<f32x4*>(xs + (eidx >> 3) * 32 + 4 * q) = ((okmask >> e) & 1u) ? Seg3dQuad<BF>::cvt(xst[e]) : zero;
    }
#pragma unroll
    for (int e = 0; e < SEG3D_WG_YE; ++e) {
      const int eidx = tid + e * 256;
      *reinterpret_cast<f32x4*>(dys + (eidx >> 3) * 32 + 4 * q) =
          ((okmask >> (16 + e)) & 1u) ? Seg3dQuad<BF>::cvt(yst[e]) : zero;
    }
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();  // previous tile fully consumed
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);

#pragma unroll 4
    for (int kp = 0; kp < SEG3D_WG_MT / 2; ++kp) {
      const int v = 2 * kp + lh;
      const int tx = v % SEG3D_WG_TX;
      const int t = v / SEG3D_WG_TX;
      const int ty = t % SEG3D_WG_TY;
      const int tz = t / SEG3D_WG_TY;
      const int base = ((tz * SEG3D_WG_HY + ty) * SEG3D_WG_HX + tx) * 32 + li;
      const float bvv = dys[v * 32 + li];
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        const float a = xs[base + tapoff[j]];
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc[j], 0, 0, 0);
      }
    }
  }

  // part[slab = blockIdx.x][pair = blockIdx.y][tap][ci row][co col]
  float* dst = part + ((i64)blockIdx.x * gridDim.y + blockIdx.y) * 27 * 1024;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = wave * 7 + j;
    if (tap < 27) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[tap * 1024 + mfma_row(r, lh) * 32 + li] = acc[j][r];
    }
  }
}

__global__ __launch_bounds__(256, 2) void conv3d_k3_wgrad_mfma_bf16_kernel(const void* __restrict__ x,
                                                                             const void* __restrict__ dy,
                                                                             float* __restrict__ part, int N, int D, int H,
                                                                             int W, int Cin, int Cout, int ntz, int nty,
                                                                             int ntx, int ntiles, int COB32) {
  conv3d_k3_wgrad_mfma_body<true>(x, dy, part, N, D, H, W, Cin, Cout, ntz, nty, ntx, ntiles, COB32);
}

// ----------------------------------------------------------------------------------------------------------------
// Weight gradient, second generation (conv3d_k3_wgrad2_kernel): the same recipe as conv3d_k3_mfma2_kernel.
//   ONE persistent workgroup per CU (one wave per SIMD); a workgroup owns one 32-channel block of x, NB 32-channel
//   blocks of dy and one spatial slab (tiles slab, slab + slabs, ...), keeping its 7 x NB accumulators per wave (taps
//   7w..7w+6) in registers across all its tiles.  The next tile (x halo tile [360][32] + dy tile [NB][128][32]) arrives
//   by LDS-DMA into the second LDS buffer while the current one is multiplied: one 1-KiB piece every third k-pair,
//   its address arithmetic done right there in the shadow of the MFMAs (no integer division, no staging registers,
//   no ds_write pass); one barrier per tile.  The x tile is staged once for NB*32 output channels.
// ----------------------------------------------------------------------------------------------------------------
// Tile shapes: 4x4x8 (default), and 4x4x4 / 2x6x6 for levels whose extent they divide where 4x4x8 would leave half-empty
// tiles (12^3, 6^3): every tile then takes the fast DMA path and no MFMA row is wasted.
template <int NB, int TZ, int TY, int TX>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad2_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                    float* __restrict__ part, int N, int D, int H, int W,
                                                                    int Cin, int Cout, int ntz, int nty, int ntx, int ntiles,
                                                                    int slabs, int COG) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HY = TY + 2, HX = TX + 2;
  constexpr int NVH = (TZ + 2) * HY * HX;              // halo voxels of the x tile
  constexpr int MTV = TZ * TY * TX;                    // voxels of the dy tile
  constexpr int XPC = NVH / 8, YPC = MTV / 8;          // 1-KiB DMA pieces (8 voxels x 32 channels) of x / of one dy block
  static_assert(NVH % 8 == 0 && MTV % 8 == 0 && TX % 2 == 0, "tile shape");
  constexpr int XS = NVH * 32;                         // floats
  constexpr int BUF = XS + NB * MTV * 32;
  constexpr int NP = XPC + NB * YPC;  // pieces per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x % slabs;
  const int pg = blockIdx.x / slabs;                   // (ci block, co group)
  const int cib = pg / COG, cog = pg % COG;
  const int ci0 = cib * 32, co0 = cog * NB * 32;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };  // exact for the small ranges used here

  int tapoff[7];
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    int tap = wave * 7 + j;
    if (tap > 26) tap = 26;  // idle slot of wave 3 recomputes tap 26 into a discarded accumulator
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    tapoff[j] = ((kz * HY + ky) * HX + kx) * 32;
  }
  f32x16 acc[7][NB];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int q = 0; q < NB; ++q)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[j][q][r] = 0.f;

  // one DMA piece of `tile` into `buf`: piece p = wave + 4 g; lane -> voxel 8 p' + (lane >> 3), channels 4 (lane & 7)..+3
  const int lv = lane >> 3, lq = lane & 7;
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
  auto issue_piece = [&](int g, float* buf) {
    const int p = wave + 4 * g;
    if (p < XPC) {
      const int v = p * 8 + lv;                                   // halo voxel: (hz, hy, hx)
      const int t = fdiv(v, 1.0f / (float)HX);
      const int hx = v - t * HX;
      const int hz = fdiv(t, 1.0f / (float)HY);
      const int hy = t - hz * HY;
      const int gz = tz0 + hz - 1, gy = ty0 + hy - 1, gx = tx0 + hx - 1;
      const bool ok = ci0 + 4 * lq < Cin && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const float* src = ok ? x + ((i64)(((tn * D + gz) * H + gy) * W + gx) * Cin + ci0 + 4 * lq) : seg3d_zero16;
      seg3d_glds16(src, buf + p * 256);
    } else if (p < NP) {
      const int pp = p - XPC;
      const int nb = pp / YPC;
      const int v = (pp - nb * YPC) * 8 + lv;            // tile voxel: (tz, ty, tx)
      const int gz = tz0 + v / (TX * TY), gy = ty0 + (v / TX) % TY, gx = tx0 + v % TX;
      const int co = co0 + nb * 32 + 4 * lq;
      const bool ok = co < Cout && gz < D && gy < H && gx < W;
      const float* src = ok ? dy + ((i64)(((tn * D + gz) * H + gy) * W + gx) * Cout + co) : seg3d_zero16;
      seg3d_glds16(src, buf + XS + pp * 256);
    }
  };
  constexpr int NG = (NP + 3) / 4;  // piece groups per wave per tile (<= 20): one every third k-pair

  // Fast path for tiles that lie completely inside the volume (all of them when D, H, W are multiples of 4, 4, 8):
  // a piece's source is  tile origin (uniform)  +  a tile-invariant per-lane offset, and it is zero padding exactly
  // when its halo face lies outside the volume -- 5 instructions per piece instead of ~40.
  int prel[NG], pflag[NG];  // float offset from the tile-origin voxel; face bits (bit 6: channel slice out of range)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int p = wave + 4 * g;
    prel[g] = 0;
    pflag[g] = -1;
    if (p < XPC) {
      const int v = p * 8 + lv;
      const int t = fdiv(v, 1.0f / (float)HX);
      const int hx = v - t * HX;
      const int hz = fdiv(t, 1.0f / (float)HY);
      const int hy = t - hz * HY;
      prel[g] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 4 * lq;
      pflag[g] = (hz == 0 ? 1 : 0) | (hz == TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TY + 1 ? 8 : 0) |
                 (hx == 0 ? 16 : 0) | (hx == TX + 1 ? 32 : 0) | (ci0 + 4 * lq < Cin ? 0 : 64);
    } else if (p < NP) {
      const int pp = p - XPC;
      const int nb = pp / YPC;
      const int v = (pp - nb * YPC) * 8 + lv;
      const int co = co0 + nb * 32 + 4 * lq;
      prel[g] = (((v / (TX * TY)) * H + (v / TX) % TY) * W + v % TX) * Cout + co;
      pflag[g] = co < Cout ? 0 : 64;
    }
  }
  auto issue_piece_fast = [&](int g, float* buf, const float* xbase, const float* ybase, int faces) {
    const int p = wave + 4 * g;
    if (p < NP) {
      const float* base = p < XPC ? xbase : ybase;  // uniform
      const float* src = (pflag[g] & faces) ? seg3d_zero16 : base + prel[g];
      seg3d_glds16(src, buf + p * 256);
    }
  };

  // tile walk: plain stride, or (SEG3D_XCD_WALK, slabs % 8 == 0) XCD-contiguous as in the forward kernels
  int tile = slab, tstride = slabs, tlimit = ntiles;
  if (SEG3D_XCD_WALK && (slabs & 7) == 0) {
    const int per_xcd = (ntiles + 7) >> 3, xcd = slab & 7;
    tile = xcd * per_xcd + (slab >> 3);
    tstride = slabs >> 3;
    tlimit = (xcd + 1) * per_xcd < ntiles ? (xcd + 1) * per_xcd : ntiles;
  }
  if (tile < tlimit) {
    set_tile(tile);
#pragma unroll 1
    for (int g = 0; g < NG; ++g) issue_piece(g, lds);
  }
  __syncthreads();
  int parity = 0;
#ifdef SEG3D_STAMPS
  int stamp_k = 0;
#endif
  for (; tile < tlimit; tile += tstride) {
#ifdef SEG3D_STAMPS
    if (stamp_k < 5) SEG3D_STAMP(blockIdx.x, 3 * stamp_k);
#endif
    const float* cur = lds + parity * BUF;
    float* nxt = lds + (parity ^ 1) * BUF;
    const bool more = tile + tstride < tlimit;
    if (more) set_tile(tile + tstride);
    // the tile being fetched: inside the volume?  which of its halo faces stick out?
    const bool regular = tz0 + TZ <= D && ty0 + TY <= H && tx0 + TX <= W;
    const int faces = 64 | (tz0 == 0 ? 1 : 0) | (tz0 + TZ >= D ? 2 : 0) | (ty0 == 0 ? 4 : 0) |
                      (ty0 + TY >= H ? 8 : 0) | (tx0 == 0 ? 16 : 0) | (tx0 + TX >= W ? 32 : 0);
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    const float* xbase = x + origin * Cin;
    const float* ybase = dy + origin * Cout;
    if (more && !regular) {  // tiles sticking out of the volume (dims not multiples of 4, 4, 8): generic path, up front
#pragma unroll 1
      for (int g = 0; g < NG; ++g) issue_piece(g, nxt);
    }
    const float* xa = cur + lh * 32 + li;          // + voxel * 32 + tap offset
    const float* yb = cur + XS + lh * 32 + li;     // + nb * 4096 + voxel * 32
    // voxel pair (2 kp, 2 kp + 1) lies in one row of the tile (TX even), so the lane half only shifts by one voxel.
    // Operands of pair kp+1 are read while pair kp is multiplied (one wave per SIMD: nothing else hides LDS latency).
    auto xoff = [](int kp) {
      const int v0 = 2 * kp;
      return (((v0 / (TX * TY)) * HY + (v0 / TX) % TY) * HX + v0 % TX) * 32;
    };
    // two pairs ahead: the scheduler interleaves reads and MFMAs one to one, so a distance of one pair leaves a read
    // only ~2 MFMAs of cover
    float a1[7], b1[NB], a2[7], b2[NB];
#pragma unroll
    for (int q = 0; q < NB; ++q) {
      b1[q] = yb[q * MTV * 32];
      b2[q] = yb[q * MTV * 32 + 2 * 32];
    }
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      a1[j] = xa[xoff(0) + tapoff[j]];
      a2[j] = xa[xoff(1) + tapoff[j]];
    }
#pragma unroll
    for (int kp = 0; kp < MTV / 2; ++kp) {
      float a[7], bvv[NB];
#pragma unroll
      for (int q = 0; q < NB; ++q) {
        bvv[q] = b1[q];
        b1[q] = b2[q];
      }
#pragma unroll
      for (int j = 0; j < 7; ++j) {
        a[j] = a1[j];
        a1[j] = a2[j];
      }
      if (kp + 2 < MTV / 2) {
#pragma unroll
        for (int q = 0; q < NB; ++q) b2[q] = yb[q * MTV * 32 + (2 * kp + 4) * 32];
#pragma unroll
        for (int j = 0; j < 7; ++j) a2[j] = xa[xoff(kp + 2) + tapoff[j]];
      }
#ifndef SEG3D_EXPERIMENT_NODMA
      if (more && regular && kp % 3 == 0 && kp / 3 < NG) issue_piece_fast(kp / 3, nxt, xbase, ybase, faces);
#endif
#pragma unroll
      for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int q = 0; q < NB; ++q) acc[j][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], bvv[q], acc[j][q], 0, 0, 0);
      if (NB > 1) __builtin_amdgcn_sched_barrier(0);  // keeps the scheduler from hoisting reads of many pairs (spills)
    }
#ifdef SEG3D_STAMPS
    if (stamp_k < 5) SEG3D_STAMP(blockIdx.x, 3 * stamp_k + 1);
#endif
    if (more) __syncthreads();  // own DMAs landed (vmcnt(0)), everyone done with `cur`, `nxt` visible
#ifdef SEG3D_STAMPS
    if (stamp_k < 5) SEG3D_STAMP(blockIdx.x, 3 * stamp_k + 2);
    ++stamp_k;
#endif
    parity ^= 1;
  }

  // part[slab][pair = cib * COB32 + cob][tap][ci row][co col]
  const int COB32 = COG * NB;
#pragma unroll
  for (int q = 0; q < NB; ++q) {
    const int pair = cib * COB32 + cog * NB + q;
    float* dst = part + ((i64)slab * (COB32 * ((Cin + 31) / 32)) + pair) * 27 * 1024;
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      const int tap = wave * 7 + j;
      if (tap < 27) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dst[tap * 1024 + mfma_row(r, lh) * 32 + li] = acc[j][q][r];
      }
    }
  }
}

// ----------------------------------------------------------------------------------------------------------------
// Weight gradient on the bf16 matrix cores (conv3d_k3_wgrad3_bf16_kernel), bf16 mode.
//   dW[tap][ci][co] = sum_v x[v + tap][ci] dy[v][co]: M = ci, N = co, K = voxels, v_mfma_f32_32x32x16_bf16.  Both operands
//   want 8 consecutive K values (voxels) of ONE channel per lane, while memory (and an LDS-DMA image of it) is
//   [voxel][channel]: the hardware transposing read ds_read_b64_tr_b16 closes the gap -- per 16-lane group it takes four
//   rows (voxels, each with its own address: the tap shift needs no alignment) x 16 channels and hands every lane one
//   channel's four voxels.  Two such reads make one operand.
//   Structure as conv3d_k3_wgrad2_kernel: one persistent workgroup per CU, 7 accumulators per wave (taps 7w..7w+6) kept
//   across the tiles of its slab, tiles (x halo tile [NVH][32] + dy tile [MTV][32], 64-byte rows) arriving by LDS-DMA,
//   one barrier per tile.  Per 16-voxel step a wave issues 16 transposing reads and 7 MFMAs: a tile is multiplied in
//   ~1800 cycles, LESS than one trip to memory, so a single tile of lookahead (the fp32 kernels' scheme) left the kernel
//   waiting for its DMAs at every barrier (measured 4600 cycles per tile).  Hence a ring of 4 LDS buffers: while tile t
//   is multiplied, tiles t+1 and t+2 are landing and the pieces of t+3 are issued one per step; the barrier of a tile is
//   preceded by a COUNTED s_waitcnt vmcnt that leaves the younger tiles' DMAs in flight (every wave issues exactly NG
//   pieces per tile -- the ring slots are padded to whole groups -- so the count is a constant).
//   Tiles that stick out of the volume (IRR instantiation) pad by the voxel's coordinates; channel counts that are not
//   multiples of 8 take the register-staged kernel.
// ----------------------------------------------------------------------------------------------------------------
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int N, int I = 0, class F>
__device__ __forceinline__ void seg3d_static_for(F&& f) {   // f(integral_constant<int, I>) for I = 0 .. N-1
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    seg3d_static_for<N, I + 1>(f);
  }
}

#define SEG3D_WG3_NBUF 4
#ifndef SEG3D_WG3_EXP_NODMA   // measurement builds only (tools/ubench/README.md): 1 = no DMA in the steady state
#define SEG3D_WG3_EXP_NODMA 0
#endif
#ifndef SEG3D_WG3_EXP_NOMFMA  // 1 = skip the MFMAs (reads and DMA only)
#define SEG3D_WG3_EXP_NOMFMA 0
#endif
// s_waitcnt with only vmcnt set (gfx9 encoding: vmcnt = bits [3:0] and [15:14], expcnt [6:4], lgkmcnt [11:8])
#define SEG3D_WAIT_VMCNT(n) __builtin_amdgcn_s_waitcnt((((n) & 15) | (((n) >> 4) << 14) | (7 << 4) | (15 << 8)))

// The transposing reads are issued as inline asm: through the builtin the compiler treats them as LDS reads that may
// alias the LDS-DMA writes in flight and puts an s_waitcnt vmcnt(0) in front of every step, which serialises the ring.
// The asm form is invisible to that analysis; the price is a hand-placed s_waitcnt lgkmcnt(0) (SEG3D_TR_WAIT, tied to
// the operand registers so that no MFMA can be scheduled above it).
#define SEG3D_TR_READ(dst, addr, imm) \
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
__device__ __forceinline__ bf16x8 seg3d_tr_join(s16x4 lo, s16x4 hi) {
  return __builtin_bit_cast(bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

// IRR: tiles may stick out of the volume (D, H, W not multiples of the tile, e.g. the 6^3 level): a piece is zero padding
// when its voxel lies outside the volume, tested on the voxel's coordinates instead of the tile's halo-face bits
template <int TZ, int TY, int TX, bool IRR>
__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad3_bf16_kernel(const seg3d_bf16* __restrict__ x,
                                                                         const seg3d_bf16* __restrict__ dy,
                                                                         float* __restrict__ part, int N, int D, int H,
                                                                         int W, int Cin, int Cout, int ntz, int nty, int ntx,
                                                                         int ntiles, int slabs, int COB32) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  constexpr int HY = TY + 2, HX = TX + 2;
  constexpr int NVH = (TZ + 2) * HY * HX;              // halo voxels of the x tile
  constexpr int MTV = TZ * TY * TX;                    // voxels of the dy tile
  static_assert(MTV % 16 == 0 && (TX == 8 || TX == 4) && TY % 2 == 0, "tile shape");
  constexpr int XPC = (NVH + 15) / 16, YPC = MTV / 16; // 1-KiB DMA pieces (16 voxels x 32 bf16 channels)
  constexpr int NP = XPC + YPC;
  constexpr int NG = (NP + 3) / 4;                     // piece groups per wave per tile
  constexpr int BUFB = NG * 4 * 1024;                  // bytes per ring slot (padded to whole groups: see the header)
  constexpr int NBUF = SEG3D_WG3_NBUF;
  constexpr int STEPS = MTV / 16;                      // K steps (16 voxels) per tile
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x % slabs;
  const int pg = blockIdx.x / slabs;                   // (ci block, co block)
  const int cib = pg / COB32, cob = pg % COB32;
  const int ci0 = cib * 32, co0 = cob * 32;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };

  int tapoff[7];   // bytes
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    int tap = wave * 7 + j;
    if (tap > 26) tap = 26;  // idle slot of wave 3 recomputes tap 26 into a discarded accumulator
    const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
    tapoff[j] = ((kz * HY + ky) * HX + kx) * 64;
  }
  f32x16 acc[7];
#pragma unroll
  for (int j = 0; j < 7; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // transposing-read addresses of this lane (bytes): group of 16 lanes = (channel half, K half); lane 4q + p of a group
  // supplies row q (a voxel), channels 4p .. 4p+3 of the group's 16
  const int l16 = lane & 15, rq = l16 >> 2, rp = l16 & 3;
  const int colb = (16 * ((lane >> 4) & 1) + 4 * rp) * 2;
  const int r0 = 8 * lh + rq, r1 = r0 + 4;             // voxel of the 16-voxel step
  const int xa0 = ((r0 / TX) * HX + (r0 % TX)) * 64 + colb, xa1 = ((r1 / TX) * HX + (r1 % TX)) * 64 + colb;
  const int yb0 = XPC * 1024 + r0 * 64 + colb, yb1 = XPC * 1024 + r1 * 64 + colb;

  // DMA pieces: piece p = wave + 4 g; lane -> voxel 16 p' + (lane >> 2), channels 8 (lane & 3) .. +7 (16 bytes)
  const int lv = lane >> 2, lq = lane & 3;
  int prel[NG], pflag[NG];  // bf16-element offset from the tile-origin voxel; halo-face bits (bit 6: always zero)
  int pcoord[NG];           // IRR: halo coordinates hz | hy << 8 | hx << 16 of the piece's voxel (dy pieces: tile coords + 1)
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int p = wave + 4 * g;
    prel[g] = 0;
    pflag[g] = 64;
    pcoord[g] = 0;
    if (p < XPC) {
      const int v = p * 16 + lv;
      if (v < NVH) {
        const int hx = v % HX, hy = (v / HX) % HY, hz = v / (HX * HY);
        prel[g] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 8 * lq;
        pflag[g] = (hz == 0 ? 1 : 0) | (hz == TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TY + 1 ? 8 : 0) |
                   (hx == 0 ? 16 : 0) | (hx == TX + 1 ? 32 : 0) | (ci0 + 8 * lq < Cin ? 0 : 64);
        pcoord[g] = hz | (hy << 8) | (hx << 16);
      }
    } else if (p < NP) {
      const int v = (p - XPC) * 16 + lv;
      const int co = co0 + 8 * lq;
      prel[g] = (((v / (TX * TY)) * H + (v / TX) % TY) * W + v % TX) * Cout + co;
      pflag[g] = co < Cout ? 0 : 64;
      pcoord[g] = (v / (TX * TY) + 1) | (((v / TX) % TY + 1) << 8) | ((v % TX + 1) << 16);
    }                                                   // p >= NP: padding piece (zeros into the slot's tail)
  }
  int tn = 0, tz0 = 0, ty0 = 0, tx0 = 0;
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
  auto issue_piece = [&](int g, char* buf, const seg3d_bf16* xbase, const seg3d_bf16* ybase, int faces) {
    const int p = wave + 4 * g;                          // every wave issues all NG pieces (counted vmcnt waits)
    const seg3d_bf16* base = p < XPC ? xbase : ybase;    // uniform
    bool pad = (pflag[g] & faces) != 0;
    if (IRR) {   // (faces only carries bit 6 here: the coordinate test covers halo faces and the overhang alike)
      const int gz = tz0 + (pcoord[g] & 255) - 1, gy = ty0 + ((pcoord[g] >> 8) & 255) - 1, gx = tx0 + (pcoord[g] >> 16) - 1;
      pad = pad || (unsigned)gz >= (unsigned)D || (unsigned)gy >= (unsigned)H || (unsigned)gx >= (unsigned)W;
    }
    const void* src = pad ? (const void*)seg3d_zero16 : (const void*)(base + prel[g]);
    seg3d_glds16(reinterpret_cast<const float*>(src), reinterpret_cast<float*>(buf + p * 1024));
  };
  auto tile_sources = [&](const seg3d_bf16*& xbase, const seg3d_bf16*& ybase, int& faces) {
    faces = IRR ? 64
                : 64 | (tz0 == 0 ? 1 : 0) | (tz0 + TZ >= D ? 2 : 0) | (ty0 == 0 ? 4 : 0) | (ty0 + TY >= H ? 8 : 0) |
                      (tx0 == 0 ? 16 : 0) | (tx0 + TX >= W ? 32 : 0);
    const i64 origin = ((i64)(tn * D + tz0) * H + ty0) * W + tx0;
    xbase = x + origin * Cin;
    ybase = dy + origin * Cout;
  };

  char* ldsb = reinterpret_cast<char*>(lds);
  const unsigned lds_addr = (unsigned)(size_t)((__attribute__((address_space(3))) char*)ldsb);   // LDS byte offset
  // Tile walk.  Workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD with its own 4 MB L2, and the
  // halo tiles of neighbouring tiles overlap by 2.8x: XCD j therefore owns the CONTIGUOUS eighth j of the tile list and its
  // slabs / 8 workgroups walk it side by side (tile = first_j + k * (slabs / 8) + local), so that the x / y neighbours of
  // a tile are fetched through the same L2 at about the same time.  (slabs % 8 != 0: plain strided walk.)
  const bool xcd_walk = (slabs & 7) == 0;
  const int per_xcd = (ntiles + 7) >> 3, wg_per_xcd = slabs >> 3;
  const int xcd = slab & 7, local = slab >> 3;
  const int first = xcd_walk ? xcd * per_xcd + local : slab;
  const int stride = xcd_walk ? wg_per_xcd : slabs;
  int limit = ntiles;
  if (xcd_walk && (xcd + 1) * per_xcd < ntiles) limit = (xcd + 1) * per_xcd;
  // ring: tile number k of this workgroup (tile index first + k * stride) lives in slot k % NBUF
  const int mytiles = first < limit ? (limit - first + stride - 1) / stride : 0;
  auto issue_tile_all = [&](int k) {   // prologue only: all pieces of tile k back to back
    set_tile(first + k * stride);
    const seg3d_bf16 *xb, *yb;
    int faces;
    tile_sources(xb, yb, faces);
#pragma unroll
    for (int g = 0; g < NG; ++g) issue_piece(g, ldsb + (k % NBUF) * BUFB, xb, yb, faces);
  };
  for (int k = 0; k < NBUF - 1 && k < mytiles; ++k) issue_tile_all(k);
  for (int k = 0; k < mytiles; ++k) {
    // tile k has landed once at most the DMAs of the younger tiles in flight (k+1 .. min(k + NBUF - 2, last)) are pending
    const int younger = (mytiles - 1 - k) < (NBUF - 2) ? (mytiles - 1 - k) : (NBUF - 2);
    if (younger >= 2) SEG3D_WAIT_VMCNT(2 * NG);
    else if (younger == 1) SEG3D_WAIT_VMCNT(NG);
    else SEG3D_WAIT_VMCNT(0);
    __builtin_amdgcn_s_barrier();      // everyone's pieces of tile k landed; everyone done with tile k - 1 (its slot is free)
    const int kn = k + NBUF - 1;       // tile whose pieces are issued during this one, into the slot of tile k - 1
    const bool more = kn < mytiles;
    char* nxt = ldsb + (kn % NBUF) * BUFB;
    const seg3d_bf16 *xb = x, *yb = dy;
    int faces = 64;
    if (more) {
      set_tile(first + kn * stride);
      tile_sources(xb, yb, faces);
    }
    // LDS byte addresses of this tile's reads: per-lane base + a compile-time offset per step
    const unsigned cb = lds_addr + (unsigned)((k % NBUF) * BUFB);
    unsigned xad0[7], xad1[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
      xad0[j] = cb + (unsigned)(tapoff[j] + xa0);
      xad1[j] = cb + (unsigned)(tapoff[j] + xa1);
    }
    const unsigned yad0 = cb + (unsigned)yb0, yad1 = cb + (unsigned)yb1;
    // software pipeline: the 16 reads of step st + 1 are issued before the 7 MFMAs of step st and waited for after them
    // (macros, not lambdas: the asm offsets must be integer constant expressions)
    s16x4 lo[2][8], hi[2][8];   // [pipeline slot][0..6 = taps, 7 = dy]
#define SEG3D_WG3_XO(ST) ((((16 * (ST) / (TX * TY)) * HY + (16 * (ST) / TX) % TY) * HX + (16 * (ST)) % TX) * 64)
#define SEG3D_WG3_READS(ST, SLOT)                                                   \
  do {                                                                              \
    SEG3D_TR_READ(lo[SLOT][7], yad0, 16 * (ST) * 64);                               \
    SEG3D_TR_READ(hi[SLOT][7], yad1, 16 * (ST) * 64);                               \
    SEG3D_TR_READ(lo[SLOT][0], xad0[0], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][0], xad1[0], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][1], xad0[1], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][1], xad1[1], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][2], xad0[2], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][2], xad1[2], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][3], xad0[3], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][3], xad1[3], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][4], xad0[4], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][4], xad1[4], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][5], xad0[5], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][5], xad1[5], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(lo[SLOT][6], xad0[6], SEG3D_WG3_XO(ST));                          \
    SEG3D_TR_READ(hi[SLOT][6], xad1[6], SEG3D_WG3_XO(ST));                          \
  } while (0)
// The reads are a continuous in-order stream, per step: dy, tap 0, tap 1 .. tap 6 (2 reads each).  The reads of
// (step + 1, tap j) go out right after MFMA (step, j) -- dy with tap 0 -- and an MFMA waits with a COUNTED lgkmcnt that
// leaves everything younger than its own operands in flight (14 reads, 12 for tap 0; draining to lgkmcnt(0) every 16 reads
// ran the LDS at a third of its rate with one wave per SIMD).  The empty asm ties the wait to the operand registers.
#define SEG3D_WG3_LGKM(n) __builtin_amdgcn_s_waitcnt((15 | (3 << 14) | (7 << 4) | ((n) << 8)))
#define SEG3D_WG3_TIE(SLOT, J) asm volatile("" : "+v"(lo[SLOT][J]), "+v"(hi[SLOT][J]))
#define SEG3D_WG3_MF(ST, J)                                                                                           \
  SEG3D_WG3_LGKM(((ST) + 1 < STEPS) ? ((J) == 0 ? 12 : 14) : 2 * (6 - (J)));                                          \
  SEG3D_WG3_TIE((ST) & 1, J);                                                                                         \
  if ((J) == 0) {                                                                                                     \
    SEG3D_WG3_TIE((ST) & 1, 7);                                                                                       \
    bop = seg3d_tr_join(lo[(ST) & 1][7], hi[(ST) & 1][7]);                                                            \
  }                                                                                                                   \
  if (!SEG3D_WG3_EXP_NOMFMA || (ST) == 0)                                                                             \
    acc[J] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(seg3d_tr_join(lo[(ST) & 1][J], hi[(ST) & 1][J]), bop, acc[J], 0, \
                                                     0, 0);                                                           \
  if constexpr ((ST) + 1 < STEPS) {                                                                                   \
    if ((J) == 0) {                                                                                                   \
      SEG3D_TR_READ(lo[((ST) + 1) & 1][7], yad0, 16 * ((ST) + 1) * 64);                                               \
      SEG3D_TR_READ(hi[((ST) + 1) & 1][7], yad1, 16 * ((ST) + 1) * 64);                                               \
    }                                                                                                                 \
    SEG3D_TR_READ(lo[((ST) + 1) & 1][J], xad0[J], SEG3D_WG3_XO((ST) + 1));                                            \
    SEG3D_TR_READ(hi[((ST) + 1) & 1][J], xad1[J], SEG3D_WG3_XO((ST) + 1));                                            \
  }
#define SEG3D_WG3_STEP(ST)                                                                                            \
  if constexpr ((ST) < STEPS) {                                                                                       \
    SEG3D_WG3_MF(ST, 0) SEG3D_WG3_MF(ST, 1) SEG3D_WG3_MF(ST, 2) SEG3D_WG3_MF(ST, 3)                                   \
    if (more && !SEG3D_WG3_EXP_NODMA) { /* the pieces of tile k + NBUF - 1, spread over the steps */                  \
      _Pragma("unroll") for (int g = ((ST) * NG) / STEPS; g < (((ST) + 1) * NG) / STEPS; ++g)                         \
          issue_piece(g, nxt, xb, yb, faces);                                                                         \
    }                                                                                                                 \
    SEG3D_WG3_MF(ST, 4) SEG3D_WG3_MF(ST, 5) SEG3D_WG3_MF(ST, 6)                                                       \
  }
    bf16x8 bop;
    SEG3D_WG3_READS(0, 0);   // dy, tap 0 .. tap 6 of step 0
    SEG3D_WG3_STEP(0) SEG3D_WG3_STEP(1) SEG3D_WG3_STEP(2) SEG3D_WG3_STEP(3)
    SEG3D_WG3_STEP(4) SEG3D_WG3_STEP(5) SEG3D_WG3_STEP(6) SEG3D_WG3_STEP(7)
    static_assert(STEPS <= 8, "steps");
#undef SEG3D_WG3_STEP
#undef SEG3D_WG3_MF
#undef SEG3D_WG3_TIE
#undef SEG3D_WG3_LGKM
#undef SEG3D_WG3_READS
#undef SEG3D_WG3_XO
  }

  // part[slab][pair = cib * COB32 + cob][tap][ci row][co col]
  const int pair = cib * COB32 + cob;
  float* dst = part + ((i64)slab * (COB32 * ((Cin + 31) / 32)) + pair) * 27 * 1024;
#pragma unroll
  for (int j = 0; j < 7; ++j) {
    const int tap = wave * 7 + j;
    if (tap < 27) {
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[tap * 1024 + mfma_row(r, lh) * 32 + li] = acc[j][r];
    }
  }
}

// dw[a*sa + b*sb + t] = sum_slab part[slab][a/32][b/32][t][a%32][b%32]   (a = reduction-side channel, b = output channel)
// A lane owns FOUR consecutive outputs (one 16-byte load per slab), a wave reads 1 KB per slab, the G waves of a workgroup
// take the slabs k = g, g + G, ..; 64 lanes x G partial quads are combined through LDS in a fixed order (bitwise
// reproducible).  Round 1's form (one float per lane, 4 slab groups: 64 dependent 4-byte loads per thread) ran the
// 28 MB of a 256-slab reduction at 1.8 TB/s (15.6 us per launch, 18 launches per step).
template <int G>
__global__ __launch_bounds__(64 * G) void conv3d_k3_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                          int slabs, int A, int B, int BB32, int npairs, i64 sa,
                                                                          i64 sb, int accumulate) {
  constexpr int T = 27;
  __shared__ f32x4 red[G * 64];
  const i64 totalq = (i64)npairs * T * 256;                     // float4 quads per slab
  const i64 qidx = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  f32x4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (qidx < totalq) {
    const f32x4* p = reinterpret_cast<const f32x4*>(part) + qidx;
    int k = g;
    for (; k + G < slabs; k += 2 * G) {                           // two loads in flight per trip, fixed order
      const f32x4 v0 = p[(i64)k * totalq], v1 = p[(i64)(k + G) * totalq];
      s0 += v0;
      s1 += v1;
    }
    if (k < slabs) s0 += p[(i64)k * totalq];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && qidx < totalq) {
    f32x4 v = red[threadIdx.x];
#pragma unroll
    for (int j = 1; j < G; ++j) v += red[j * 64 + threadIdx.x];
    const i64 pidx = qidx * 4;
    const int b32 = (int)(pidx & 31), a32 = (int)((pidx >> 5) & 31);
    const i64 r = pidx >> 10;
    const int t = (int)(r % T);
    const int pair = (int)(r / T);
    const int a = (pair / BB32) * 32 + a32, b = (pair % BB32) * 32 + b32;
    if (a < A) {
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if (b + j < B) {
          float* d = dw + a * sa + (b + j) * sb + t;
          *d = accumulate ? *d + v[j] : v[j];
        }
    }
  }
}

// many slabs (the 32- and 64-channel levels): 16 waves per 64 output quads, else 4
static void seg3d_launch_wgrad_reduce(const float* workspace, float* dw, int slabs, int A, int B, int BB32, int npairs, i64 sa,
                                      i64 sb, int accumulate, hipStream_t s) {
  const i64 totalq = (i64)npairs * 27 * 256;
  const unsigned grid = (unsigned)((totalq + 63) / 64);
  if (slabs >= 32)
    hipLaunchKernelGGL(conv3d_k3_wgrad_reduce_kernel<16>, dim3(grid), dim3(1024), 0, s, workspace, dw, slabs, A, B, BB32, npairs, sa,
                       sb, accumulate);
  else
    hipLaunchKernelGGL(conv3d_k3_wgrad_reduce_kernel<4>, dim3(grid), dim3(256), 0, s, workspace, dw, slabs, A, B, BB32, npairs, sa,
                       sb, accumulate);
}

static int seg3d_wgrad_slabs(int N, int D, int H, int W, int npairs) {
  const int ntiles = N * seg3d_cdiv(D, SEG3D_WG_TZ) * seg3d_cdiv(H, SEG3D_WG_TY) * seg3d_cdiv(W, SEG3D_WG_TX);
  int slabs = 512 / npairs;                    // 2 resident workgroups per CU over the whole grid
  if (slabs > (ntiles + 3) / 4) slabs = (ntiles + 3) / 4;  // small levels: >= 4 tiles per workgroup, fewer partial slabs
  if (slabs < 1) slabs = 1;
  return slabs;
}

struct Seg3dWgradPlan {
  int version;  // 2: one persistent workgroup per CU, LDS-DMA (the only fp32 form; kept for the variant codes)
  int nb;       // 32-channel blocks of dy per workgroup (version 2)
  int slabs;
  int tz, ty, tx;  // spatial tile (version 2; version 1 is 4 x 4 x 8)
};

static Seg3dWgradPlan seg3d_wgrad_plan(int N, int D, int H, int W, int Cin, int Cout) {
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  Seg3dWgradPlan p;
  p.version = 2;
  p.nb = 1;   // NB = 2 (two dy blocks per workgroup) halves the x staging, but hipcc spills the unrolled loop's piece
              // table and reloads it behind vmcnt(0) waits (negative results): not used until that is solved
  p.tz = SEG3D_WG_TZ, p.ty = SEG3D_WG_TY, p.tx = SEG3D_WG_TX;
  // a tile shape that divides the level: no half-empty tiles, every tile on the fast DMA path
  if (D % 4 == 0 && H % 4 == 0 && W % 8 == 0) {
  } else if (D % 4 == 0 && H % 4 == 0 && W % 4 == 0) {
    p.tz = 4, p.ty = 4, p.tx = 4;
  } else if (D % 2 == 0 && H % 6 == 0 && W % 6 == 0) {
    p.tz = 2, p.ty = 6, p.tx = 6;
  }
  const i64 ntiles = (i64)N * seg3d_cdiv(D, p.tz) * seg3d_cdiv(H, p.ty) * seg3d_cdiv(W, p.tx);
  const int npg = CIB32 * (COB32 / p.nb);
  i64 slabs = 256 / npg;  // one resident workgroup per CU over the whole grid
  if (slabs > (ntiles + 1) / 2) slabs = (ntiles + 1) / 2;  // small levels: >= 2 tiles per workgroup
  if (slabs < 1) slabs = 1;
  p.slabs = (int)slabs;
  return p;
}

extern "C" long long seg3d_conv3d_k3_mfma_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  return (long long)seg3d_wgrad_plan(N, D, H, W, Cin, Cout).slabs * npairs * 27 * 1024;
}

template <int NB, int TZ, int TY, int TX>
static int launch_wgrad2(const float* x, const float* dy, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                         int slabs, hipStream_t s) {
  const int ntz = seg3d_cdiv(D, TZ), nty = seg3d_cdiv(H, TY), ntx = seg3d_cdiv(W, TX);
  const int ntiles = N * ntz * nty * ntx;
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wgrad2_kernel<NB, TZ, TY, TX>), configured, "conv3d_k3_wgrad2")) return rc;
  const int CIB32 = (Cin + 31) / 32, COG = (Cout + 31) / 32 / NB;
  const size_t lds = (size_t)2 * ((TZ + 2) * (TY + 2) * (TX + 2) * 32 + NB * TZ * TY * TX * 32) * 4;
  hipLaunchKernelGGL((conv3d_k3_wgrad2_kernel<NB, TZ, TY, TX>), dim3((unsigned)(slabs * CIB32 * COG)), dim3(256), lds, s, x, dy, workspace,
                     N, D, H, W, Cin, Cout, ntz, nty, ntx, ntiles, slabs, COG);
  return SEG3D_OK;
}

// dw is written in the reference Conv3d layout [Cout][Cin][3][3][3]  (sa = 27 for ci, sb = Cin*27 for co).
extern "C" int seg3d_conv3d_k3_mfma_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D,
                                          int H, int W, int Cin, int Cout, int accumulate, void* stream) {
  SEG3D_REQUIRE(x && dy && dw && workspace, "seg3d_conv3d_k3_mfma_wgrad: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k3_mfma_wgrad: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0,
                "seg3d_conv3d_k3_mfma_wgrad: Cin and Cout must be multiples of 4 (got %d, %d)", Cin, Cout);
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  const int npairs = CIB32 * COB32;
  SEG3D_REQUIRE((i64)N * D * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31),
                "seg3d_conv3d_k3_mfma_wgrad: tensor exceeds 2^31 elements");
  const Seg3dWgradPlan plan = seg3d_wgrad_plan(N, D, H, W, Cin, Cout);
  // (tile decoding by float reciprocal is exact below 2^22 tiles only)
  SEG3D_REQUIRE((i64)N * seg3d_cdiv(D, plan.tz) * seg3d_cdiv(H, plan.ty) * seg3d_cdiv(W, plan.tx) < SEG3D_FDIV_MAX,
                "seg3d_conv3d_k3_mfma_wgrad: more than 2^22 tiles");
  const int slabs = plan.slabs;
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (plan.tx == 4) rc = launch_wgrad2<1, 4, 4, 4>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s);
  else if (plan.tx == 6) rc = launch_wgrad2<1, 2, 6, 6>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s);
  else rc = launch_wgrad2<1, 4, 4, 8>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s);
  if (rc != SEG3D_OK) return rc;
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_wgrad");
  seg3d_launch_wgrad_reduce(workspace, dw, slabs, Cin, Cout, COB32, npairs, (i64)27, (i64)Cin * 27, accumulate, s);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_mfma_wgrad(reduce)");
  return SEG3D_OK;
}

// bf16 mode: x and dy bf16.  Levels whose extents are multiples of a tile (4x4x8, else 4x4x4) run the bf16-MFMA kernel
// conv3d_k3_wgrad3_bf16_kernel; other shapes the first-generation kernel with the operands widened while staging.
struct Seg3dWgrad16Plan {
  int version;  // 3: bf16 MFMA, transposing LDS reads; 1: register-staged, fp32 MFMA
  int tx;       // 8 or 4 (version 3)
  int slabs;
};

static Seg3dWgrad16Plan seg3d_wgrad16_plan(int N, int D, int H, int W, int Cin, int Cout) {
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  const int npairs = CIB32 * COB32;
  Seg3dWgrad16Plan p;
  p.version = 1;
  p.tx = 8;
  p.slabs = seg3d_wgrad_slabs(N, D, H, W, npairs);
  if ((Cin & 7) || (Cout & 7)) return p;
  // tile 4x4x8 unless 4x4x4 wastes fewer voxels (levels that are multiples of 4 but not of 8, e.g. 12^3)
  const i64 vol8 = (i64)seg3d_cdiv(D, 4) * seg3d_cdiv(H, 4) * seg3d_cdiv(W, 8) * 128;
  const i64 vol4 = (i64)seg3d_cdiv(D, 4) * seg3d_cdiv(H, 4) * seg3d_cdiv(W, 4) * 64;
  const int tx = (vol4 < vol8) ? 4 : 8;
  const i64 ntiles = (i64)N * seg3d_cdiv(D, 4) * seg3d_cdiv(H, 4) * seg3d_cdiv(W, tx);
  if (ntiles >= SEG3D_FDIV_MAX) return p;
  p.version = 3;
  p.tx = tx;
  int slabs = 256 / npairs;   // one resident workgroup per CU over the whole grid
  if (slabs > (ntiles + 1) / 2) slabs = (int)((ntiles + 1) / 2);
  if (slabs < 1) slabs = 1;
  p.slabs = slabs;
  return p;
}

extern "C" long long seg3d_conv3d_k3_bf16_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  const int npairs = ((Cin + 31) / 32) * ((Cout + 31) / 32);
  return (long long)seg3d_wgrad16_plan(N, D, H, W, Cin, Cout).slabs * npairs * 27 * 1024;
}

template <int TX, bool IRR>
static int launch_wgrad3(const void* x, const void* dy, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                         int slabs, hipStream_t s) {
  constexpr int TZ = 4, TY = 4;
  const int ntz = seg3d_cdiv(D, TZ), nty = seg3d_cdiv(H, TY), ntx = seg3d_cdiv(W, TX);
  const int ntiles = N * ntz * nty * ntx;
  static Seg3dOncePerDevice configured;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wgrad3_bf16_kernel<TZ, TY, TX, IRR>), configured, "conv3d_k3_wgrad3_bf16")) return rc;
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  constexpr int NVH = (TZ + 2) * (TY + 2) * (TX + 2), MTV = TZ * TY * TX;
  const size_t lds = (size_t)SEG3D_WG3_NBUF * ((((NVH + 15) / 16) + MTV / 16 + 3) / 4) * 4 * 1024;
  hipLaunchKernelGGL((conv3d_k3_wgrad3_bf16_kernel<TZ, TY, TX, IRR>), dim3((unsigned)(slabs * CIB32 * COB32)), dim3(256), lds, s,
                     reinterpret_cast<const seg3d_bf16*>(x), reinterpret_cast<const seg3d_bf16*>(dy), workspace, N, D, H, W,
                     Cin, Cout, ntz, nty, ntx, ntiles, slabs, COB32);
  return SEG3D_OK;
}

extern "C" int seg3d_conv3d_k3_bf16_wgrad(const void* x, const void* dy, float* dw, float* workspace, int N, int D, int H,
                                          int W, int Cin, int Cout, int accumulate, void* stream) {
  SEG3D_REQUIRE(x && dy && dw && workspace, "seg3d_conv3d_k3_bf16_wgrad: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k3_bf16_wgrad: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0,
                "seg3d_conv3d_k3_bf16_wgrad: Cin and Cout must be multiples of 4 (got %d, %d)", Cin, Cout);
  SEG3D_REQUIRE((i64)N * D * H * W * (Cin > Cout ? Cin : Cout) < (1ll << 31),
                "seg3d_conv3d_k3_bf16_wgrad: tensor exceeds 2^31 elements");
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32;
  const int npairs = CIB32 * COB32;
  const Seg3dWgrad16Plan plan = seg3d_wgrad16_plan(N, D, H, W, Cin, Cout);
  const int slabs = plan.slabs;
  hipStream_t s = (hipStream_t)stream;
  if (plan.version == 3) {
    const bool irr = D % 4 || H % 4 || W % plan.tx;
    int rc;
    if (irr)
      rc = plan.tx == 8 ? launch_wgrad3<8, true>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s)
                        : launch_wgrad3<4, true>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s);
    else
      rc = plan.tx == 8 ? launch_wgrad3<8, false>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s)
                        : launch_wgrad3<4, false>(x, dy, workspace, N, D, H, W, Cin, Cout, slabs, s);
    if (rc != SEG3D_OK) return rc;
  } else {
    const int ntz = seg3d_cdiv(D, SEG3D_WG_TZ), nty = seg3d_cdiv(H, SEG3D_WG_TY), ntx = seg3d_cdiv(W, SEG3D_WG_TX);
    const int ntiles = N * ntz * nty * ntx;
    hipLaunchKernelGGL(conv3d_k3_wgrad_mfma_bf16_kernel, dim3(slabs, npairs), dim3(256), 0, s, x, dy, workspace, N, D, H, W,
                       Cin, Cout, ntz, nty, ntx, ntiles, COB32);
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_bf16_wgrad");
  seg3d_launch_wgrad_reduce(workspace, dw, slabs, Cin, Cout, COB32, npairs, (i64)27, (i64)Cin * 27, accumulate, s);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_bf16_wgrad(reduce)");
  return SEG3D_OK;
}
