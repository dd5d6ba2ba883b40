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
#include <type_traits>

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
// Input / arithmetic mode of the gather and scatter kernels:
//   0  fp32 input, fp32 weight image, four v_mfma_f32_32x32x2_f32 per tap and 8-channel chunk
//   1  bf16 input widened to fp32 when the staged chunk is written to LDS; weights and MFMAs as in mode 0
//   2  bf16 input AND bf16 weight image (seg3d_pack_weights_mfma_bf16, T = 8): the staged 16 bytes per (voxel, half) are
//      8 bf16 channels, a chunk is 16 channels, one v_mfma_f32_32x32x16_bf16 per tap -- same LDS bytes, 1/8 of the MFMA
//      cycles (the fp32 forms of these layers were MFMA-bound: N = 16..32 output channels fill half an MFMA tile)
typedef __bf16 k2_bf16x8 __attribute__((ext_vector_type(8)));
// input rows through a buffer resource of the SAMPLE (k2_make_rsrc): voff = byte offset of the entry inside the sample, or K2_OOB --
// an offset past the resource's range returns zeros, so neither a zero source nor a select at the LDS store is needed, and the
// chunk's channel offset rides in the instruction's scalar offset: no vector instruction per load
typedef unsigned k2_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned k2_u32x2 __attribute__((ext_vector_type(2)));
#define K2_OOB 0x80000000u
__device__ __forceinline__ __amdgpu_buffer_rsrc_t k2_make_rsrc(const void* base_wave_uniform, unsigned num_bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base_wave_uniform), (short)0, (int)num_bytes, 0x00020000);
}
template <int MODE> struct K2In;
template <> struct K2In<0> {
  typedef f32x4 raw;
  static constexpr int CPH = 4;   // channels per (voxel, half) entry
  static constexpr int ESZ = 4;   // bytes per element
  static __device__ __forceinline__ raw bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  }
  static __device__ __forceinline__ raw load(const void* p, i64 e) { return Seg3dQuad<false>::load(p, e); }
  static __device__ __forceinline__ f32x4 cvt(raw r) { return r; }
};
template <> struct K2In<1> {
  typedef uint2 raw;
  static constexpr int CPH = 4;
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ raw bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    const k2_u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(r, (int)voff, (int)soff, 0);
    return uint2{v[0], v[1]};
  }
  static __device__ __forceinline__ raw load(const void* p, i64 e) { return Seg3dQuad<true>::load(p, e); }
  static __device__ __forceinline__ f32x4 cvt(raw r) { return Seg3dQuad<true>::cvt(r); }
};
template <> struct K2In<2> {
  typedef f32x4 raw;              // 16 raw bytes = 8 bf16 channels
  static constexpr int CPH = 8;
  static constexpr int ESZ = 2;
  static __device__ __forceinline__ raw bload(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
    return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r, (int)voff, (int)soff, 0));
  }
  static __device__ __forceinline__ raw load(const void* p, i64 e) {
    return *reinterpret_cast<const f32x4*>(reinterpret_cast<const seg3d_bf16*>(p) + e);
  }
  static __device__ __forceinline__ f32x4 cvt(raw r) { return r; }
};
template <int MODE>
__device__ __forceinline__ f32x16 k2_mfma_step(f32x4 bw, f32x4 av, f32x16 acc) {
  if constexpr (MODE == 2) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(k2_bf16x8, bw), __builtin_bit_cast(k2_bf16x8, av), acc,
                                                   0, 0, 0);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[r], av[r], acc, 0, 0, 0);
    return acc;
  }
}

template <int MODE, bool OUT_BF = false>
__global__ __launch_bounds__(256, 2) void conv3d_k2s2_mfma_kernel(const void* __restrict__ x,
                                                                    const float* __restrict__ wp,
                                                                    const float* __restrict__ bias, float* __restrict__ y,
                                                                    float* __restrict__ stats, int N, int Do, int Ho,
                                                                    int Wo, int Cin, int Cout, int TZ, int TY, int TX,
                                                                    int ntz, int nty, int ntx, int ldx) {
  // ldx: elements between consecutive voxel rows of x (= Cin, or the width of the concatenated buffer x is a channel slice of)
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
  constexpr int CPH = K2In<MODE>::CPH, CPC = 2 * CPH;   // channels per half / per chunk
  const int CIB = (Cin + CPC - 1) / CPC;
  const int cob = blockIdx.y;
  // A workgroup owns ONE tile, so this prologue is paid per tile -- and in fp32 every vector instruction of it is paid on the
  // pipe the MFMAs use (round 3: 10.6 vector instructions per MFMA at the top level, 60 % of them here).  The tile edges are
  // powers of two (k2_pick_tile): lane -> voxel decoding is shifts and masks, not reciprocal multiplies.
  const int lgHX = __builtin_ctz(HX), lgHY = __builtin_ctz(HY), lgTX = __builtin_ctz(TX), lgTY = __builtin_ctz(TY);
  int b = blockIdx.x;
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * TX;

  // staged entries of this thread: byte offset inside the sample (K2_OOB: outside the volume / past the tile -> zeros)
  constexpr int ESZ = K2In<MODE>::ESZ;
  unsigned goff[K2_MAXE];
  const int hh = tid & 1;
#pragma unroll
  for (int e = 0; e < K2_MAXE; ++e) {
    const int eidx = tid + e * 256;
    goff[e] = K2_OOB;
    if (eidx < 2 * NV) {
      const int v = eidx >> 1;
      const int hx = v & (HX - 1), t = v >> lgHX;
      const int hy = t & (HY - 1), hz = t >> lgHY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      if (gz < Di && gy < Hi && gx < Wi) goff[e] = (unsigned)((((gz * Hi + gy) * Wi + gx) * ldx + hh * CPH) * ESZ);
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = k2_make_rsrc(reinterpret_cast<const char*>(x) + (i64)n * Di * Hi * Wi * ldx * ESZ,
                                                  (unsigned)Di * Hi * Wi * ldx * ESZ);
  for (int idx = tid; idx < MT; idx += 256) {
    const int tx = idx & (TX - 1), t = idx >> lgTX;
    const int ty = t & (TY - 1), tz = t >> lgTY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    voff[idx] = (gz < Do && gy < Ho && gx < Wo) ? ((n * Do + gz) * Ho + gy) * Wo + gx : -1;
  }
  int abase;
  {
    const int idx = wave * 32 + li;
    int vb = 0;
    if (idx < MT) {
      const int tx = idx & (TX - 1), t = idx >> lgTX;
      const int ty = t & (TY - 1), tz = t >> lgTY;
      vb = ((2 * tz) * HY + 2 * ty) * HX + 2 * tx;
    }
    abase = (lh * NV + vb) * 4;
  }
  const int bbase = (lh * 32 + li) * 4;
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  // Register-prefetch pipeline: all loads of chunk c+1 are issued back to back (branch-free, clamped addresses; the
  // zero-select happens at the LDS store) before the MFMAs of chunk c, so a workgroup keeps 32 KB in flight -- at the
  // top level this kernel is HBM-bound and a load consumed right where it is issued serialises on memory latency.
  // TWO chunks in flight (round 4): a tile of the top level has only two chunks, and with one in flight their memory round trips
  // were serialised (load 0 -> wait -> store -> issue load 1 -> MFMAs 0 -> wait ...); both register sets are requested up front
  typename K2In<MODE>::raw xst[2][K2_MAXE];
  f32x4 wst[2][2];
  auto load_chunk = [&](int cib, int set) {
    // (a half past the last channel: K2_OOB ored in -- only the last chunk of a channel count that is not a multiple of CPC)
    const unsigned hk = cib * CPC + hh * CPH < Cin ? 0u : K2_OOB;
#pragma unroll
    for (int e = 0; e < K2_MAXE; ++e) xst[set][e] = K2In<MODE>::bload(xrs, goff[e] | hk, (unsigned)(cib * CPC * ESZ));
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + ((i64)cob * CIB + cib) * K2_W_CHUNK);
#pragma unroll
    for (int k = 0; k < 2; ++k) wst[set][k] = wsrc[tid + k * 256];
  };
  auto store_chunk = [&](int set) {
#pragma unroll
    for (int e = 0; e < K2_MAXE; ++e) {
      const int eidx = tid + e * 256;
      if (eidx < 2 * NV) *reinterpret_cast<f32x4*>(xs + (hh * NV + (eidx >> 1)) * 4) = K2In<MODE>::cvt(xst[set][e]);   // (zeros where out of range)
    }
    f32x4* wdst = reinterpret_cast<f32x4*>(ws);
#pragma unroll
    for (int k = 0; k < 2; ++k) wdst[tid + k * 256] = wst[set][k];
  };
  auto multiply_chunk = [&]() {
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
      const int tapoff = ((kz * HY + ky) * HX + kx) * 4;
      const f32x4 bw = *reinterpret_cast<const f32x4*>(ws + tap * 256 + bbase);
      const f32x4 av = *reinterpret_cast<const f32x4*>(xs + abase + tapoff);
      // A = weights, B = voxels: D[co][voxel] -- a lane owns voxel (lane & 31) and channels 8 g + 4 (lane >> 5) + c
      acc = k2_mfma_step<MODE>(bw, av, acc);
    }
  };
  load_chunk(0, 0);
  if (CIB > 1) load_chunk(1, 1);
  for (int cib = 0; cib < CIB; cib += 2) {
    __syncthreads();  // previous chunk fully consumed (also publishes voff on the first pass)
    store_chunk(0);
    __syncthreads();
    if (cib + 2 < CIB) load_chunk(cib + 2, 0);
    multiply_chunk();
    if (cib + 1 < CIB) {
      __syncthreads();
      store_chunk(1);
      __syncthreads();
      if (cib + 3 < CIB) load_chunk(cib + 3, 1);
      multiply_chunk();
    }
  }

  // epilogue: 4 dwordx4 stores per lane (the lane's voxel, 4 x 4 consecutive channels) instead of 16 dword stores --
  // at the top level this kernel is HBM-bound and the store INSTRUCTION rate was a third of its time
  float s[2] = {0.f, 0.f};
  {
    const int idx = wave * 32 + li;
    const int vo = idx < MT ? voff[idx] : -1;
    const int co0 = cob * 32 + 4 * lh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int co = co0 + 8 * g4;
      if (vo >= 0 && co < Cout) {   // Cout % 4 == 0 (host-checked)
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + co);
        f32x4 v;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          v[c] = acc[4 * g4 + c] + bv[c];
          s[0] += v[c];
          s[1] += v[c] * v[c];
        }
        Seg3dQuad<OUT_BF>::store(y, (i64)vo * Cout + co, v);
      }
    }
  }
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, xs);
    if (tid < 4) {   // slots [tile][4][column block] (the direct kernel's waves write one each): the tile's sums in the first
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + ((((i64)n * tiles_per_sample + tile) * 4 + tid) * gridDim.y + cob) * 2;
      dst[0] = tid == 0 ? s[0] : 0.f;
      dst[1] = tid == 0 ? s[1] : 0.f;
    }
  }
}

#ifdef K2_STAMPS   // diagnostic build only (tools/ubench/k2_stamp.hip): s_memtime stamps of every wave of the direct gather kernel
__device__ long long* k2_stamp_buf;
#define K2_STAMP(k)                                                                                                          \
  do {                                                                                                                       \
    if ((threadIdx.x & 63) == 0)                                                                                             \
      k2_stamp_buf[(((long long)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define K2_STAMP(k)
#endif
// The gather kernel WITHOUT LDS (round 4): a lane of the 32x32 MFMA needs, per tap and 8-channel chunk, exactly the four
// channels 4 lh .. 4 lh + 3 of ITS voxel's input voxel and the four weights (k = 4 lh + r, co = li) of the packed image -- one
// 16-byte load each, the second perfectly coalesced (the T = 8 image is [tap][half][co][4]).  So nothing is staged and nothing
// is shared: no barrier in the K loop, every wave runs on its own with the operands of TWO chunks in flight (a tap's registers
// are refilled right behind its MFMAs), and the other waves of the SIMD cover the latency.  The staged kernel above spent its
// time between barriers (wave cycles waiting 0.54-0.69, matrix pipe busy 0.13-0.35: profiles/r04_c_pmc_sq_summary.txt);
// its tiles, grid, statistics slots and epilogue are kept.  The weights of a column block are re-read by every wave that
// works on it (8 KB per chunk from L1 / L2: 64 B/clk/CU against 16 KB per 2 048 MFMA cycles).
// NCOB = 2: a wave multiplies its voxels' operands into TWO column blocks (the input is read once where Cout >= 64: as separate
//   workgroups the column blocks of a tile are dispatched a whole grid apart and the second read came from HBM again).
// KSPLIT = 4 (the lower levels: a few hundred 32 x 32 wave tiles with K = 8 x 64 .. 8 x 256): the four waves of a workgroup share
//   ONE 32-voxel sub-tile and take one (kz, ky) run each -- a quarter of K -- then add their accumulators through LDS in wave
//   order 0..3 (fixed: run to run bitwise the same) and store one channel quad each.
// Statistics slots: [sample][tile][4][column block] -- a wave's own sums (no barrier), or the sub-tile's with KSPLIT.
template <int MODE, bool OUT_BF = false, int NCOB = 1, int KSPLIT = 1>
__global__ __launch_bounds__(256, 2) void conv3d_k2s2_direct_kernel(const void* __restrict__ x, const float* __restrict__ wp,
                                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                                      float* __restrict__ stats, int N, int Do, int Ho, int Wo,
                                                                      int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
                                                                      int ntx, int ldx) {
  static_assert((KSPLIT == 1 || KSPLIT == 4) && (NCOB == 1 || NCOB == 2) && (KSPLIT == 1 || NCOB == 1), "variants");
  constexpr int NSET = 2;   // steps in flight (a ring of register sets; four measured no faster, also alone on a SIMD)
  __shared__ __attribute__((aligned(16))) float red[KSPLIT == 4 ? 4 * 16 * 64 : 8];
  __shared__ __attribute__((aligned(16))) float tbuf[OUT_BF ? 4 : (KSPLIT == 4 ? 32 * 36 : 4 * 32 * 36)];   // epilogue transpose: [voxel 32][36] per wave
  __shared__ int vos[OUT_BF ? 1 : (KSPLIT == 4 ? 32 : 128)];
  K2_STAMP(0);
  const int MT = TZ * TY * TX;
  const int Di = 2 * Do, Hi = 2 * Ho, Wi = 2 * Wo;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int CPH = K2In<MODE>::CPH, CPC = 2 * CPH, ESZ = K2In<MODE>::ESZ;
  const int CIB = (Cin + CPC - 1) / CPC;
  const int cob0 = blockIdx.y * NCOB, ncob = gridDim.y * NCOB;
  const int lgTX = __builtin_ctz(TX), lgTY = __builtin_ctz(TY);
  int b = KSPLIT == 4 ? blockIdx.x >> 2 : blockIdx.x;
  const int sub = KSPLIT == 4 ? (int)(blockIdx.x & 3) : wave;   // 32-voxel quarter of the tile this wave works on
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  // this lane's output voxel and the byte offset of its 2^3 input cell inside the sample (K2_OOB: past the tile / the volume)
  int vo = -1;
  unsigned xoff = K2_OOB;
  {
    const int idx = sub * 32 + li;
    const int tx = idx & (TX - 1), t = idx >> lgTX;
    const int ty = t & (TY - 1), tz = t >> lgTY;
    const int gz = tiz * TZ + tz, gy = tiy * TY + ty, gx = tix * TX + tx;
    if (idx < MT && gz < Do && gy < Ho && gx < Wo) {
      vo = ((n * Do + gz) * Ho + gy) * Wo + gx;
      xoff = (unsigned)((((2 * gz * Hi + 2 * gy) * Wi + 2 * gx) * ldx + lh * CPH) * ESZ);
    }
  }
  const __amdgpu_buffer_rsrc_t xrs = k2_make_rsrc(reinterpret_cast<const char*>(x) + (i64)n * Di * Hi * Wi * ldx * ESZ,
                                                  (unsigned)Di * Hi * Wi * ldx * ESZ);
  // K order: the (kx, channel) run of a (kz, ky) pair is 2 Cin contiguous floats of the input (ldx == Cin), and a STEP is four
  // consecutive 8-channel chunks of it -- 128 contiguous bytes per lane's voxel, one cache line, requested by four back-to-back
  // instructions.  (Chunk-major order -- all eight taps of chunk 0, then chunk 1, ... -- used 32 bytes of a line per visit and came
  // back for the rest a whole chunk later, when the line had left the 32-KB L1: every chunk re-fetched whole lines from L2 and the
  // launch ran at the L1 fill rate, 16 useful B/clk/CU: 23 600 of a wave's 59 700 cycles passed before its first MFMA.)
  const int SPR = CIB >> 1;                      // steps per (kz, ky) run; host-checked: CIB even (KSPLIT: SPR even)
  const int NS = KSPLIT == 4 ? SPR : 4 * SPR;    // steps of this wave
  const f32x4* wbase = reinterpret_cast<const f32x4*>(wp + (i64)cob0 * CIB * K2_W_CHUNK) + lane;   // + (cib * 8 + tap) * 64
  f32x16 acc[NCOB];
#pragma unroll
  for (int c = 0; c < NCOB; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  typename K2In<MODE>::raw xv[NSET][4];
  f32x4 wv[NSET][4][NCOB];
  int ld_run = KSPLIT == 4 ? wave : 0, ld_piece = 0;   // (kz, ky) run and 128-byte piece of the step being fetched: scalar counters
  auto load_step = [&](int set, int j) __attribute__((always_inline)) {   // (set, j: constants after unrolling)
    const int q = ld_piece * 4 + j;
    const int kx = q >= CIB ? 1 : 0, cib = q - kx * CIB;
    const int kz = ld_run >> 1, ky = ld_run & 1;
    const unsigned soff = __builtin_amdgcn_readfirstlane((unsigned)((((kz * Hi + ky) * Wi + kx) * ldx + cib * CPC) * ESZ));
    const unsigned hk = cib * CPC + lh * CPH < Cin ? 0u : K2_OOB;   // (a half past the last channel: zeros)
    xv[set][j] = K2In<MODE>::bload(xrs, xoff | hk, soff);
    const int wi = __builtin_amdgcn_readfirstlane((cib * 8 + ld_run * 2 + kx) * 64);
#pragma unroll
    for (int c = 0; c < NCOB; ++c) wv[set][j][c] = wbase[wi + c * CIB * (K2_W_CHUNK / 4)];
  };
  auto load_advance = [&]() __attribute__((always_inline)) {
    if (++ld_piece == SPR) ld_piece = 0, ++ld_run;
  };
  auto multiply = [&](int set, int j) __attribute__((always_inline)) {
    const f32x4 xb = K2In<MODE>::cvt(xv[set][j]);
#pragma unroll
    for (int c = 0; c < NCOB; ++c) acc[c] = k2_mfma_step<MODE>(wv[set][j][c], xb, acc[c]);
  };
  // the refills are UNCONDITIONAL inside their blocks (a load behind a branch makes the compiler's s_waitcnt bookkeeping assume
  // the worst at the join: a first version waited for vmcnt(0), i.e. for the refills it had just issued), the tail is peeled
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
#pragma unroll
    for (int j = 0; j < 4; ++j) load_step(set, j);
    load_advance();
  }
  K2_STAMP(1);
  for (int st = 0; st + 2 * NSET <= NS; st += NSET) {   // steps st .. st + NSET - 1 multiplied, the NSET behind them requested
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        multiply(set, j);
        load_step(set, j);
      }
      load_advance();
    }
  }
#pragma unroll
  for (int set = 0; set < NSET; ++set)
#pragma unroll
    for (int j = 0; j < 4; ++j) multiply(set, j);
  K2_STAMP(2);

  if constexpr (KSPLIT == 4) {   // acc[0] of the four waves -> the sum's registers 4 wave .. 4 wave + 3 (channel quad g4 = wave)
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[0][r];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int r = 4 * wave + c;
      acc[0][c] = ((red[(0 * 16 + r) * 64 + lane] + red[(1 * 16 + r) * 64 + lane]) + red[(2 * 16 + r) * 64 + lane]) + red[(3 * 16 + r) * 64 + lane];
    }
    __syncthreads();   // (red is reused by the statistics)
  }
  K2_STAMP(3);
  const int tile = (tiz * nty + tiy) * ntx + tix, tiles_per_sample = ntz * nty * ntx;
  // fp32 outputs go through an LDS transpose: a lane owns 16-byte pieces of ITS voxel's row, so a store instruction wrote 64
  // pieces 128 .. 512 bytes apart (tools/ubench/stride_load.hip: 3.6 TB/s at best, 1.0 TB/s at 256 bytes -- the 64-channel rows of
  // the top level's data-gradient); transposed, eight consecutive lanes write the 128 contiguous bytes of a voxel's column block
  constexpr int ROW = 36;                          // floats per voxel row in LDS: 32 + 4 (b128 writes of 16 lanes: conflict-free)
  float* tr = KSPLIT == 4 ? tbuf : tbuf + wave * (32 * ROW);
  if (!OUT_BF && lh == 0) vos[(KSPLIT == 4 ? 0 : wave * 32) + li] = vo;
#pragma unroll
  for (int c = 0; c < NCOB; ++c) {
    float s[2] = {0.f, 0.f};
    const int co0 = (cob0 + c) * 32 + 4 * lh;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      if (KSPLIT == 4 && g4 > 0) break;
      const int gq = KSPLIT == 4 ? wave : g4;      // channel quad 8 gq + 4 lh of the column block
      const int co = co0 + 8 * gq;
      if (vo >= 0 && co < Cout) {   // Cout % 4 == 0 (host-checked)
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + co);
        f32x4 v;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[c][4 * g4 + e] + bv[e];
          s[0] += v[e];
          s[1] += v[e] * v[e];
        }
        if (OUT_BF) Seg3dQuad<OUT_BF>::store(y, (i64)vo * Cout + co, v);
        else *reinterpret_cast<f32x4*>(tr + li * ROW + 8 * gq + 4 * lh) = v;
      }
    }
    if constexpr (!OUT_BF) {
      if (KSPLIT == 4) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
      const int t = KSPLIT == 4 ? tid : lane;      // KSPLIT: the workgroup's 256 threads store the sub-tile's 32 rows at once
#pragma unroll
      for (int it = 0; it < (KSPLIT == 4 ? 1 : 4); ++it) {
        const int v = it * 8 + (t >> 3), piece = t & 7;
        const int vo_v = vos[(KSPLIT == 4 ? 0 : wave * 32) + v];
        const int co = (cob0 + c) * 32 + 4 * piece;
        if (vo_v >= 0 && co < Cout) *reinterpret_cast<f32x4*>(y + (i64)vo_v * Cout + co) = *reinterpret_cast<const f32x4*>(tr + v * ROW + 4 * piece);
      }
      if (NCOB > 1 || KSPLIT == 4) { if (KSPLIT == 4) __syncthreads(); else __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
    }
    if (stats) {
      float* dst = stats + ((((i64)n * tiles_per_sample + tile) * 4 + sub) * ncob + cob0 + c) * 2;
      if constexpr (KSPLIT == 4) {
        block_sum_256<2>(s, red);
        if (tid == 0) dst[0] = s[0], dst[1] = s[1];
      } else {
        s[0] = wave_sum(s[0]), s[1] = wave_sum(s[1]);
        if (lane == 0) dst[0] = s[0], dst[1] = s[1];
      }
    }
  }
  K2_STAMP(4);
}

extern "C" long long seg3d_conv3d_k2s2_mfma_stats_count(int Do, int Ho, int Wo, int Cout) {
  K2Tile t = k2_pick_tile(Do, Ho, Wo);
  return (long long)seg3d_cdiv(Do, t.tz) * seg3d_cdiv(Ho, t.ty) * seg3d_cdiv(Wo, t.tx) * 4 * ((Cout + 31) / 32);   // [tile][4][column block]
}

// x [N][2Do][2Ho][2Wo][Cin] -> y [N][Do][Ho][Wo][Cout];  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 8)
static int k2_gather_launch(const void* x, int x_bf16, const float* wp, const float* bias, float* y, float* stats, int N,
                            int Do, int Ho, int Wo, int Cin, int Cout, void* stream, int out_bf16 = 0, int ldx = 0) {
  if (ldx == 0) ldx = Cin;
  SEG3D_REQUIRE(ldx >= Cin && (ldx % 4) == 0, "seg3d_conv3d_k2s2_mfma_fwd: ld_x must be a multiple of 4 >= Cin");
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k2s2_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && Do > 0 && Ho > 0 && Wo > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k2s2_mfma_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0,
                "seg3d_conv3d_k2s2_mfma_fwd: Cin and Cout must be multiples of 4 (got %d, %d)", Cin, Cout);
  SEG3D_REQUIRE((i64)N * Do * Ho * Wo * 8 * ldx < (1ll << 31) && (i64)N * Do * Ho * Wo * Cout < (1ll << 31),
                "seg3d_conv3d_k2s2_mfma_fwd: tensor exceeds 2^31 elements");
  SEG3D_REQUIRE((i64)Do * Ho * Wo * 8 * ldx * 4 < (1ll << 31), "seg3d_conv3d_k2s2_mfma_fwd: one sample of x exceeds 2^31 bytes");
  K2Tile t = k2_pick_tile(Do, Ho, Wo);
  const int ntz = seg3d_cdiv(Do, t.tz), nty = seg3d_cdiv(Ho, t.ty), ntx = seg3d_cdiv(Wo, t.tx);
  const int mt = t.tz * t.ty * t.tx;
  const size_t lds = (size_t)(8 * 8 * mt + K2_W_CHUNK + ((mt + 3) & ~3)) * 4;
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "2x2x2 stride-2 MFMA kernels: more than 2^22 tiles");
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  // x_bf16: 0 = fp32 input, 1 = bf16 input widened while staging (fp32 weight image), 2 = bf16 input + bf16 weight image
  SEG3D_REQUIRE(x_bf16 != 2 || (Cin % 16) == 0, "seg3d_conv3d_k2s2_bf16_fwd: the bf16 weight image needs Cin %% 16 == 0");
#ifndef K2_GATHER_DIRECT
#define K2_GATHER_DIRECT 1   // 0: the LDS-staged kernel (kept for same-box A/B builds)
#endif
  // the direct kernel's variants: KSPLIT where 128-voxel tiles give fewer than two waves per SIMD and K is deep enough to deal
  // out ((kz, ky) runs of at least two steps: Cin >= 32 in fp32), two column blocks per wave where that still leaves three
  const i64 waves1 = (i64)grid.x * grid.y * 4;
#define K2_GATHER(MODE_, OB_)                                                                                        \
  do {                                                                                                               \
    const int cib_ = (Cin + (MODE_ == 2 ? 15 : 7)) / (MODE_ == 2 ? 16 : 8);   /* 2 cib_ steps of four chunks */      \
    if (K2_GATHER_DIRECT && cib_ % 4 == 0 && waves1 < 3072)                                                          \
      hipLaunchKernelGGL((conv3d_k2s2_direct_kernel<MODE_, OB_, 1, 4>), dim3(grid.x * 4, grid.y), dim3(256), 0,      \
                         (hipStream_t)stream, x, wp, bias, y, stats, N, Do, Ho, Wo, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ldx); \
    else if (K2_GATHER_DIRECT && cib_ % 2 == 0 && Cout % 64 == 0 && waves1 >= 16384)                                \
      hipLaunchKernelGGL((conv3d_k2s2_direct_kernel<MODE_, OB_, 2, 1>), dim3(grid.x, grid.y / 2), dim3(256), 0,      \
                         (hipStream_t)stream, x, wp, bias, y, stats, N, Do, Ho, Wo, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ldx); \
    else if (K2_GATHER_DIRECT && cib_ % 2 == 0)                                                                      \
      hipLaunchKernelGGL((conv3d_k2s2_direct_kernel<MODE_, OB_, 1, 1>), grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Do, Ho, Wo, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ldx);                     \
    else                                                                                                             \
      hipLaunchKernelGGL((conv3d_k2s2_mfma_kernel<MODE_, OB_>), grid, dim3(256), lds, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Do, Ho, Wo, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ldx);                     \
  } while (0)
  if (x_bf16 == 2 && out_bf16) K2_GATHER(2, true);
  else if (x_bf16 == 2) K2_GATHER(2, false);
  else if (x_bf16 && out_bf16) K2_GATHER(1, true);
  else if (x_bf16) K2_GATHER(1, false);
  else K2_GATHER(0, false);
#undef K2_GATHER
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k2s2_mfma_fwd");
  return SEG3D_OK;
}

extern "C" int seg3d_conv3d_k2s2_mfma_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int N,
                                          int Do, int Ho, int Wo, int Cin, int Cout, void* stream) {
  return k2_gather_launch(x, 0, wp, bias, y, stats, N, Do, Ho, Wo, Cin, Cout, stream);
}

// the same with x a channel slice of a wider NDHWC buffer: ld_x floats between consecutive voxel rows (inference: an
// encoder feature that was normalised straight into its half of a decoder's concatenated skip buffer)
extern "C" int seg3d_conv3d_k2s2_mfma_fwd_ld(const float* x, int ld_x, const float* wp, const float* bias, float* y,
                                             float* stats, int N, int Do, int Ho, int Wo, int Cin, int Cout, void* stream) {
  return k2_gather_launch(x, 0, wp, bias, y, stats, N, Do, Ho, Wo, Cin, Cout, stream, 0, ld_x);
}

// bf16 mode: x is bf16 ([N][2Do][2Ho][2Wo][Cin]); wp = fp32 image (w_bf16 = 0) or seg3d_pack_weights_mfma_bf16 image
// (w_bf16 = 1, Cin % 16 == 0: bf16 MFMA); bias, statistics fp32; y fp32 or bf16 (out_bf16)
extern "C" int seg3d_conv3d_k2s2_bf16_fwd(const void* x_bf16, const void* wp, const float* bias, void* y, float* stats,
                                          int N, int Do, int Ho, int Wo, int Cin, int Cout, int out_bf16, int w_bf16,
                                          void* stream) {
  return k2_gather_launch(x_bf16, w_bf16 ? 2 : 1, reinterpret_cast<const float*>(wp), bias, reinterpret_cast<float*>(y),
                          stats, N, Do, Ho, Wo, Cin, Cout, stream, out_bf16);
}

// ---------------------------------------------------------------------------------------------------------------
// scatter: input tile TZ x TY x TX (<= 128 voxels), one accumulator per tap, output cell 2^3 per input voxel
// ---------------------------------------------------------------------------------------------------------------
// ADD (data-gradient of a stride-2 conv whose input has a second consumer -- the skip connection): y = result + addend,
// addend in y's dtype with row stride lda (a channel slice of the concatenated gradient); Cout % 8 == 0
// epilogue of the scatter kernels (operands swapped: D[co][voxel]): per tap 4 dwordx4 stores of the lane's voxel, 32 per lane
// instead of 128 dword stores.  Bias first, stores last, no load in between (stores count in vmcnt on gfx9).
// o = output voxel index of tap (0, 0, 0) of this lane's input voxel, or -1
template <bool OUT_BF, bool ADD, bool PAIR, int NACC>
__device__ __forceinline__ void k2_scatter_epilogue(f32x16 (&acc)[NACC], int o, int cob, int lh, const float* __restrict__ bias,
                                                    float* __restrict__ y, const void* __restrict__ addend, int lda, int Cout,
                                                    int Ho, int Wo, float (&s)[2]) {
  const int co0 = cob * 32 + 4 * lh;
  f32x4 bv[4];
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int co = co0 + 8 * g4;
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    bv[g4] = zero;
    if (bias && co < Cout) bv[g4] = *reinterpret_cast<const f32x4*>(bias + co);
  }
  const int ng = (Cout - cob * 32 + 7) / 8 < 4 ? (Cout - cob * 32 + 7) / 8 : 4;   // uniform; Cout % 8 == 0 fast path
  if (PAIR) {
    // accumulator t holds taps 2 t (rows 0..15) and 2 t + 1 (rows 16..31): quads g4 = 0, 1 are channels 4 lh + 8 g4 of
    // the even tap, g4 = 2, 3 the same channels of the odd tap.  Cout % 8 == 0 and Cout <= 16 (host-checked).
    if (o >= 0) {
      const int nq = Cout >> 3;   // channel quads per lane half: 1 or 2
      // ADD: all addend quads of the lane's 2^3 cell (8 taps x <= 2 quads) are requested BEFORE the first store.  Loaded tap by
      // tap, each load followed the previous tap's stores, and on gfx9 the wait for a load also waits for every store issued
      // before it (one in-order vmcnt): the epilogue ran at the latency of eight store + load round trips (round 2: 192 us
      // for the 32 -> 16 data-gradient + skip addend of the top level, 0.33 of the HBM peak).  64 registers at most, beside 64
      // of accumulators: inside the 256 of two workgroups per CU.
      typename Seg3dQuad<OUT_BF>::raw ar[8][2];
      if (ADD) {
#pragma unroll
        for (int tap = 0; tap < 8; ++tap) {
          const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
          const i64 src = ((i64)o + (kz * Ho + ky) * Wo + kx) * lda + co0;
#pragma unroll
          for (int g = 0; g < 2; ++g)
            if (g < nq) ar[tap][g] = Seg3dQuad<OUT_BF>::load(addend, src + 8 * g);
        }
      }
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) {
        const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
        const i64 dsto = ((i64)o + (kz * Ho + ky) * Wo + kx) * Cout + co0;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (g < nq) {
            const int g4 = 2 * (tap & 1) + g;
            f32x4 v;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (ADD) av = Seg3dQuad<OUT_BF>::cvt(ar[tap][g]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[PAIR ? (tap >> 1) : 0][4 * g4 + c] + bv[g][c] + av[c];
              s[0] += v[c];
              s[1] += v[c] * v[c];
            }
            Seg3dQuad<OUT_BF>::store(y, dsto + 8 * g, v);
          }
        }
      }
    }
  } else if ((Cout & 7) == 0) {
    if (o >= 0) {
      // ADD: the addend quads of tap t + 1 are requested before tap t is stored (loads and stores share vmcnt)
      typename Seg3dQuad<OUT_BF>::raw ar[2][4];
      auto load_addend = [&](int tap, typename Seg3dQuad<OUT_BF>::raw (&dst)[4]) {
        const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
        const i64 src = ((i64)o + (kz * Ho + ky) * Wo + kx) * lda + co0;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4)
          if (g4 < ng) dst[g4] = Seg3dQuad<OUT_BF>::load(addend, src + 8 * g4);
      };
      if (ADD) load_addend(0, ar[0]);
#pragma unroll
      for (int tap = 0; tap < 8; ++tap) {
        const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
        const i64 dsto = ((i64)o + (kz * Ho + ky) * Wo + kx) * Cout + co0;
        if (ADD && tap + 1 < 8) load_addend(tap + 1, ar[(tap + 1) & 1]);
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          if (g4 < ng) {
            f32x4 v;
            f32x4 av = {0.f, 0.f, 0.f, 0.f};
            if (ADD) av = Seg3dQuad<OUT_BF>::cvt(ar[tap & 1][g4]);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
              v[c] = acc[PAIR ? 0 : tap][4 * g4 + c] + bv[g4][c] + av[c];
              s[0] += v[c];
              s[1] += v[c] * v[c];
            }
            Seg3dQuad<OUT_BF>::store(y, dsto + 8 * g4, v);
          }
        }
      }
    }
  } else {
#pragma unroll
    for (int tap = 0; tap < 8; ++tap) {
      const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const int co = co0 + 8 * g4;
        if (o >= 0 && co < Cout) {
          f32x4 v;
#pragma unroll
          for (int c = 0; c < 4; ++c) {
            v[c] = acc[PAIR ? 0 : tap][4 * g4 + c] + bv[g4][c];
            s[0] += v[c];
            s[1] += v[c] * v[c];
          }
          Seg3dQuad<OUT_BF>::store(y, ((i64)o + (kz * Ho + ky) * Wo + kx) * Cout + co, v);
        }
      }
    }
  }
}

// PAIR (Cout <= 16, e.g. the 64 -> 16 up-convolution of the top level and the 32 -> 16 data-gradient next to it): the 32
// MFMA rows would be half empty -- and with 8 taps x Cin / 2 MFMAs of 64 cycles per 32 voxels these layers are MFMA-bound
// at the top level, not HBM-bound -- so two taps share one MFMA: row = (tap & 1) * 16 + co.  The weight image is
// re-paired while it is staged into LDS; 4 accumulators instead of 8.
template <int MODE, bool OUT_BF = false, bool ADD = false, bool PAIR = false>
__global__ __launch_bounds__(256, 2) void convT3d_k2s2_mfma_kernel(const void* __restrict__ x,
                                                                     const float* __restrict__ wp,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ stats,
                                                                     int N, int Di, int Hi, int Wi, int Cin, int Cout,
                                                                     int TZ, int TY, int TX, int ntz, int nty, int ntx,
                                                                     const void* __restrict__ addend, int lda) {
  __shared__ __attribute__((aligned(16))) float xs[2 * 128 * 4];     // [2][MT<=128][4]
  __shared__ __attribute__((aligned(16))) float ws[K2_W_CHUNK];      // [8][2][32][4]
  __shared__ int obase[128];                                         // output voxel index of tap (0,0,0) or -1
  const int MT = TZ * TY * TX;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int CPH = K2In<MODE>::CPH, CPC = 2 * CPH;
  const int CIB = (Cin + CPC - 1) / CPC;
  const int cob = blockIdx.y;
  int b = blockIdx.x;
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  const int z0 = tiz * TZ, y0 = tiy * TY, x0 = tix * TX;

  // one staged float4 per thread: voxel tid >> 1, half tid & 1
  const int sv = tid >> 1, hh = tid & 1;
  int goff = -1;
  if (sv < MT) {
    const int t = seg3d_fdiv(sv, 1.0f / (float)TX);
    const int tx = sv - t * TX;
    const int tz = seg3d_fdiv(t, 1.0f / (float)TY);
    const int ty = t - tz * TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    if (gz < Di && gy < Hi && gx < Wi) {
      goff = (((n * Di + gz) * Hi + gy) * Wi + gx) * Cin + hh * CPH;
      if (hh == 0) obase[sv] = (((n * 2 * Di + 2 * gz) * Ho + 2 * gy) * Wo + 2 * gx);
    } else if (hh == 0) {
      obase[sv] = -1;
    }
  }
  const int abase = (lh * MT + wave * 32 + li) * 4;  // rows >= MT read garbage inside xs (never stored)
  const int bbase = (lh * 32 + li) * 4;
  constexpr int NACC = PAIR ? 4 : 8;
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // register prefetch of the next chunk (one float4 of x + two of weights per thread) behind the MFMAs of this one.
  // (Measured at the top level, 64 -> 16 at 48^3 -> 96^3: a ring of 8 x chunks in flight, and a whole-K variant that stages
  // the complete x tile and weight image at once, changed nothing -- 116..129 us either way: per wave the epilogue's ~800
  // vector instructions cost as much as its 128 MFMAs, and fp32 MFMA and VALU share one datapath.)
  typename K2In<MODE>::raw xst;
  f32x4 wst[2];
  auto load_chunk = [&](int cib) {
    const bool ok = goff >= 0 && cib * CPC + hh * CPH < Cin;
    xst = K2In<MODE>::load(x, ok ? (i64)goff + cib * CPC : (i64)0);
    const f32x4* wsrc = reinterpret_cast<const f32x4*>(wp + ((i64)cob * CIB + cib) * K2_W_CHUNK);
#pragma unroll
    for (int k = 0; k < 2; ++k) wst[k] = wsrc[tid + k * 256];
  };
  load_chunk(0);
  for (int cib = 0; cib < CIB; ++cib) {
    __syncthreads();
    if (sv < MT) {
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
      *reinterpret_cast<f32x4*>(xs + (hh * MT + sv) * 4) =
          (goff >= 0 && cib * CPC + hh * CPH < Cin) ? K2In<MODE>::cvt(xst) : zero;
    }
    {
      f32x4* wdst = reinterpret_cast<f32x4*>(ws);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + k * 256;          // entry ((tap * 2 + half) * 32 + co) of the standard image
        if (PAIR) {
          const int co = e & 31, th = e >> 5, tap = th >> 1, half = th & 1;
          if (co < 16) wdst[(((tap >> 1) * 2 + half) << 5) + ((tap & 1) << 4) + co] = wst[k];
        } else {
          wdst[e] = wst[k];
        }
      }
    }
    __syncthreads();
    if (cib + 1 < CIB) load_chunk(cib + 1);
    f32x4 av = {0.f, 0.f, 0.f, 0.f};
    if (wave * 32 + li < MT) av = *reinterpret_cast<const f32x4*>(xs + abase);
#pragma unroll
    for (int t = 0; t < NACC; ++t) {
      const f32x4 bw = *reinterpret_cast<const f32x4*>(ws + t * 256 + bbase);
      acc[t] = k2_mfma_step<MODE>(bw, av, acc[t]);
    }
  }

  float s[2] = {0.f, 0.f};
  {
    const int idx = wave * 32 + li;
    const int o = idx < MT ? obase[idx] : -1;
    k2_scatter_epilogue<OUT_BF, ADD, PAIR, NACC>(acc, o, cob, lh, bias, y, addend, lda, Cout, Ho, Wo, s);
  }
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, xs);
    if (tid < 4) {   // slots [tile][4][column block] (the direct kernel: one per 32-voxel sub-tile): the tile's sums in the first
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + ((((i64)n * tiles_per_sample + tile) * 4 + tid) * gridDim.y + cob) * 2;
      dst[0] = tid == 0 ? s[0] : 0.f;
      dst[1] = tid == 0 ? s[1] : 0.f;
    }
  }
}

// The scatter kernel WITHOUT operand staging (round 4), the counterpart of conv3d_k2s2_direct_kernel: a workgroup owns one
// 32-voxel sub-tile and one column block, wave w the tap pair (kz, ky) = (w >> 1, w & 1), kx = 0, 1 -- whose outputs of one input
// voxel are 2 Cout CONTIGUOUS floats.  Operands come straight from global memory into the MFMA registers (x: 16 bytes of the
// lane's voxel row per chunk, two consecutive chunks per step; weights: the coalesced 1-KB image of a (chunk, tap)), a ring of two
// steps in flight, no barrier in the K loop; four times the waves of the staged kernel, each with two accumulators instead of
// eight (the lower levels had 48 - 432 workgroups on 256 CUs with K loops of 16 - 32 chunks).  The epilogue transposes through
// LDS: eight consecutive lanes write (and, ADD, read the addend of) 128 contiguous bytes.
// PAIR (Cout <= 16): ONE accumulator, rows 0..15 = kx 0, 16..31 = kx 1: the lanes read the standard image at their own offsets.
template <int MODE, bool ADD, bool PAIR>
__global__ __launch_bounds__(256, 2) void convT3d_k2s2_direct_kernel(const void* __restrict__ x, const float* __restrict__ wp,
                                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                                       float* __restrict__ stats, int N, int Di, int Hi, int Wi,
                                                                       int Cin, int Cout, int TZ, int TY, int TX, int ntz, int nty,
                                                                       int ntx, const float* __restrict__ addend, int lda) {
  constexpr int NSET = 2, NACC = PAIR ? 1 : 2, ROW = 36;
  __shared__ __attribute__((aligned(16))) float tbuf[4 * 32 * ROW];   // epilogue transpose: per wave [voxel 32][36]
  __shared__ int obs[32];
  __shared__ float red[8];
  const int MT = TZ * TY * TX;
  const int Ho = 2 * Hi, Wo = 2 * Wi;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  constexpr int CPH = K2In<MODE>::CPH, CPC = 2 * CPH, ESZ = K2In<MODE>::ESZ;
  const int CIB = (Cin + CPC - 1) / CPC;
  const int cob = blockIdx.y;
  const int kz = wave >> 1, ky = wave & 1;
  const int lgTX = __builtin_ctz(TX), lgTY = __builtin_ctz(TY);
  int b = blockIdx.x >> 2;
  const int sub = blockIdx.x & 3;
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  unsigned xoff = K2_OOB;
  {
    const int idx = sub * 32 + li;
    const int tx = idx & (TX - 1), t = idx >> lgTX;
    const int ty = t & (TY - 1), tz = t >> lgTY;
    const int gz = tiz * TZ + tz, gy = tiy * TY + ty, gx = tix * TX + tx;
    int o = -1;
    if (idx < MT && gz < Di && gy < Hi && gx < Wi) {
      o = ((n * 2 * Di + 2 * gz) * Ho + 2 * gy) * Wo + 2 * gx;   // output voxel of tap (0, 0, 0)
      xoff = (unsigned)((((gz * Hi + gy) * Wi + gx) * Cin + lh * CPH) * ESZ);
    }
    if (wave == 0 && lh == 0) obs[li] = o;
  }
  const __amdgpu_buffer_rsrc_t xrs = k2_make_rsrc(reinterpret_cast<const char*>(x) + (i64)n * Di * Hi * Wi * Cin * ESZ,
                                                  (unsigned)Di * Hi * Wi * Cin * ESZ);
  // this lane's weight quad of (chunk cib, accumulator a): float4 index (cib * 8 + tap) * 64 + half * 32 + co
  const int tap0 = 4 * kz + 2 * ky;
  const f32x4* wbase = reinterpret_cast<const f32x4*>(wp + (i64)cob * CIB * K2_W_CHUNK) +
                       (PAIR ? (tap0 + (li >> 4)) * 64 + lh * 32 + (li & 15) : tap0 * 64 + lane);
  f32x16 acc[NACC];
#pragma unroll
  for (int a = 0; a < NACC; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
  typename K2In<MODE>::raw xv[NSET][2];
  f32x4 wv[NSET][2][NACC];
  int ld_cib = 0;
  auto load_step = [&](int set, int j) __attribute__((always_inline)) {
    const int cib = ld_cib + j;
    const unsigned hk = cib * CPC + lh * CPH < Cin ? 0u : K2_OOB;
    xv[set][j] = K2In<MODE>::bload(xrs, xoff | hk, __builtin_amdgcn_readfirstlane((unsigned)(cib * CPC * ESZ)));
#pragma unroll
    for (int a = 0; a < NACC; ++a) wv[set][j][a] = wbase[__builtin_amdgcn_readfirstlane((cib * 8 + a) * 64)];
  };
  auto multiply = [&](int set, int j) __attribute__((always_inline)) {
    const f32x4 xb = K2In<MODE>::cvt(xv[set][j]);
#pragma unroll
    for (int a = 0; a < NACC; ++a) acc[a] = k2_mfma_step<MODE>(wv[set][j][a], xb, acc[a]);
  };
  const int NS = CIB >> 1;   // steps of two chunks; host-checked: CIB % 4 == 0
#pragma unroll
  for (int set = 0; set < NSET; ++set) {
    load_step(set, 0), load_step(set, 1);
    ld_cib += 2;
  }
  for (int st = 0; st + 2 * NSET <= NS; st += NSET) {
#pragma unroll
    for (int set = 0; set < NSET; ++set) {
      multiply(set, 0);
      load_step(set, 0);
      multiply(set, 1);
      load_step(set, 1);
      ld_cib += 2;
    }
  }
#pragma unroll
  for (int set = 0; set < NSET; ++set) multiply(set, 0), multiply(set, 1);

  // ---- epilogue: bias, statistics, transpose, (addend,) full-line stores.  Accumulator a = tap kx = a (PAIR: rows 16.. = kx 1) ----
  __syncthreads();   // obs
  float* tr = tbuf + wave * (32 * ROW);
  const int o_l = obs[li];
  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int a = 0; a < NACC; ++a) {
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int row = 8 * g4 + 4 * lh;                    // MFMA rows row .. row + 3 of this lane
      const int co = PAIR ? (row & 15) : cob * 32 + row;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (o_l >= 0 && co < Cout) {
        f32x4 bv = {0.f, 0.f, 0.f, 0.f};
        if (bias) bv = *reinterpret_cast<const f32x4*>(bias + co);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[e] = acc[a][4 * g4 + e] + bv[e];
          if (!ADD) s[0] += v[e], s[1] += v[e] * v[e];
        }
      }
      *reinterpret_cast<f32x4*>(tr + li * ROW + row) = v;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // lane -> (voxel it * 8 + (lane >> 3), 16-byte piece lane & 7) of the 128 bytes this accumulator holds per voxel:
    //   PAIR: [kx 0: 16 channels][kx 1: 16 channels] = the two output voxels' whole rows; else channels cob * 32 .. + 31 of voxel kx = a
    f32x4 ad[4];
    i64 dst[4];
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int v = it * 8 + (lane >> 3), piece = lane & 7;
      const int o_v = obs[v];
      const int kx = PAIR ? (piece >> 2) : a;
      const int co = PAIR ? 4 * (piece & 3) : cob * 32 + 4 * piece;
      dst[it] = (o_v >= 0 && co < Cout) ? ((i64)o_v + (kz * Ho + ky) * Wo + kx) : (i64)-1;
      if (ADD && dst[it] >= 0) ad[it] = *reinterpret_cast<const f32x4*>(addend + dst[it] * lda + co);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int v = it * 8 + (lane >> 3), piece = lane & 7;
      const int co = PAIR ? 4 * (piece & 3) : cob * 32 + 4 * piece;
      if (dst[it] >= 0) {
        f32x4 o4 = *reinterpret_cast<const f32x4*>(tr + v * ROW + 4 * piece);
        if (ADD) o4 += ad[it];
        *reinterpret_cast<f32x4*>(y + dst[it] * Cout + co) = o4;
      }
    }
    if (NACC > 1) __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  }
  if (stats) {
    block_sum_256<2>(s, red);
    if (tid == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dstp = stats + ((((i64)n * tiles_per_sample + tile) * 4 + sub) * gridDim.y + cob) * 2;
      dstp[0] = s[0];
      dstp[1] = s[1];
    }
  }
}

extern "C" long long seg3d_convT3d_k2s2_mfma_stats_count(int Di, int Hi, int Wi, int Cout) {
  K2Tile t = k2_pick_tile(Di, Hi, Wi);
  return (long long)seg3d_cdiv(Di, t.tz) * seg3d_cdiv(Hi, t.ty) * seg3d_cdiv(Wi, t.tx) * 4 * ((Cout + 31) / 32);   // [tile][4][column block]
}

// x [N][Di][Hi][Wi][Cin] -> y [N][2Di][2Hi][2Wi][Cout];  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 8)
static int k2_scatter_launch(const void* x, int x_bf16, const float* wp, const float* bias, float* y, float* stats, int N,
                             int Di, int Hi, int Wi, int Cin, int Cout, void* stream, int out_bf16 = 0,
                             const void* addend = nullptr, int lda = 0) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_convT3d_k2s2_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && Di > 0 && Hi > 0 && Wi > 0 && Cin > 0 && Cout > 0, "seg3d_convT3d_k2s2_mfma_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0 && (Cout % 4) == 0,
                "seg3d_convT3d_k2s2_mfma_fwd: Cin and Cout must be multiples of 4 (got %d, %d)", Cin, Cout);
  SEG3D_REQUIRE((i64)N * Di * Hi * Wi * 8 * Cout < (1ll << 31) && (i64)N * Di * Hi * Wi * Cin < (1ll << 31),
                "seg3d_convT3d_k2s2_mfma_fwd: tensor exceeds 2^31 elements");
  K2Tile t = k2_pick_tile(Di, Hi, Wi);
  const int ntz = seg3d_cdiv(Di, t.tz), nty = seg3d_cdiv(Hi, t.ty), ntx = seg3d_cdiv(Wi, t.tx);
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "2x2x2 stride-2 MFMA kernels: more than 2^22 tiles");
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  SEG3D_REQUIRE(x_bf16 != 2 || (Cin % 16) == 0, "seg3d_convT3d_k2s2_bf16_fwd: the bf16 weight image needs Cin %% 16 == 0");
  SEG3D_REQUIRE(!addend || ((Cout & 7) == 0 && lda >= Cout && (lda & 3) == 0),
                "seg3d_convT3d_k2s2_scatter_addend: needs Cout %% 8 == 0 and a row stride >= Cout, multiple of 4");
  const bool pair = Cout <= 16 && (Cout & 7) == 0;   // two taps per MFMA (rows (tap & 1) * 16 + co)
#ifndef K2_SCATTER_DIRECT
#define K2_SCATTER_DIRECT 1   // 0: the LDS-staged kernel (kept for same-box A/B builds)
#endif
  // the direct kernel: fp32 outputs, whole steps (Cin a multiple of four chunks), whole 16-byte pieces of whole column blocks
  // -- and where it measured faster (tools/bench_k2.py, same box): every launch with a fused addend (its loads are coalesced too:
  // 168 -> 137 us at the top level, 50 -> 21 at 12^3) and the forward launches whose staged grid is under one workgroup per CU
  // (47 -> 21 us at 6^3, 49 -> 42 at 12^3); the forward at 24^3 / 48^3 is 4 % / 1.5 % faster staged (51 vs 56, 116 vs 118 us)
  const bool direct = K2_SCATTER_DIRECT && !out_bf16 && (x_bf16 == 2 ? Cin % 64 == 0 : Cin % 32 == 0) &&
                      (Cout == 16 || Cout % 32 == 0) && (!addend || (lda % 4) == 0) &&
                      (addend || (i64)grid.x * grid.y < 256);
#define K2_SCATTER_D(MODE_)                                                                                          \
  do {                                                                                                               \
    const dim3 g4(grid.x * 4, grid.y);                                                                               \
    const float* ad_ = reinterpret_cast<const float*>(addend);                                                       \
    if (addend && pair)                                                                                              \
      hipLaunchKernelGGL((convT3d_k2s2_direct_kernel<MODE_, true, true>), g4, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ad_, lda);                \
    else if (pair)                                                                                                   \
      hipLaunchKernelGGL((convT3d_k2s2_direct_kernel<MODE_, false, true>), g4, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ad_, lda);                \
    else if (addend)                                                                                                 \
      hipLaunchKernelGGL((convT3d_k2s2_direct_kernel<MODE_, true, false>), g4, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ad_, lda);                \
    else                                                                                                             \
      hipLaunchKernelGGL((convT3d_k2s2_direct_kernel<MODE_, false, false>), g4, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, ad_, lda);                \
  } while (0)
#define K2_SCATTER(MODE_, OB_)                                                                                       \
  do {                                                                                                               \
    if (addend && pair)                                                                                              \
      hipLaunchKernelGGL((convT3d_k2s2_mfma_kernel<MODE_, OB_, true, true>), grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, \
                         y, stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, addend, lda);         \
    else if (pair)                                                                                                   \
      hipLaunchKernelGGL((convT3d_k2s2_mfma_kernel<MODE_, OB_, false, true>), grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, \
                         y, stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, addend, lda);         \
    else if (addend)                                                                                                      \
      hipLaunchKernelGGL((convT3d_k2s2_mfma_kernel<MODE_, OB_, true>), grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, addend, lda);             \
    else                                                                                                             \
      hipLaunchKernelGGL((convT3d_k2s2_mfma_kernel<MODE_, OB_, false>), grid, dim3(256), 0, (hipStream_t)stream, x, wp, bias, y, \
                         stats, N, Di, Hi, Wi, Cin, Cout, t.tz, t.ty, t.tx, ntz, nty, ntx, addend, lda);             \
  } while (0)
  if (direct && x_bf16 == 2) K2_SCATTER_D(2);
  else if (direct && x_bf16) K2_SCATTER_D(1);
  else if (direct) K2_SCATTER_D(0);
  else if (x_bf16 == 2 && out_bf16) K2_SCATTER(2, true);
  else if (x_bf16 == 2) K2_SCATTER(2, false);
  else if (x_bf16 && out_bf16) K2_SCATTER(1, true);
  else if (x_bf16) K2_SCATTER(1, false);
  else K2_SCATTER(0, false);
#undef K2_SCATTER
#undef K2_SCATTER_D
  SEG3D_LAUNCH_CHECK("seg3d_convT3d_k2s2_mfma_fwd");
  return SEG3D_OK;
}

extern "C" int seg3d_convT3d_k2s2_mfma_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats,
                                           int N, int Di, int Hi, int Wi, int Cin, int Cout, void* stream) {
  return k2_scatter_launch(x, 0, wp, bias, y, stats, N, Di, Hi, Wi, Cin, Cout, stream);
}

extern "C" int seg3d_convT3d_k2s2_bf16_fwd(const void* x_bf16, const void* wp, const float* bias, void* y, float* stats,
                                           int N, int Di, int Hi, int Wi, int Cin, int Cout, int out_bf16, int w_bf16,
                                           void* stream) {
  return k2_scatter_launch(x_bf16, w_bf16 ? 2 : 1, reinterpret_cast<const float*>(wp), bias, reinterpret_cast<float*>(y),
                           stats, N, Di, Hi, Wi, Cin, Cout, stream, out_bf16);
}

// data-gradient form with a second gradient of the same tensor folded into the epilogue: y = scatter(x) + addend.
// x_mode: 0 fp32 x / fp32 image, 1 bf16 x / fp32 image, 2 bf16 x / bf16 image; addend has y's dtype (bf16 iff out_bf16)
// and row stride ld_addend elements (a channel slice of a wider tensor); Cout % 8 == 0
extern "C" int seg3d_convT3d_k2s2_scatter_addend(const void* x, int x_mode, const void* wp, const void* addend, int ld_addend,
                                                 void* y, int N, int Di, int Hi, int Wi, int Cin, int Cout, int out_bf16,
                                                 void* stream) {
  SEG3D_REQUIRE(addend && x_mode >= 0 && x_mode <= 2 && (x_mode != 0 || !out_bf16),
                "seg3d_convT3d_k2s2_scatter_addend: bad arguments");
  return k2_scatter_launch(x, x_mode, reinterpret_cast<const float*>(wp), nullptr, reinterpret_cast<float*>(y), nullptr, N, Di,
                           Hi, Wi, Cin, Cout, stream, out_bf16, addend, ld_addend);
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

// BF: P and Q are bf16 (bf16 mode: activations and their gradients), widened to fp32 when the tile is written to LDS
template <bool BF>
__global__ __launch_bounds__(256, 2) void k2_wgrad_mfma_kernel(const void* __restrict__ P, const void* __restrict__ Q,
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
  typename Seg3dQuad<BF>::raw pst[PE], qst[QE];
  unsigned okmask = 0;  // zero-select deferred to store_tile: a select right at the load would serialise the loads
  // tile-invariant part of every entry's address (tile dims are compile-time constants: shifts and masks only); a
  // tile inside the volume adds its origin to these -- two instructions per load instead of ~70 of index arithmetic,
  // which at 64 MFMAs per tile and wave decided the kernel's speed
  int prel[PE], qrel[QE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int v = (tid + e * 256) >> 3;
    const int hx = v % K2W_HX, hy = (v / K2W_HX) % K2W_HY, hz = v / (K2W_HX * K2W_HY);
    prel[e] = ((hz * Hp + hy) * Wp + hx) * CA + a0 + 4 * q;
  }
#pragma unroll
  for (int e = 0; e < QE; ++e) {
    const int v = (tid + e * 256) >> 3;
    const int tx = v % K2W_TX, ty = (v / K2W_TX) % K2W_TY, tz = v / (K2W_TX * K2W_TY);
    qrel[e] = ((tz * Hq + ty) * Wq + tx) * CB + b0 + 4 * q;
  }
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto load_tile = [&](int tile) {
    int b = tile;
    int qd = seg3d_fdiv(b, rNTX);
    const int tix = b - qd * ntx; b = qd;
    qd = seg3d_fdiv(b, rNTY);
    const int tiy = b - qd * nty; b = qd;
    qd = seg3d_fdiv(b, rNTZ);
    const int tiz = b - qd * ntz;
    const int n = qd;
    const int z0 = tiz * K2W_TZ, y0 = tiy * K2W_TY, x0 = tix * K2W_TX;
    if (z0 + K2W_TZ <= Dq && y0 + K2W_TY <= Hq && x0 + K2W_TX <= Wq) {  // whole tile inside the volume
      const i64 pbase = ((((i64)n * Dp + 2 * z0) * Hp + 2 * y0) * Wp + 2 * x0) * CA;
      const i64 qbase = ((((i64)n * Dq + z0) * Hq + y0) * Wq + x0) * CB;
      const unsigned pm = pq_ok ? (1u << PE) - 1u : 0u, qm = qq_ok ? ((1u << QE) - 1u) << PE : 0u;
#pragma unroll
      for (int e = 0; e < PE; ++e) pst[e] = Seg3dQuad<BF>::load(P, pq_ok ? pbase + prel[e] : (i64)0);
#pragma unroll
      for (int e = 0; e < QE; ++e) qst[e] = Seg3dQuad<BF>::load(Q, qq_ok ? qbase + qrel[e] : (i64)0);
      okmask = pm | qm;
      return;
    }
    okmask = 0;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const int v = (tid + e * 256) >> 3;
      const int hx = v % K2W_HX;
      const int t = v / K2W_HX;
      const int hy = t % K2W_HY;
      const int hz = t / K2W_HY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      const bool ok = pq_ok && gz < Dp && gy < Hp && gx < Wp;
      pst[e] = Seg3dQuad<BF>::load(P, ok ? ((((i64)n * Dp + gz) * Hp + gy) * Wp + gx) * CA + a0 + 4 * q : (i64)0);
      okmask |= (ok ? 1u : 0u) << e;
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
      qst[e] = Seg3dQuad<BF>::load(Q, ok ? ((((i64)n * Dq + gz) * Hq + gy) * Wq + gx) * CB + b0 + 4 * q : (i64)0);
      okmask |= (ok ? 1u : 0u) << (PE + e);
    }
  };
  auto store_tile = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < PE; ++e)
      *reinterpret_cast<f32x4*>(ps + ((tid + e * 256) >> 3) * 32 + 4 * q) =
          ((okmask >> e) & 1u) ? Seg3dQuad<BF>::cvt(pst[e]) : zero;
#pragma unroll
    for (int e = 0; e < QE; ++e)
      *reinterpret_cast<f32x4*>(qs + ((tid + e * 256) >> 3) * 32 + 4 * q) =
          ((okmask >> (PE + e)) & 1u) ? Seg3dQuad<BF>::cvt(qst[e]) : zero;
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

// ---------------------------------------------------------------------------------------------------------------
// The same weight gradient for CA <= 16 (fp32; the two stride-2 layers of the top level: P = the 16-channel tensor).  With
// 32 MFMA rows for 16 channels half of every MFMA was idle; here the two taps of a wave share one MFMA -- rows
// (tap & 1) * 16 + a -- so a tile is 32 MFMAs per wave instead of 64, the P tile is staged as [voxel][16] (half the LDS and
// half the staging loads, none of them masked), and the voxel-pair loop is fully unrolled (constant LDS offsets; the
// rolled loop spent ~10 vector instructions of index arithmetic per pair beside 2 MFMAs).  Same partial layout and reduce
// kernel (rows a >= 16 of a slab are never written and never used).
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k2_wgrad_pair_kernel(const float* __restrict__ P, const float* __restrict__ Q,
                                                                 float* __restrict__ part, int N, int Dq, int Hq, int Wq, int CA,
                                                                 int CB, int ntz, int nty, int ntx, int ntiles, int BB32) {
  __shared__ __attribute__((aligned(16))) float ps[K2W_NV * 16];
  __shared__ __attribute__((aligned(16))) float qs[K2W_MT * 32];
  const int Dp = 2 * Dq, Hp = 2 * Hq, Wp = 2 * Wq;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int b0 = (blockIdx.y % BB32) * 32;
  // this lane's MFMA row: tap 2 wave + (li >> 4), channel li & 15
  const int tap = wave * 2 + (li >> 4);
  const int aoff = ((((tap >> 2) * K2W_HY + ((tap >> 1) & 1)) * K2W_HX + (tap & 1)) + 2 * lh) * 16 + (li & 15);
  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  const int q4 = tid & 3, q8 = tid & 7;
  const bool pq_ok = 4 * q4 < CA;
  const bool qq_ok = b0 + 4 * q8 < CB;
  constexpr int PE = (K2W_NV * 4) / 256, QE = (K2W_MT * 8) / 256;   // 8 + 2 16-byte loads per thread and tile
  f32x4 pst[PE], qst[QE];
  unsigned okmask = 0;
  int prel[PE], qrel[QE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int v = (tid + e * 256) >> 2;
    const int hx = v % K2W_HX, hy = (v / K2W_HX) % K2W_HY, hz = v / (K2W_HX * K2W_HY);
    prel[e] = ((hz * Hp + hy) * Wp + hx) * CA + 4 * q4;
  }
#pragma unroll
  for (int e = 0; e < QE; ++e) {
    const int v = (tid + e * 256) >> 3;
    const int tx = v % K2W_TX, ty = (v / K2W_TX) % K2W_TY, tz = v / (K2W_TX * K2W_TY);
    qrel[e] = ((tz * Hq + ty) * Wq + tx) * CB + b0 + 4 * q8;
  }
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto load_tile = [&](int tile) {
    int b = tile;
    int qd = seg3d_fdiv(b, rNTX);
    const int tix = b - qd * ntx; b = qd;
    qd = seg3d_fdiv(b, rNTY);
    const int tiy = b - qd * nty; b = qd;
    qd = seg3d_fdiv(b, rNTZ);
    const int tiz = b - qd * ntz;
    const int n = qd;
    const int z0 = tiz * K2W_TZ, y0 = tiy * K2W_TY, x0 = tix * K2W_TX;
    if (z0 + K2W_TZ <= Dq && y0 + K2W_TY <= Hq && x0 + K2W_TX <= Wq) {  // whole tile inside the volume
      const i64 pbase = ((((i64)n * Dp + 2 * z0) * Hp + 2 * y0) * Wp + 2 * x0) * CA;
      const i64 qbase = ((((i64)n * Dq + z0) * Hq + y0) * Wq + x0) * CB;
      const unsigned pm = pq_ok ? (1u << PE) - 1u : 0u, qm = qq_ok ? ((1u << QE) - 1u) << PE : 0u;
#pragma unroll
      for (int e = 0; e < PE; ++e) pst[e] = *reinterpret_cast<const f32x4*>(P + (pq_ok ? pbase + prel[e] : (i64)0));
#pragma unroll
      for (int e = 0; e < QE; ++e) qst[e] = *reinterpret_cast<const f32x4*>(Q + (qq_ok ? qbase + qrel[e] : (i64)0));
      okmask = pm | qm;
      return;
    }
    okmask = 0;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const int v = (tid + e * 256) >> 2;
      const int hx = v % K2W_HX;
      const int t = v / K2W_HX;
      const int hy = t % K2W_HY;
      const int hz = t / K2W_HY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      const bool ok = pq_ok && gz < Dp && gy < Hp && gx < Wp;
      pst[e] = *reinterpret_cast<const f32x4*>(P + (ok ? ((((i64)n * Dp + gz) * Hp + gy) * Wp + gx) * CA + 4 * q4 : (i64)0));
      okmask |= (ok ? 1u : 0u) << e;
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
      qst[e] = *reinterpret_cast<const f32x4*>(Q + (ok ? ((((i64)n * Dq + gz) * Hq + gy) * Wq + gx) * CB + b0 + 4 * q8 : (i64)0));
      okmask |= (ok ? 1u : 0u) << (PE + e);
    }
  };
  auto store_tile = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < PE; ++e)
      *reinterpret_cast<f32x4*>(ps + ((tid + e * 256) >> 2) * 16 + 4 * q4) = ((okmask >> e) & 1u) ? pst[e] : zero;
#pragma unroll
    for (int e = 0; e < QE; ++e)
      *reinterpret_cast<f32x4*>(qs + ((tid + e * 256) >> 3) * 32 + 4 * q8) = ((okmask >> (PE + e)) & 1u) ? qst[e] : zero;
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    const float* pa = ps + aoff;
    const float* qb = qs + lh * 32 + li;
#pragma unroll
    for (int kp = 0; kp < K2W_MT / 2; ++kp) {   // voxels 2 kp (lane half 0) and 2 kp + 1: neighbours in one row
      const int v = 2 * kp;
      const int tx = v % K2W_TX, ty = (v / K2W_TX) % K2W_TY, tz = v / (K2W_TX * K2W_TY);
      const float a = pa[(((2 * tz) * K2W_HY + 2 * ty) * K2W_HX + 2 * tx) * 16];
      const float bvv = qb[v * 32];
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bvv, acc, 0, 0, 0);
    }
  }
  float* dst = part + ((i64)blockIdx.x * gridDim.y + blockIdx.y) * 8 * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = k2_row(r, lh);
    dst[(wave * 2 + (row >> 4)) * 1024 + (row & 15) * 32 + li] = acc[r];
  }
}

// ---------------------------------------------------------------------------------------------------------------
// pair-reduce weight gradient on the bf16 matrix cores (bf16 mode): the tiles stay bf16 in LDS ([voxel][32 channels],
// 64-byte rows, written by the same register-staged pipeline without widening), and ds_read_b64_tr_b16 -- the transposing
// LDS read, four rows with independent addresses per 16-lane group -- turns them into the K-major MFMA operands
// (8 consecutive Q voxels of one channel per lane; for P the rows are the stride-2 positions 2v + t of those voxels).
// 4 K-steps x 2 taps per wave = 8 v_mfma_f32_32x32x16_bf16 per tile instead of 64 fp32 MFMAs: the kernel is bound by its
// tile loads.  Same partial layout and reduce kernel as k2_wgrad_mfma_kernel.  Needs CA % 8 == 0 and CB % 8 == 0.
// ---------------------------------------------------------------------------------------------------------------
typedef short k2_s16x4 __attribute__((ext_vector_type(4)));
#define K2_TR_READ(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
__device__ __forceinline__ k2_bf16x8 k2_tr_join(k2_s16x4 lo, k2_s16x4 hi) {
  return __builtin_bit_cast(k2_bf16x8, __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7));
}

__global__ __launch_bounds__(256, 2) void k2_wgrad_bf16_mfma_kernel(const seg3d_bf16* __restrict__ P,
                                                                      const seg3d_bf16* __restrict__ Q,
                                                                      float* __restrict__ part, int N, int Dq, int Hq,
                                                                      int Wq, int CA, int CB, int ntz, int nty, int ntx,
                                                                      int ntiles, int BB32) {
  __shared__ __attribute__((aligned(16))) seg3d_bf16 ps[K2W_NV * 32];   // 32 KB
  __shared__ __attribute__((aligned(16))) seg3d_bf16 qs[K2W_MT * 32];   //  4 KB
  const int Dp = 2 * Dq, Hp = 2 * Hq, Wp = 2 * Wq;
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int ab = blockIdx.y / BB32, bb = blockIdx.y % BB32;
  const int a0 = ab * 32, b0 = bb * 32;
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  // staging: entry = (voxel, 8-channel quarter); 16 bytes per entry
  const int q8 = tid & 3;
  const bool pq_ok = a0 + 8 * q8 < CA;
  const bool qq_ok = b0 + 8 * q8 < CB;
  constexpr int PE = (K2W_NV * 4) / 256, QE = (K2W_MT * 4) / 256;   // 8, 1
  f32x4 pst[PE], qst[QE];
  unsigned okmask = 0;
  int prel[PE], qrel[QE];
#pragma unroll
  for (int e = 0; e < PE; ++e) {
    const int v = (tid + e * 256) >> 2;
    const int hx = v % K2W_HX, hy = (v / K2W_HX) % K2W_HY, hz = v / (K2W_HX * K2W_HY);
    prel[e] = ((hz * Hp + hy) * Wp + hx) * CA + a0 + 8 * q8;
  }
#pragma unroll
  for (int e = 0; e < QE; ++e) {
    const int v = (tid + e * 256) >> 2;
    const int tx = v % K2W_TX, ty = (v / K2W_TX) % K2W_TY, tz = v / (K2W_TX * K2W_TY);
    qrel[e] = ((tz * Hq + ty) * Wq + tx) * CB + b0 + 8 * q8;
  }
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto ld16 = [](const seg3d_bf16* p, i64 e) { return *reinterpret_cast<const f32x4*>(p + e); };
  auto load_tile = [&](int tile) {
    int b = tile;
    int qd = seg3d_fdiv(b, rNTX);
    const int tix = b - qd * ntx; b = qd;
    qd = seg3d_fdiv(b, rNTY);
    const int tiy = b - qd * nty; b = qd;
    qd = seg3d_fdiv(b, rNTZ);
    const int tiz = b - qd * ntz;
    const int n = qd;
    const int z0 = tiz * K2W_TZ, y0 = tiy * K2W_TY, x0 = tix * K2W_TX;
    if (z0 + K2W_TZ <= Dq && y0 + K2W_TY <= Hq && x0 + K2W_TX <= Wq) {  // whole tile inside the volume
      const i64 pbase = ((((i64)n * Dp + 2 * z0) * Hp + 2 * y0) * Wp + 2 * x0) * CA;
      const i64 qbase = ((((i64)n * Dq + z0) * Hq + y0) * Wq + x0) * CB;
      const unsigned pm = pq_ok ? (1u << PE) - 1u : 0u, qm = qq_ok ? ((1u << QE) - 1u) << PE : 0u;
#pragma unroll
      for (int e = 0; e < PE; ++e) pst[e] = ld16(P, pq_ok ? pbase + prel[e] : (i64)0);
#pragma unroll
      for (int e = 0; e < QE; ++e) qst[e] = ld16(Q, qq_ok ? qbase + qrel[e] : (i64)0);
      okmask = pm | qm;
      return;
    }
    okmask = 0;
#pragma unroll
    for (int e = 0; e < PE; ++e) {
      const int v = (tid + e * 256) >> 2;
      const int hx = v % K2W_HX;
      const int t = v / K2W_HX;
      const int hy = t % K2W_HY;
      const int hz = t / K2W_HY;
      const int gz = 2 * z0 + hz, gy = 2 * y0 + hy, gx = 2 * x0 + hx;
      const bool ok = pq_ok && gz < Dp && gy < Hp && gx < Wp;
      pst[e] = ld16(P, ok ? ((((i64)n * Dp + gz) * Hp + gy) * Wp + gx) * CA + a0 + 8 * q8 : (i64)0);
      okmask |= (ok ? 1u : 0u) << e;
    }
#pragma unroll
    for (int e = 0; e < QE; ++e) {
      const int v = (tid + e * 256) >> 2;
      const int tx = v % K2W_TX;
      const int t = v / K2W_TX;
      const int ty = t % K2W_TY;
      const int tz = t / K2W_TY;
      const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
      const bool ok = qq_ok && gz < Dq && gy < Hq && gx < Wq;
      qst[e] = ld16(Q, ok ? ((((i64)n * Dq + gz) * Hq + gy) * Wq + gx) * CB + b0 + 8 * q8 : (i64)0);
      okmask |= (ok ? 1u : 0u) << (PE + e);
    }
  };
  auto store_tile = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int e = 0; e < PE; ++e)
      *reinterpret_cast<f32x4*>(ps + ((tid + e * 256) >> 2) * 32 + 8 * q8) = ((okmask >> e) & 1u) ? pst[e] : zero;
#pragma unroll
    for (int e = 0; e < QE; ++e)
      *reinterpret_cast<f32x4*>(qs + ((tid + e * 256) >> 2) * 32 + 8 * q8) = ((okmask >> (PE + e)) & 1u) ? qst[e] : zero;
  };
  // transposing-read addresses (bytes): lane 4q + p of a 16-lane group supplies row q, channels 4p..4p+3 of the group's 16
  const int l16 = lane & 15, rq = l16 >> 2, rp = l16 & 3;
  const int colb = (16 * ((lane >> 4) & 1) + 4 * rp) * 2;
  const int r0 = 8 * lh + rq, r1 = r0 + 4;
  const unsigned qbase_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)qs);
  const unsigned pbase_lds = (unsigned)(size_t)((__attribute__((address_space(3))) char*)ps);
  const unsigned qa0 = qbase_lds + r0 * 64 + colb, qa1 = qbase_lds + r1 * 64 + colb;
  // P row of Q voxel v = 16 s + r, tap (kz, ky, kx): ((2 tz + kz) HY + 2 ty + ky) HX + 2 tx + kx with tz = s >> 1,
  // ty = 2 (s & 1) + (r >> 3), tx = r & 7  ->  per-lane part below, per-(step, tap) part as the asm offset
  unsigned pa0[2], pa1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int tap = wave * 2 + j;
    const int kz = tap >> 2, ky = (tap >> 1) & 1, kx = tap & 1;
    const int tapb = ((kz * K2W_HY + ky) * K2W_HX + kx) * 64;
    pa0[j] = pbase_lds + tapb + ((2 * (r0 >> 3)) * K2W_HX + 2 * (r0 & 7)) * 64 + colb;
    pa1[j] = pbase_lds + tapb + ((2 * (r1 >> 3)) * K2W_HX + 2 * (r1 & 7)) * 64 + colb;
  }
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    k2_s16x4 blo[4], bhi[4], alo[4][2], ahi[4][2];
    // step s: Q rows 16 s .. 16 s + 15; P offset of the step: ((2 (s >> 1)) HY + 4 (s & 1)) HX rows
#define K2W3_POFF(S) ((((2 * ((S) >> 1)) * K2W_HY + 4 * ((S) & 1)) * K2W_HX) * 64)
#define K2W3_STEP_READS(S)                                   \
  K2_TR_READ(blo[S], qa0, (S) * 16 * 64);                    \
  K2_TR_READ(bhi[S], qa1, (S) * 16 * 64);                    \
  K2_TR_READ(alo[S][0], pa0[0], K2W3_POFF(S));               \
  K2_TR_READ(ahi[S][0], pa1[0], K2W3_POFF(S));               \
  K2_TR_READ(alo[S][1], pa0[1], K2W3_POFF(S));               \
  K2_TR_READ(ahi[S][1], pa1[1], K2W3_POFF(S));
    K2W3_STEP_READS(0) K2W3_STEP_READS(1) K2W3_STEP_READS(2) K2W3_STEP_READS(3)
#undef K2W3_STEP_READS
#undef K2W3_POFF
    asm volatile("s_waitcnt lgkmcnt(0)"
                 : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]),
                   "+v"(bhi[3]), "+v"(alo[0][0]), "+v"(ahi[0][0]), "+v"(alo[0][1]), "+v"(ahi[0][1]), "+v"(alo[1][0]),
                   "+v"(ahi[1][0]), "+v"(alo[1][1]), "+v"(ahi[1][1]));
    asm volatile("" : "+v"(alo[2][0]), "+v"(ahi[2][0]), "+v"(alo[2][1]), "+v"(ahi[2][1]), "+v"(alo[3][0]), "+v"(ahi[3][0]),
                      "+v"(alo[3][1]), "+v"(ahi[3][1]));
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const k2_bf16x8 bop = k2_tr_join(blo[st], bhi[st]);
#pragma unroll
      for (int j = 0; j < 2; ++j)
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(k2_tr_join(alo[st][j], ahi[st][j]), bop, acc[j], 0, 0, 0);
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

// the same reduce with a float4 per lane (one 16-byte load per slab) and G waves of a workgroup striding over the slabs, for
// the many-slab levels (as conv3d_k3_wgrad_reduce_kernel in conv_mfma.hip): the 4-byte form above ran 512 slabs x 32 KB in
// 15.6 us (1.1 TB/s)
template <int G>
__global__ __launch_bounds__(64 * G) void k2_wgrad_reduce4_kernel(const float* __restrict__ part, float* __restrict__ dw, int slabs,
                                                                   int A, int B, int BB32, int npairs, i64 sa, i64 sb,
                                                                   int accumulate) {
  constexpr int T = 8;
  typedef float k2f4 __attribute__((ext_vector_type(4)));
  __shared__ k2f4 red[G * 64];
  const i64 totalq = (i64)npairs * T * 256;                     // float4 quads per slab
  const i64 qidx = (i64)blockIdx.x * 64 + (threadIdx.x & 63);
  const int g = threadIdx.x >> 6;
  k2f4 s0 = {0.f, 0.f, 0.f, 0.f}, s1 = s0;
  if (qidx < totalq) {
    const k2f4* p = reinterpret_cast<const k2f4*>(part) + qidx;
    int k = g;
    for (; k + G < slabs; k += 2 * G) {                           // two loads in flight per trip, fixed order
      const k2f4 v0 = p[(i64)k * totalq], v1 = p[(i64)(k + G) * totalq];
      s0 += v0;
      s1 += v1;
    }
    if (k < slabs) s0 += p[(i64)k * totalq];
  }
  red[threadIdx.x] = s0 + s1;
  __syncthreads();
  if (g == 0 && qidx < totalq) {
    k2f4 v = red[threadIdx.x];
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

static int k2_wgrad_slabs(int N, int Dq, int Hq, int Wq, int npairs, int CA) {
  const int ntiles = N * seg3d_cdiv(Dq, K2W_TZ) * seg3d_cdiv(Hq, K2W_TY) * seg3d_cdiv(Wq, K2W_TX);
  (void)CA;
  int slabs = 512 / npairs;   // 2 workgroups per CU (4 of the paired-tap kernel measured slower: twice the slabs to reduce)
  if (slabs > (ntiles + 3) / 4) slabs = (ntiles + 3) / 4;
  if (slabs < 1) slabs = 1;
  return slabs;
}

extern "C" long long seg3d_k2_mfma_wgrad_workspace_floats(int N, int Dq, int Hq, int Wq, int CA, int CB) {
  const int npairs = ((CA + 31) / 32) * ((CB + 31) / 32);
  return (long long)k2_wgrad_slabs(N, Dq, Hq, Wq, npairs, CA) * npairs * 8 * 1024;
}

// P [N][2Dq][2Hq][2Wq][CA], Q [N][Dq][Hq][Wq][CB];  dw[a*sa + b*sb + t] (t < 8) receives the gradient
static int k2_wgrad_launch(const void* P, const void* Q, int bf16, float* dw, float* workspace, int N, int Dq, int Hq,
                           int Wq, int CA, int CB, long long sa, long long sb, int accumulate, void* stream) {
  SEG3D_REQUIRE(P && Q && dw && workspace, "seg3d_k2_mfma_wgrad: null pointer");
  SEG3D_REQUIRE(N > 0 && Dq > 0 && Hq > 0 && Wq > 0 && CA > 0 && CB > 0, "seg3d_k2_mfma_wgrad: bad dims");
  SEG3D_REQUIRE((CA % 4) == 0 && (CB % 4) == 0, "seg3d_k2_mfma_wgrad: channel counts must be multiples of 4 (got %d, %d)",
                CA, CB);
  const int AB32 = (CA + 31) / 32, BB32 = (CB + 31) / 32;
  const int npairs = AB32 * BB32;
  const int ntz = seg3d_cdiv(Dq, K2W_TZ), nty = seg3d_cdiv(Hq, K2W_TY), ntx = seg3d_cdiv(Wq, K2W_TX);
  const int ntiles = N * ntz * nty * ntx;
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "seg3d_k2_mfma_wgrad: more than 2^22 tiles");
  const int slabs = k2_wgrad_slabs(N, Dq, Hq, Wq, npairs, CA);
  hipStream_t s = (hipStream_t)stream;
  if (bf16 && (CA & 7) == 0 && (CB & 7) == 0)
    hipLaunchKernelGGL(k2_wgrad_bf16_mfma_kernel, dim3(slabs, npairs), dim3(256), 0, s,
                       reinterpret_cast<const seg3d_bf16*>(P), reinterpret_cast<const seg3d_bf16*>(Q), workspace, N, Dq, Hq,
                       Wq, CA, CB, ntz, nty, ntx, ntiles, BB32);
  else if (bf16)
    hipLaunchKernelGGL(k2_wgrad_mfma_kernel<true>, dim3(slabs, npairs), dim3(256), 0, s, P, Q, workspace, N, Dq, Hq, Wq, CA,
                       CB, ntz, nty, ntx, ntiles, BB32);
  else if (CA <= 16)   // two taps per MFMA (rows (tap & 1) * 16 + a)
    hipLaunchKernelGGL(k2_wgrad_pair_kernel, dim3(slabs, npairs), dim3(256), 0, s, reinterpret_cast<const float*>(P),
                       reinterpret_cast<const float*>(Q), workspace, N, Dq, Hq, Wq, CA, CB, ntz, nty, ntx, ntiles, BB32);
  else
    hipLaunchKernelGGL(k2_wgrad_mfma_kernel<false>, dim3(slabs, npairs), dim3(256), 0, s, P, Q, workspace, N, Dq, Hq, Wq, CA,
                       CB, ntz, nty, ntx, ntiles, BB32);
  SEG3D_LAUNCH_CHECK("seg3d_k2_mfma_wgrad");
  const i64 total = (i64)npairs * 8 * 1024;
  if (slabs >= 32)
    hipLaunchKernelGGL(k2_wgrad_reduce4_kernel<16>, dim3((unsigned)((total / 4 + 63) / 64)), dim3(1024), 0, s, workspace, dw, slabs,
                       CA, CB, BB32, npairs, (i64)sa, (i64)sb, accumulate);
  else
    hipLaunchKernelGGL(k2_wgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, workspace, dw, slabs, CA, CB,
                       BB32, npairs, (i64)sa, (i64)sb, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_k2_mfma_wgrad(reduce)");
  return SEG3D_OK;
}

extern "C" int seg3d_k2_mfma_wgrad(const float* P, const float* Q, float* dw, float* workspace, int N, int Dq, int Hq,
                                   int Wq, int CA, int CB, long long sa, long long sb, int accumulate, void* stream) {
  return k2_wgrad_launch(P, Q, 0, dw, workspace, N, Dq, Hq, Wq, CA, CB, sa, sb, accumulate, stream);
}

// bf16 mode: P and Q bf16, dw and workspace fp32 (same sizes as seg3d_k2_mfma_wgrad_workspace_floats)
extern "C" int seg3d_k2_bf16_wgrad(const void* P_bf16, const void* Q_bf16, float* dw, float* workspace, int N, int Dq,
                                   int Hq, int Wq, int CA, int CB, long long sa, long long sb, int accumulate,
                                   void* stream) {
  return k2_wgrad_launch(P_bf16, Q_bf16, 1, dw, workspace, N, Dq, Hq, Wq, CA, CB, sa, sb, accumulate, stream);
}
