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
//   * tile 8 x 8 x 8 voxels = 128 output quads x 32 output channels per item; EIGHT waves, two per SIMD: wave (zw, h)
//     owns the z planes 2 zw, 2 zw + 1 (32 quads) and the output channels 16 h .. 16 h + 15 of the column block:
//     16 point accumulators x 2 planes x 4 registers = 128 of its 256 registers, on v_mfma_f32_16x16x4_f32;
//   * K chunks of FOUR input channels (the weight image of a chunk is 48 x 512 B = 24 KB; chunks of 8 would not fit);
//   * everything a chunk needs arrives or is made INSIDE the previous chunk's MFMA loop, one barrier per chunk: while chunk c
//     is multiplied, the weights of chunk c + 1 and the RAW halo tile [10 x 10 x 10 voxels][4] of chunk c + 2 land by
//     LDS-DMA (steps 0-4), and the RAW tile of chunk c + 1 is transformed into the second T buffer
//     T[p = 16][z = 10][quad = 16][4] in eight stages spread over steps 12-33 (320 (z, quad, channel pair) tasks, 40 per wave,
//     packed fp32 adds: 32 vector instructions per wave and chunk beside 96 MFMAs);
//   * 48 steps of 2 MFMAs per chunk: (kz, point); one A read (16 channels x 4 K) and two B reads (16 quads x 4 K each) per
//     step, read two steps ahead (the weight image is [t][co][4], T is [p][z][quad][4]: a wave's reads are contiguous);
//   * the weight image is the T = 48 pack of seg3d_pack_weights_mfma: t = kz * 16 + py * 4 + px, laid out per 8-channel chunk
//     [t][half][32 co][4] -- a 4-channel K chunk is one half;
//   * bias and fused addend enter through the accumulators at item start, the epilogue contains no load (see the kernel);
//   * LDS: 2 x 16 KB RAW + 2 x 40 KB T + 2 x 24 KB weights = 160 KB, all of it.
// Levels that are whole 4 x 4 x 4 cells but not 8 x 8 x 8 tiles (the 12^3 level) run the same algorithm on CELLS
// (conv3d_k3_wino2d_c4_kernel, further down): a cell's 4 z planes x 2 x 2 quads are the 16 columns of one MFMA group.
// Rounds 2 and early 3 ran this with FOUR waves (one per SIMD, 32 quads x 32 channels on v_mfma_f32_32x32x2_f32, 256
// accumulator registers): the eight-wave form issues the same instructions per SIMD in the K loop (that loop is bound by what
// it issues: 96 MFMAs + ~100 vector / LDS-write / DMA instructions on the one fp32 pipe) but halves the
// epilogue's exposed time (eight waves' stores in flight), frees half the registers -- which is what lets the addend go
// through the accumulators without a spill -- and measured 4-7 % faster per launch on every shape.
#include "seg3d_common.h"
#include <type_traits>
#include <utility>
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
#define W2_LDS_FLOATS (2 * W2_RAW + 2 * W2_T + 2 * W2_W)  // 40960 floats = 163840 bytes: all of the CU's LDS

__device__ __attribute__((aligned(16))) float w2_zero16[4];   // DMA source for zero padding

// LDS-DMA of 16 bytes per lane, written as inline assembly ON PURPOSE: while a __builtin_amdgcn_global_load_lds is
// outstanding hipcc's wait-count pass treats it as a pending FLAT access and turns every LDS wait of the loop into
// lgkmcnt(0) (tools/ubench/waitcnt_dma.hip) -- a full drain of the operand reads issued for the following steps, which one
// wave per SIMD cannot hide.  The compiler does not see this load, so the kernel waits for it itself (w2_dma_wait) before
// the barrier that publishes the data.  Nothing else in this kernel uses M0.
typedef __attribute__((address_space(3))) float w2_lds_float;
__device__ __forceinline__ void w2_glds16(const float* src, float* lds_dst_wave_uniform) {
  // (readfirstlane: the "s" operand must be an SGPR also when the compiler cannot prove the address uniform; folded away when it can)
  const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds_dst_wave_uniform);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(off) : "memory", "m0");
}
// the same with a wave-uniform base (SGPR pair) and a per-lane byte offset: no vector instruction forms the address
__device__ __forceinline__ void w2_glds16_sbase(const float* base_wave_uniform, unsigned lane_byte_offset, float* lds_dst_wave_uniform) {
  const unsigned off = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds_dst_wave_uniform);
  const uint64_t b = (uint64_t)(uintptr_t)base_wave_uniform;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const uint64_t sb = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_byte_offset), "s"(sb), "s"(off) : "memory", "m0");
}
// one ds_read_b64 that stays one (volatile: hipcc otherwise pairs neighbouring reads into ds_read2_b64, which costs 8 LDS
// cycles per pair and banks modulo 32 in 16-lane groups, against 2 cycles per ds_read_b64 with 64 banks over 32-lane halves:
// MI355X_MICROARCH.md, LDS table); the explicit LDS address space keeps the volatile access from becoming a FLAT load
typedef __attribute__((address_space(3))) f32x2 w2_lds_f32x2;
__device__ __forceinline__ f32x2 w2_lds_read_b64(const float* p) {
  return *reinterpret_cast<const volatile w2_lds_f32x2*>((w2_lds_float*)p);
}
// likewise one ds_read_b32 with a 16-bit immediate offset (merged into ds_read2_b32 the 8-bit offsets cost a v_add_u32 per point,
// and in this fp32-MFMA loop every vector instruction is paid in full)
__device__ __forceinline__ float w2_lds_read_b32(const float* p) {
  return *reinterpret_cast<const volatile w2_lds_float*>((w2_lds_float*)p);
}
// The forward kernels' form: the LDS destination as a wave-uniform BYTE OFFSET inside the workgroup's LDS (no generic -> LDS
// pointer conversion with its null check per issue: six scalar instructions each)
__device__ __forceinline__ void w2_glds16_at(const float* src, unsigned lds_byte_off_wave_uniform) {
  lds_byte_off_wave_uniform = __builtin_amdgcn_readfirstlane(lds_byte_off_wave_uniform);
  asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" : : "v"(src), "s"(lds_byte_off_wave_uniform) : "memory", "m0");
}
__device__ __forceinline__ void w2_glds16_sbase_at(const float* base_wave_uniform, unsigned lane_byte_offset, unsigned lds_byte_off_wave_uniform) {
  lds_byte_off_wave_uniform = __builtin_amdgcn_readfirstlane(lds_byte_off_wave_uniform);
  const uint64_t b = (uint64_t)(uintptr_t)base_wave_uniform;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)b), hi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
  const uint64_t sb = ((uint64_t)hi << 32) | lo;
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" : : "v"(lane_byte_offset), "s"(sb), "s"(lds_byte_off_wave_uniform) : "memory", "m0");
}
// LDS-DMA through a BUFFER RESOURCE (base + num_records in four scalars): lane l's 16 bytes at byte offset voff_l land at
// M0 + 16 l, and a lane whose offset is out of range -- past the end, or "negative" = wrapped -- leaves ZEROS (measured:
// tools/ubench/buffer_lds_test.hip).  The weight-gradient kernel gets its z padding from that range check and its (y, x) padding
// from a per-column offset table, so a piece costs ONE vector add where the pointer form needed and / compare / two selects / a
// 64-bit add.  `pred` (wave-uniform) is tested by a scalar branch inside the block: a skipped piece costs no issue slot at all.
typedef int w2_srd __attribute__((ext_vector_type(4)));
__device__ __forceinline__ w2_srd w2_make_srd(const void* base_wave_uniform, unsigned num_bytes) {
  const uint64_t b = (uint64_t)(uintptr_t)base_wave_uniform;
  w2_srd r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)b);
  r[1] = __builtin_amdgcn_readfirstlane((int)(unsigned)((b >> 32) & 0xffffu));   // stride 0: a raw buffer, range-checked by byte offset
  r[2] = __builtin_amdgcn_readfirstlane((int)num_bytes);
  r[3] = 0x00020000;                                                              // gfx9 family: DATA_FORMAT = 32 bit, untyped
  return r;
}
__device__ __forceinline__ void w2_bufdma16_if(int pred_wave_uniform, unsigned voff, w2_srd srd, unsigned lds_byte_off_wave_uniform) {
  pred_wave_uniform = __builtin_amdgcn_readfirstlane(pred_wave_uniform);
  lds_byte_off_wave_uniform = __builtin_amdgcn_readfirstlane(lds_byte_off_wave_uniform);
  asm volatile("s_cmp_lg_u32 %3, 0\n\ts_cbranch_scc0 1f\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds\n1:"
               : : "v"(voff), "s"(srd), "s"(lds_byte_off_wave_uniform), "s"(pred_wave_uniform) : "memory", "m0", "scc");
}
// the same unconditionally, with a wave-uniform byte offset in the instruction's scalar-offset field (the K chunk's channels)
__device__ __forceinline__ void w2_bufdma16_soff(unsigned voff, w2_srd srd, unsigned soff_wave_uniform, unsigned lds_byte_off_wave_uniform) {
  soff_wave_uniform = __builtin_amdgcn_readfirstlane(soff_wave_uniform);
  lds_byte_off_wave_uniform = __builtin_amdgcn_readfirstlane(lds_byte_off_wave_uniform);
  asm volatile("s_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %2 offen lds"
               : : "v"(voff), "s"(srd), "s"(soff_wave_uniform), "s"(lds_byte_off_wave_uniform) : "memory", "m0");
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
__device__ __forceinline__ f32x2 w2_pk_fma(f32x2 a, f32x2 b, f32x2 c) {
  f32x2 d;
  asm("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}

#ifdef W2_STAMPS   // diagnostic build only (tools/ubench/wino2d_stamp.hip): s_memtime stamps of the first chunks of a few workgroups
__device__ long long* w2_stamp_buf;
#define W2_STAMP(chunk, k)                                                                                        \
  do {                                                                                                            \
    if (blockIdx.x < 8 && (chunk) < 64 && (threadIdx.x & 63) == 0)                                                \
      w2_stamp_buf[((blockIdx.x * 8 + (threadIdx.x >> 6)) * 64 + (chunk)) * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define W2_STAMP(chunk, k)
#endif

#define W2_NW 8                                   // waves per workgroup
#define W2_XPW 2                                  // raw pieces per wave (16 / 8)
#define W2_WPW 3                                  // weight pieces per wave (24 / 8)
#ifndef W2_TR0
#define W2_TR0 12                                 // first step of the RAW -> T transform stages
#endif
#ifndef W2_PF
#define W2_PF 2                                   // operand prefetch distance in steps
#endif

// step -> (kz, point) of a K chunk: seg3d_w2_step_kz / seg3d_w2_step_pi / seg3d_w2_point (seg3d_common.h) -- points in pairs, a
// pair's six steps a0 b0 a1 b1 a2 b2, the four corner points 0, 3, 12, 15 (whose accumulators may still be waiting for a fused
// addend, see the kernel) in the last two pairs = steps 36-47
// f(integral_constant<int, 0>) ... f(integral_constant<int, N - 1>): the steps index the accumulators with constant expressions
template <int LO, class F, int... I>
__device__ __forceinline__ void w2_steps(F&& f, std::integer_sequence<int, I...>) {
  (f(std::integral_constant<int, LO + I>{}), ...);
}
// steps of a chunk multiplied BEFORE the chunk's barrier; the rest (operands already in registers) runs behind it, in the shadow
// of the next chunk's first operand reads and address arithmetic
#ifndef W2_HEAD
#define W2_HEAD 42
#endif

// the lane's four bias values bias[ub + 4 kq .. + 3] from the wave's 16 (ub is wave-uniform), each made a SCALAR by
// readfirstlane right after its load.  A plain per-lane vector load puts its vmcnt wait in front of the first MFMA that reads
// the accumulator -- step 3 of EVERY chunk of the one loop body, where it also waits for the chunk's own LDS-DMAs (cell kernel,
// 12^3 256 -> 256: 0.176 ms with, 0.160 without) -- and hipcc turns a select among uniformly addressed loads back into exactly
// that load; values already in SGPRs cannot be folded.  (An inline-assembly s_load_dwordx16 faulted: nothing makes the hazard
// recogniser wait between the v_readfirstlane that forms its base and the scalar load.)
__device__ __forceinline__ f32x4 w2_bias_quad(const float* __restrict__ bias, int ub, int kq) {
  ub = __builtin_amdgcn_readfirstlane(ub);
  float bs[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) bs[r] = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(bias[ub + r])));
  f32x4 b4;
#pragma unroll
  for (int r = 0; r < 4; ++r) b4[r] = kq == 0 ? bs[r] : (kq == 1 ? bs[4 + r] : (kq == 2 ? bs[8 + r] : bs[12 + r]));
  return b4;
}

template <bool BIAS, bool ADD>
__global__ __launch_bounds__(512, 1) void conv3d_k3_wino2d_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                                      float* __restrict__ stats, int N, int D, int H, int W,
                                                                      int Cin, int Cout, int ntz, int nty, int ntx, int ncog,
                                                                      int nitems, const float* __restrict__ addend, int sk_per,
                                                                      float* __restrict__ skws) {
  // sk_per > 0: STREAM-K.  The launch's nitems * NSC K chunks are dealt to the workgroups in contiguous ranges of sk_per chunks
  // (sk_per >= NSC, so an item is cut at most once); a workgroup whose range starts / ends inside an item writes that piece's
  // outputs -- plain partial sums in the output domain, Y = A^T M A is linear -- into its slab 0 / 1 of skws instead of y, and
  // conv3d_k3_wino2d_sk_finish_kernel adds the two pieces of every cut item.  The piece that holds chunk 0 carries the bias
  // and the fused addend as usual, the other starts from zero: the finish pass only adds.  No workgroup waits for another.
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                               // [2][NV][4] (+ padding)
  float* timg = lds + 2 * W2_RAW;                 // [2][16][10][16][4]
  float* wbuf = lds + 2 * W2_RAW + 2 * W2_T;      // [2][48][32][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int zw = wave & 3, hh = wave >> 2;        // z-plane pair, output-channel half
  const int l16 = lane & 15, kq = lane >> 4;      // MFMA column (quad / channel) and K index (= D row group)
  const int NSC = Cin >> 2;
  const int AB = Cin >> 3;
  const int G = gridDim.x;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz, rNCOG = 1.0f / (float)ncog;

  // RAW pieces of this wave come through a buffer resource of the SAMPLE (w2_bufdma16 below): xconst = the byte offset of the
  // lane's halo voxel relative to the tile origin (negative in front of it), OOB for the padding slots 1000 .. 1023; z padding =
  // the resource's range check (plane -1 wraps, plane D is past the end), (y, x) padding = a select per item on xflag
  const unsigned OOB = 0x80000000u;                    // stays out of range after any tile / channel offset (sample bytes < 2^31)
  unsigned xconst[W2_XPW];
  int xflag[W2_XPW];
#pragma unroll
  for (int j = 0; j < W2_XPW; ++j) {
    const int e = (wave + W2_NW * j) * 64 + lane;
    xconst[j] = OOB;
    xflag[j] = 0;
    if (e < W2_NV) {   // RAW slot e holds the halo voxel (hz, hy, hx ^ swizzle): see the LDS images below
      const int t = e / W2_H;
      const int hxs = e - t * W2_H;
      const int hz = t / W2_H;
      const int hy = t - hz * W2_H;
      const int hx = hxs ^ ((hy >> 1) & 1);
      xconst[j] = (unsigned)((((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin * 4);
      xflag[j] = (hy == 0 ? 4 : 0) | (hy == W2_H - 1 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == W2_H - 1 ? 32 : 0);
#ifdef W2_EXP_COMPACT   // (diagnostic: what would the kernel cost if a chunk's RAW tile were 16 contiguous KB -- a channel-group-planar layout?)
      xconst[j] = (unsigned)(e * 16), xflag[j] = 0;
#endif
    }
  }
  const unsigned xbytes = (unsigned)D * H * W * Cin * 4u;   // one sample
  // ---- LDS images, laid out for conflict-free access (tools/lds_bank_sim.py models every access kind below; round 3's images
  // [t][co][4] / [p][z][quad][4] made every operand read a 2-way conflict: lanes (l16, kq) and (l16 + 8, kq) on one bank) ----
  //   weights [g 12][kq 4][co 32][4 steps]: a lane's A operands of the steps 4 g .. 4 g + 3 are ONE ds_read_b128
  //   T       [p 16][h 2][z 10][quad 16][2]: channel 2 h + c of (plane z, quad) at c; a 32-lane half (16 quads x kq pair) reads 32
  //           consecutive floats (ds_read_b32); the planes 2 zw .. 2 zw + 3 of a point are read once for all three kz
  //   RAW     [slot][4], slot = (hz * 10 + hy) * 10 + (hx ^ ((hy >> 1) & 1)): x pairs swapped in every other row pair, so that the 16
  //           quads of a plane sit in 16 different 16-byte bank groups for the transform's ds_read_b64 (unswizzled: quads
  //           (qy, qx + 2) and (qy + 1, qx) collide); the LDS-DMA lanes fetch any permutation for free
  const int abase = (kq * 32 + 16 * hh + l16) * 4;                          // + g * 512
  const int bbase = (kq >> 1) * 320 + (2 * zw) * 32 + l16 * 2 + (kq & 1);    // + p * 640 + j * 32: plane 2 zw + j
  // transform task of this lane: 320 (plane z, channel pair h, quad) tasks = ten blocks [h 2][quad 16] of 32; lanes 0-31 of wave w
  // take block w whole, lanes 32-39 an eighth of blocks 8 / 9 (lanes >= 40 repeat lane 39): a 16-lane write group is 16 quads
  // of one (z, h) = 32 consecutive floats of T, a 32-lane read group 16 quads x 2 pairs
  const int t_l = lane < 40 ? lane : 39;
  const int t_task = t_l < 32 ? wave * 32 + t_l : (8 + (wave >> 2)) * 32 + 8 * (wave & 3) + (t_l - 32);
  const int t_z = t_task >> 5, t_h = (t_task >> 4) & 1, t_q = t_task & 15;
  const int t_src = ((t_z * W2_H + 2 * (t_q >> 2)) * W2_H + 2 * (t_q & 3)) * 4 + 2 * t_h;
  // swizzle of the patch rows 2 qy + r: x ^= (qy + (r >> 1)) & 1 -> columns k = 0, 2 move by + 1 voxel, k = 1, 3 by - 1
  const int t_sw0 = 4 * ((t_q >> 2) & 1), t_sw1 = 4 - t_sw0;
  const int t_srcE[2] = {t_src + t_sw0, t_src + t_sw1}, t_srcO[2] = {t_src - t_sw0, t_src - t_sw1};
  const int t_dst = t_h * (W2_H * W2_NQ * 2) + t_z * (W2_NQ * 2) + t_q * 2;

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
  int item = blockIdx.x, istride = G, ilimit = nitems;
  int sc_first = 0, sc_last_end = NSC, sk_wg = 0;   // stream-K: chunk range inside the first / last item of this workgroup, its index
  if (sk_per > 0) {
    sk_wg = (G & 7) == 0 ? (int)(blockIdx.x & 7) * (G >> 3) + (int)(blockIdx.x >> 3) : (int)blockIdx.x;   // neighbouring ranges on one XCD
    const int g0 = sk_wg * sk_per, total = nitems * NSC;
    const int g1 = g0 + sk_per < total ? g0 + sk_per : total;
    if (g0 >= g1) return;
    const float rNSC = 1.0f / (float)NSC;
    item = __builtin_amdgcn_readfirstlane(fdiv(g0, rNSC));
    sc_first = g0 - item * NSC;
    const int last = __builtin_amdgcn_readfirstlane(fdiv(g1 - 1, rNSC));
    sc_last_end = g1 - last * NSC;
    istride = 1;
    ilimit = last + 1;
  } else if ((G & 7) == 0) {
    const int per_xcd = (nitems + 7) >> 3, xcd = blockIdx.x & 7;
    item = xcd * per_xcd + (blockIdx.x >> 3);
    istride = G >> 3;
    ilimit = (xcd + 1) * per_xcd < nitems ? (xcd + 1) * per_xcd : nitems;
  }
  if (item >= ilimit) return;
  const int first_item = item;

  int fx_item = item, fx_sc = sc_first;
  unsigned xvoff[W2_XPW];   // byte offset of the lane's halo voxel inside the sample of the item being fetched (OOB: padding)
  w2_srd xsrd;
  auto fx_setup = [&](int it) {
    int n, z0, y0, x0, cog, tile;
    decode(it, n, z0, y0, x0, cog, tile);
    const int fyx = (y0 == 0 ? 4 : 0) | (y0 + W2_TS >= H ? 8 : 0) | (x0 == 0 ? 16 : 0) | (x0 + W2_TS >= W ? 32 : 0);
    const unsigned origin = (unsigned)(((z0 * H + y0) * W + x0) * Cin * 4);
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) xvoff[j] = (xflag[j] & fyx) ? OOB : xconst[j] + origin;
    xsrd = w2_make_srd(x + (i64)n * D * H * W * Cin, xbytes);
  };
  auto fx_advance = [&]() {
    ++fx_sc;
    if (fx_sc == NSC) {
      fx_sc = 0;
      if (fx_item + istride < ilimit) fx_item += istride;
      fx_setup(fx_item);
    }
  };
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds);   // byte offset of the carve-up
  auto dma_x = [&](int j, unsigned rdst_float_off) {   // rdst: float offset of a RAW buffer inside the carve-up; channels 4 fx_sc .. + 3
#ifdef W2_EXP_COMPACT
    w2_bufdma16_soff(xvoff[j], xsrd, (unsigned)fx_sc * 16384u, lds0 + (rdst_float_off + (wave + W2_NW * j) * 256) * 4);
#else
    w2_bufdma16_soff(xvoff[j], xsrd, (unsigned)fx_sc * 16u, lds0 + (rdst_float_off + (wave + W2_NW * j) * 256) * 4);
#endif
  };
  int fw_item = item, fw_sc = sc_first;
  auto cog_of = [&](int it) { return __builtin_amdgcn_readfirstlane(it - fdiv(it, rNCOG) * ncog); };
  int fw_cog = cog_of(fw_item);
  const int cog_step = cog_of(istride);   // the column block advances by istride mod ncog per item: scalar adds, no division in the loop
  auto fw_src = [&]() { return wp + ((i64)fw_cog * AB + (fw_sc >> 1)) * (48 * 256) + (fw_sc & 1) * W2_W; };   // [bb][ab][h]: 6144 floats each
  auto fw_advance = [&]() {
    ++fw_sc;
    if (fw_sc == NSC) {
      fw_sc = 0;
      if (fw_item + istride < ilimit) {
        fw_item += istride;
        fw_cog += cog_step;
        if (fw_cog >= ncog) fw_cog -= ncog;
      }
    }
  };
  const unsigned w_lane_off = lane * 16;   // bytes: the image of a chunk is a straight copy, 24 pieces of 1 KiB
  auto dma_w = [&](int j, const float* wsrc, unsigned wdst_float_off) {
    const int piece = wave + W2_NW * j;
    w2_glds16_sbase_at(wsrc + piece * 256, w_lane_off, lds0 + (wdst_float_off + piece * 256) * 4);
  };
  // RAW -> T for one channel pair of one (z, quad) item: V = B^T d B, 16 points, packed over the pair
#ifdef W2_EXP_GNPRO
  const f32x2 gn_scale = {1.0f + 1e-3f * (float)(tid & 3), 1.0f}, gn_shift = {1e-3f * (float)(tid & 1), 0.f}, gn_zero = {-1e30f, -1e30f};
#endif
  f32x2 rd[2][4];
  f32x2 dxp[4][4];   // [row][px]
  auto tr_read = [&](const float* rw, int r) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      rd[r & 1][k] = w2_lds_read_b64(rw + ((k & 1) ? t_srcO[r >> 1] : t_srcE[r >> 1]) + r * (W2_H * 4) + 4 * k);
#ifdef W2_EXP_GNPRO   // pricing build: what a GroupNorm + ReLU prologue on the operand path would cost -- three
                      // instructions per RAW element pair (scale / shift per channel and sample, ReLU); halo lanes are not kept at 0 here
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      f32x2 v = w2_pk_fma(rd[r & 1][k], gn_scale, gn_shift);   // (gfx950 has no packed fp32 max: the ReLU is two instructions)
      v[0] = fmaxf(v[0], gn_zero[0]);
      v[1] = fmaxf(v[1], gn_zero[1]);
      rd[r & 1][k] = v;
    }
#endif
  };
  auto tr_x = [&](int r) {
    const f32x2* d = rd[r & 1];
    dxp[r][0] = w2_pk_sub(d[0], d[2]);
    dxp[r][1] = w2_pk_add(d[1], d[2]);
    dxp[r][2] = w2_pk_sub(d[2], d[1]);
    dxp[r][3] = w2_pk_sub(d[3], d[1]);   // column px = 3 with the opposite sign (see the accumulators)
  };
  auto tr_y = [&](float* tdst, int px) {
    float* dst = tdst + t_dst + px * (W2_H * W2_NQ * 4);   // point p = 4 py + px at p * 640
    *reinterpret_cast<f32x2*>(dst + 0 * 4 * (W2_H * W2_NQ * 4)) = w2_pk_sub(dxp[0][px], dxp[2][px]);
    *reinterpret_cast<f32x2*>(dst + 1 * 4 * (W2_H * W2_NQ * 4)) = w2_pk_add(dxp[1][px], dxp[2][px]);
    *reinterpret_cast<f32x2*>(dst + 2 * 4 * (W2_H * W2_NQ * 4)) = w2_pk_sub(dxp[2][px], dxp[1][px]);
    *reinterpret_cast<f32x2*>(dst + 3 * 4 * (W2_H * W2_NQ * 4)) = w2_pk_sub(dxp[3][px], dxp[1][px]);   // row py = 3 likewise
  };

  fx_setup(item);
  {
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, 0);
    fx_advance();
    const float* w0 = fw_src();
#pragma unroll
    for (int j = 0; j < W2_WPW; ++j) dma_w(j, w0, 2 * W2_RAW + 2 * W2_T);
    fw_advance();
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, W2_RAW);
    fx_advance();
  }
  w2_dma_wait();
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    tr_read(raw, r);
    tr_x(r);
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) tr_y(timg, px);
  __syncthreads();

  // ---- accumulators [point][plane of the pair]: D[co = 16 hh + 4 kq + r][quad l16] ----
  // Nothing but MFMAs ever adds to them: the bias AND a fused addend enter at item start.  Y = A^T M A with
  // A^T = [1 1 1 0; 0 1 -1 -1] takes M[1][1] into all four outputs of the quad with weight + 1 (bias), and M[0][0], M[0][3],
  // M[3][0], M[3][3] into exactly one output each: (0,0) +, (0,1) -, (1,0) -, (1,1) +.  This kernel computes the points of
  // column px = 3 and of row py = 3 with the opposite sign (the RAW -> T transform emits d3 - d1 instead of d1 - d3 there, at
  // no cost, and the output transform adds where it would subtract), which makes all four weights + 1: the addend's 2 x 2
  // values of a quad are LOADED INTO the accumulators of points 0, 3, 12, 15.  Those loads are issued in the previous item's
  // epilogue (in front of its stores: a load behind stores waits for them) and first needed at step 36 of the item's first
  // chunk (the step order below visits these four points last), so the addend costs neither registers nor an exposed
  // round trip: the epilogue with addend was 9 300 cycles against 3 500 without.
  f32x4 acc[16][2];
  auto quad_voxel = [&](int n, int z0, int y0, int x0) __attribute__((always_inline)) {   // first voxel of this lane's quad in plane 2 zw
    return ((n * D + z0 + 2 * zw) * H + y0 + 2 * (l16 >> 2)) * W + x0 + 2 * (l16 & 3);
  };
  auto acc_init_plane = [&](int zz, int vq, int co) __attribute__((always_inline)) {   // addend (or zero) into the four corner points of plane zz
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = (k >> 1) * 12 + (k & 1) * 3;
      if (ADD) acc[p][zz] = *reinterpret_cast<const f32x4*>(addend + (i64)(vq + zz * H * W + (k >> 1) * W + (k & 1)) * Cout + co);
      else acc[p][zz] = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto acc_init_rest = [&](int co) __attribute__((always_inline)) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (BIAS) b4 = w2_bias_quad(bias, co - 4 * kq, kq);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if (p == 0 || p == 3 || p == 12 || p == 15) continue;
      acc[p][0] = (BIAS && p == 5) ? b4 : f32x4{0.f, 0.f, 0.f, 0.f};
      acc[p][1] = (BIAS && p == 5) ? b4 : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  int cur_n, cur_z0, cur_y0, cur_x0, cur_cog, cur_tile;
  decode(item, cur_n, cur_z0, cur_y0, cur_x0, cur_cog, cur_tile);
  if (sc_first > 0) {   // (stream-K) this workgroup starts inside an item: a plain partial sum, bias and addend are in the other piece
#pragma unroll
    for (int p = 0; p < 16; ++p) acc[p][0] = acc[p][1] = f32x4{0.f, 0.f, 0.f, 0.f};
  } else {
    const int vq = quad_voxel(cur_n, cur_z0, cur_y0, cur_x0), co = cur_cog * 32 + 16 * hh + 4 * kq;
    acc_init_plane(0, vq, co);
    acc_init_plane(1, vq, co);
    acc_init_rest(co);
  }

  int ci_ = 0;
#ifdef W2_STAMPS
  int gchunk = 0;
#endif
  // operands are read ahead of their use: the A word of the steps 4 g + 4 .. 4 g + 7 at step 4 g, the eight B values of the next
  // pair of points (four planes each) during the first four steps of the current pair; the first word and pair of a chunk right
  // behind the previous chunk's barrier, in front of that chunk's last W2_HEAD .. 47 steps
  f32x4 a4[12];
  float bz[16][4];
  auto preload = [&](int ci) __attribute__((always_inline)) {
    a4[0] = *reinterpret_cast<const f32x4*>(wbuf + ci * W2_W + abase);
    const float* tb = timg + ci * W2_T + bbase;
#pragma unroll
    for (int q = 0; q < 8; ++q) bz[q >> 2][q & 3] = w2_lds_read_b32(tb + seg3d_w2_point(q >> 2) * (W2_H * W2_NQ * 4) + (q & 3) * (W2_NQ * 2));
  };
  preload(0);
  for (;;) {
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;
    const int sc_b = item == first_item ? sc_first : 0, sc_e = more_items ? NSC : sc_last_end;   // (0, NSC unless stream-K cut the item)
    for (int sc = sc_b; sc < sc_e; ++sc) {
      const float* ws = wbuf + ci_ * W2_W;
      const float* tcur = timg + ci_ * W2_T + bbase;
      const unsigned wdst1 = 2 * W2_RAW + 2 * W2_T + (ci_ ^ 1) * W2_W;   // (DMA destinations: float offsets inside the carve-up)
      float* tdst1 = timg + (ci_ ^ 1) * W2_T;
      const float* rsrc1 = raw + (ci_ ^ 1) * W2_RAW;
      const unsigned rdst2 = ci_ * W2_RAW;
      const float* wsrc1 = fw_src();
      auto lda4 = [&](int g) { return *reinterpret_cast<const f32x4*>(ws + g * 512 + abase); };          // steps 4 g .. 4 g + 3
      auto ldb = [&](int pi, int j) { return w2_lds_read_b32(tcur + seg3d_w2_point(pi) * (W2_H * W2_NQ * 4) + j * (W2_NQ * 2)); };   // plane 2 zw + j
      W2_STAMP(gchunk, 0);
      auto step = [&](auto st_c) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
        constexpr int pi = seg3d_w2_step_pi(st), kz = seg3d_w2_step_kz(st), pt = seg3d_w2_point(pi);
        constexpr int pair = st / 6, j6 = st % 6;
        if ((st & 3) == 0 && st + 4 < 48) a4[(st + 4) >> 2 < 12 ? (st + 4) >> 2 : 0] = lda4((st + 4) >> 2);
        if (pair + 1 < 8 && j6 < 4) {
          constexpr int npi = (2 * (pair + 1) + (j6 >> 1)) & 15, j0 = 2 * (j6 & 1);
          bz[npi][j0] = ldb(npi, j0), bz[npi][j0 + 1] = ldb(npi, j0 + 1);
        }
        // steps 0-4: DMA issue (weights first, they are needed first); W2_TR0 ..: four row stages, four column stages.  (Issuing the
        // pieces of the two waves of a SIMD at different steps -- 6, 12, 24 steps apart, behind a scalar branch inside the assembly
        // block -- measured 0.7-3 % SLOWER per launch: the issue cost is the instruction's own, not a queue the two waves share)
#ifndef W2_EXP_NODMA   // (W2_EXP_*: diagnostic builds of tools/ubench/wino2d_stamp.hip that drop one ingredient; results are wrong)
        if (st < W2_WPW) dma_w(st, wsrc1, wdst1);
        else if (st - W2_WPW < W2_XPW) dma_x(st - W2_WPW, rdst2);
#endif
#ifndef W2_EXP_NOTR
        if (st >= W2_TR0 && st <= W2_TR0 + 6 && ((st - W2_TR0) & 1) == 0) tr_read(rsrc1, (st - W2_TR0) >> 1);
        if (st >= W2_TR0 + 3 && st <= W2_TR0 + 9 && ((st - W2_TR0) & 1) == 1) tr_x((st - W2_TR0 - 3) >> 1);
        if (st >= W2_TR0 + 12 && st <= W2_TR0 + 21 && (st - W2_TR0 - 12) % 3 == 0) tr_y(tdst1, (st - W2_TR0 - 12) / 3);
#endif
        __builtin_amdgcn_sched_barrier(0);
        acc[pt][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st >> 2][st & 3], bz[pi][kz], acc[pt][0], 0, 0, 0);
        acc[pt][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st >> 2][st & 3], bz[pi][kz + 1], acc[pt][1], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      static_assert(W2_HEAD >= W2_TR0 + 22 && W2_HEAD >= 41 && W2_HEAD <= 48, "every LDS access of a chunk is issued before its barrier");
      w2_steps<0>(step, std::make_integer_sequence<int, W2_HEAD>{});
      W2_STAMP(gchunk, 1);
      w2_dma_wait();
      W2_STAMP(gchunk, 2);
      __syncthreads();   // (s_waitcnt lgkmcnt(0) first: every operand of the steps still to come is in registers)
      W2_STAMP(gchunk, 3);
      ci_ ^= 1;
      preload(ci_);
      w2_steps<W2_HEAD>(step, std::make_integer_sequence<int, 48 - W2_HEAD>{});
      fw_advance();
      fx_advance();
#ifdef W2_STAMPS
      ++gchunk;
#endif
    }

    // ---- output transform Y = A^T M A (signs as explained above) + epilogue, plane by plane: 4 consecutive channels (co_lane ..)
    // of quad l16, packed fp32 math over the channel pairs (0, 1), (2, 3) of an accumulator (aligned register pairs).  After a
    // plane's accumulators have been read, the NEXT item's addend is requested into its corner points, then the plane is stored ----
    int co_lane = cur_cog * 32 + 16 * hh + 4 * kq;
    int vo_q = quad_voxel(cur_n, cur_z0, cur_y0, cur_x0);
    const int stat_slot = (cur_n * (ntz * nty * ntx) + cur_tile) * ncog + cur_cog;
    // where the item's outputs go: y, or (stream-K, a cut item) this workgroup's slab 0 / 1 laid out as one 8^3 tile of 32 channels
    const bool piece = sc_b > 0 || sc_e < NSC;
    float* obase = y;
    int oHW = H * W, oW = W, oC = Cout;
    if (piece) {
      obase = skws + (i64)(2 * sk_wg + (sc_b > 0 ? 0 : 1)) * (W2_TS * W2_TS * W2_TS * 32);
      oHW = W2_TS * W2_TS, oW = W2_TS, oC = 32;
      vo_q = ((2 * zw) * W2_TS + 2 * (l16 >> 2)) * W2_TS + 2 * (l16 & 3);
      co_lane = 16 * hh + 4 * kq;
    }
    int nco_lane = co_lane, nvo_q = vo_q;
    if (more_items) {
      decode(next_item, cur_n, cur_z0, cur_y0, cur_x0, cur_cog, cur_tile);
      nco_lane = cur_cog * 32 + 16 * hh + 4 * kq;
      nvo_q = quad_voxel(cur_n, cur_z0, cur_y0, cur_x0);
    }
    f32x2 s0p = {0.f, 0.f}, s1p = {0.f, 0.f};
    f32x2 v[2][2][2][2];   // [plane][i][j][channel pair]
    auto out_transform = [&](int zz) __attribute__((always_inline)) {
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        f32x2 r0[4], r1[4];
#pragma unroll
        for (int px = 0; px < 4; ++px) {
          const f32x2 m0 = {acc[px][zz][2 * h2], acc[px][zz][2 * h2 + 1]}, m1 = {acc[4 + px][zz][2 * h2], acc[4 + px][zz][2 * h2 + 1]};
          const f32x2 m2 = {acc[8 + px][zz][2 * h2], acc[8 + px][zz][2 * h2 + 1]}, m3 = {acc[12 + px][zz][2 * h2], acc[12 + px][zz][2 * h2 + 1]};
          r0[px] = w2_pk_add(w2_pk_add(m0, m1), m2);
          r1[px] = w2_pk_add(w2_pk_sub(m1, m2), m3);
        }
        v[zz][0][0][h2] = w2_pk_add(w2_pk_add(r0[0], r0[1]), r0[2]);
        v[zz][0][1][h2] = w2_pk_add(w2_pk_sub(r0[1], r0[2]), r0[3]);
        v[zz][1][0][h2] = w2_pk_add(w2_pk_add(r1[0], r1[1]), r1[2]);
        v[zz][1][1][h2] = w2_pk_add(w2_pk_sub(r1[1], r1[2]), r1[3]);
      }
    };
    auto out_store = [&](int zz) __attribute__((always_inline)) {
      const int vo00 = vo_q + zz * oHW;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x2 lo = v[zz][i][j][0], hi = v[zz][i][j][1];
          s0p = w2_pk_add(s0p, w2_pk_add(lo, hi));
          s1p = w2_pk_fma(lo, lo, s1p);
          s1p = w2_pk_fma(hi, hi, s1p);
#ifdef W2_EXP_NOSTORE
          if (lo[0] == 123.456f)
#endif
          *reinterpret_cast<f32x4*>(obase + (i64)(vo00 + i * oW + j) * oC + co_lane) = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
    };
    // (both planes transformed first, ALL eight addend loads, then the sixteen stores -- loads no longer queued behind a plane's
    // stores -- measured the same within 1 %: 8 x 96^3 32->32 1.830 / 1.840 ms, 48^3 64->64 0.493 / 0.484, 24^3 128->128 0.472 / 0.486)
#pragma unroll
    for (int zz = 0; zz < 2; ++zz) {
      out_transform(zz);
      if (more_items) acc_init_plane(zz, nvo_q, nco_lane);
      out_store(zz);
    }
    if (more_items) acc_init_rest(nco_lane);
    if (stats && !piece) {   // (the statistics of a cut item are taken by the finish pass)
      const float s0 = wave_sum(s0p[0] + s0p[1]), s1 = wave_sum(s1p[0] + s1p[1]);
      if (lane == 0) {
        float* dst = stats + ((i64)stat_slot * W2_NW + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
    W2_STAMP(gchunk - 1, 6);
    if (!more_items) break;
    item = next_item;
  }
  w2_dma_wait();
}

// stream-K finish pass of the tile kernel: workgroup w - 1 of the grid owns the cut between the chunk ranges of the logical
// workgroups w - 1 and w (w = 1 .. G - 1); if it falls inside an item, that item's outputs are slab 1 of w - 1 (the piece with
// chunk 0: bias and addend inside) + slab 0 of w, added here, written to y, and the item's eight GroupNorm partial slots taken
__global__ __launch_bounds__(512) void conv3d_k3_wino2d_sk_finish_kernel(const float* __restrict__ skws, float* __restrict__ y,
                                                                         float* __restrict__ stats, int D, int H, int W, int Cout,
                                                                         int ntz, int nty, int ntx, int ncog, int nitems, int NSC,
                                                                         int sk_per) {
  const int w = blockIdx.x + 1;
  const long long cut = (long long)w * sk_per;
  if (cut >= (long long)nitems * NSC || cut % NSC == 0) return;
  const int item = (int)(cut / NSC);
  const int cog = item % ncog;
  int b = item / ncog;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int tile = (tiz * nty + tiy) * ntx + tix;
  constexpr int SLAB = W2_TS * W2_TS * W2_TS * 32;
  const f32x4* A = reinterpret_cast<const f32x4*>(skws + (i64)(2 * (w - 1) + 1) * SLAB);
  const f32x4* B = reinterpret_cast<const f32x4*>(skws + (i64)(2 * w) * SLAB);
  float s0 = 0.f, s1 = 0.f;
#pragma unroll
  for (int it = 0; it < SLAB / 4 / 512; ++it) {
    const int idx = it * 512 + threadIdx.x;   // (voxel of the tile, channel quad)
    const int v = idx >> 3, q = idx & 7;
    const f32x4 a = A[idx], c = B[idx];
    f32x4 o;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      o[r] = a[r] + c[r];
      s0 += o[r];
      s1 = fmaf(o[r], o[r], s1);
    }
    const int lz = v >> 6, ly = (v >> 3) & 7, lx = v & 7;
    *reinterpret_cast<f32x4*>(y + ((i64)((n * D + tiz * W2_TS + lz) * H + tiy * W2_TS + ly) * W + tix * W2_TS + lx) * Cout + cog * 32 + 4 * q) = o;
  }
  if (stats) {
    s0 = wave_sum(s0), s1 = wave_sum(s1);
    if ((threadIdx.x & 63) == 0) {
      float* dst = stats + ((i64)((n * (ntz * nty * ntx) + tile) * ncog + cog) * W2_NW + (threadIdx.x >> 6)) * 2;
      dst[0] = s0;
      dst[1] = s1;
    }
  }
}

// ================================================================================================================
// The same algorithm on 4 x 4 x 4 CELLS (conv3d_k3_wino2d_c4_kernel): levels whose edges are multiples of 4 but not of 8 -- the
// 12^3 levels of the V-Net (down_128 / up_256 residual blocks, network/vnet.py:29-31), which the 8^3-tile kernel cannot tile
// (a 12 x 12 plane is 36 quads).  A cell is 4 z planes x 2 x 2 quads = 16 (z, quad) columns of ONE 16x16x4 MFMA column group,
// with its own 6^3 halo; the T layout [p][cell][z 6][quad 4][4] makes the column (z, quad) of step kz the contiguous address
// (z + kz) * 4 + quad.  An item is FOUR consecutive cells (any four: cells are independent, they may belong to two samples)
// x one 32-channel column block: wave (c, h) owns cell c and the output channels 16 h .. 16 h + 15 -- 16 point accumulators of
// 4 registers, one A read, one B read and one MFMA per step.  Everything else is the tile kernel: K chunks of four channels,
// weights and RAW halo cells by LDS-DMA (16 RAW pieces: four per cell), the transform inside the MFMA loop (192 (cell, z, quad,
// channel pair) tasks, 24 per wave), one barrier per chunk, bias / addend through the accumulators.  Per chunk a SIMD issues
// half the MFMAs of the tile kernel for the same DMA, barrier and three fifths of the transform work, so the executed rate is
// lower (0.5 of the peak) -- on 4/9 of the direct kernel's MFMAs.
// ================================================================================================================
#define C4_H 6                                    // halo edge of a cell
#define C4_NV (C4_H * C4_H * C4_H)                // 216 halo voxels (256 slots = four 1-KiB pieces per cell)
#define C4_T (16 * 4 * 24 * 4)                    // floats of T at four cells per item: [16 p][cell][z 6][quad 4][4] = 6144
#define C4_LDS_FLOATS (2 * W2_RAW + 2 * C4_T + 2 * W2_W)   // 32 + 48 + 48 KB (the same carve-up for two cells per item)

// NC = cells per item: 4 (eight waves) where that gives enough items, else 2 (four waves, one per SIMD: half the MFMAs per chunk
// again, but twice the items -- 4 x 12^3 128 -> 128 is 108 items of four cells, 216 of two)
template <bool BIAS, bool ADD, int NC>
__global__ __launch_bounds__(128 * NC, 1) void conv3d_k3_wino2d_c4_kernel(const float* __restrict__ x, const float* __restrict__ wp,
                                                                      const float* __restrict__ bias, float* __restrict__ y,
                                                                      float* __restrict__ stats, int N, int D, int H, int W, int Cin,
                                                                      int Cout, int ncz, int ncy, int ncx, int ncells, int ncog,
                                                                      int nitems, const float* __restrict__ addend) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* raw = lds;                               // [2][4 cells][256 slots][4]
  float* timg = lds + 2 * W2_RAW;                 // [2][16][4][6][4][4]
  float* wbuf = lds + 2 * W2_RAW + 2 * C4_T;      // [2][48][32][4]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int NWV = 2 * NC;                     // waves: cell x output-channel half
  constexpr int WPW = 24 / NWV;                   // weight pieces per wave; raw pieces per wave: 4 NC / NWV = 2 = W2_XPW
  constexpr int PSTR = NC * 24 * 4;               // floats between two points of T
  const int wc = wave % NC, hh = wave / NC;       // cell of the item, output-channel half
  const int l16 = lane & 15, kq = lane >> 4;      // MFMA column (z = l16 >> 2, quad = l16 & 3) and K index
  const int NSC = Cin >> 2;
  const int AB = Cin >> 3;
  const int G = gridDim.x;
  const int cps = ncz * ncy * ncx;                // cells per sample
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };
  const float rNCX = 1.0f / (float)ncx, rNCY = 1.0f / (float)ncy, rNCZ = 1.0f / (float)ncz, rNCOG = 1.0f / (float)ncog;

  // raw DMA pieces of this wave: piece p = wave + 8 j is slots 64 (p & 3) .. + 63 of cell p >> 2 of the item
  // (through a buffer resource of the cell's sample, as the tile kernel: byte offset relative to the cell origin, (y, x) flags)
  const unsigned OOB = 0x80000000u;
  unsigned xconst[W2_XPW];
  int xflag[W2_XPW];
#pragma unroll
  for (int j = 0; j < W2_XPW; ++j) {
    const int e = ((wave + NWV * j) & 3) * 64 + lane;
    xconst[j] = OOB;
    xflag[j] = 0;
    if (e < C4_NV) {   // RAW slot e of a cell holds the halo voxel (hz, hy, hx ^ swizzle), as in the tile kernel
      const int t = e / C4_H;
      const int hxs = e - t * C4_H;
      const int hz = t / C4_H;
      const int hy = t - hz * C4_H;
      const int hx = hxs ^ ((hy >> 1) & 1);
      xconst[j] = (unsigned)((((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin * 4);
      xflag[j] = (hy == 0 ? 4 : 0) | (hy == C4_H - 1 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == C4_H - 1 ? 32 : 0);
    }
  }
  const unsigned xbytes = (unsigned)D * H * W * Cin * 4u;   // one sample
  // LDS images (conflict-free by tools/lds_bank_sim.py): weights as the tile kernel; T [p][cell][h 2][z 6][quad 4][2] -- the 32
  // lanes (column (z, quad), kq pair) of a read are 32 consecutive floats; RAW slots with the x pairs of every other row pair swapped
  const int abase = (kq * 32 + 16 * hh + l16) * 4;                                              // + g * 512
  const int bbase = wc * 96 + (kq >> 1) * 48 + (l16 >> 2) * 8 + (l16 & 3) * 2 + (kq & 1);       // + p * PSTR + kz * 8
  // transform task of this lane: 192 (cell, z, quad, channel pair) tasks, 24 per wave (lanes >= 24 repeat task 23 of their wave)
  const int t_task = wave * 24 + (lane < 24 ? lane : 23);
  const int t_cell = t_task / 48, t_rem = t_task - 48 * t_cell;
  const int t_i = t_rem >> 1, t_h = t_rem & 1;
  const int t_z = t_i >> 2, t_q = t_i & 3;
  const int t_src = t_cell * 1024 + ((t_z * C4_H + 2 * (t_q >> 1)) * C4_H + 2 * (t_q & 1)) * 4 + 2 * t_h;
  const int t_sw0 = 4 * ((t_q >> 1) & 1), t_sw1 = 4 - t_sw0;   // swizzle of the patch rows 2 qy + r: x ^= (qy + (r >> 1)) & 1
  const int t_srcE[2] = {t_src + t_sw0, t_src + t_sw1}, t_srcO[2] = {t_src - t_sw0, t_src - t_sw1};
  const int t_dst = t_cell * 96 + t_h * 48 + t_z * 8 + t_q * 2;

  // cell -> sample, origin, index inside the sample (cells past the end of a last, partial item are clamped: their waves load
  // and multiply like the others and store nothing)
  auto cell_origin = [&](int cell, int& n, int& z0, int& y0, int& x0, int& cis) {
    int b = cell < ncells ? cell : ncells - 1;
    int q = fdiv(b, rNCX);
    const int cx = b - q * ncx;
    b = q;
    q = fdiv(b, rNCY);
    const int cy = b - q * ncy;
    b = q;
    q = fdiv(b, rNCZ);
    const int cz = b - q * ncz;
    n = q;
    cis = (cz * ncy + cy) * ncx + cx;
    z0 = cz * 4, y0 = cy * 4, x0 = cx * 4;
  };
  auto group_of = [&](int it, int& cog) {
    const int grp = fdiv(it, rNCOG);
    cog = it - grp * ncog;
    return grp;
  };
  int item = blockIdx.x, istride = G, ilimit = nitems;
  if ((G & 7) == 0) {
    const int per_xcd = (nitems + 7) >> 3, xcd = blockIdx.x & 7;
    item = xcd * per_xcd + (blockIdx.x >> 3);
    istride = G >> 3;
    ilimit = (xcd + 1) * per_xcd < nitems ? (xcd + 1) * per_xcd : nitems;
  }
  if (item >= ilimit) return;

  int fx_item = item, fx_sc = 0;
  unsigned xvoff[W2_XPW];
  w2_srd xsrd[W2_XPW];    // (a wave's two pieces belong to two cells, which may lie in two samples)
  auto fx_setup = [&](int it) {
    int cog;
    const int grp = group_of(it, cog);
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) {
      int n, z0, y0, x0, cis;
      cell_origin(NC * grp + ((wave + NWV * j) >> 2), n, z0, y0, x0, cis);
      const int fyx = (y0 == 0 ? 4 : 0) | (y0 + 4 >= H ? 8 : 0) | (x0 == 0 ? 16 : 0) | (x0 + 4 >= W ? 32 : 0);
      const unsigned origin = (unsigned)(((z0 * H + y0) * W + x0) * Cin * 4);
      xvoff[j] = (xflag[j] & fyx) ? OOB : xconst[j] + origin;
      xsrd[j] = w2_make_srd(x + (i64)n * D * H * W * Cin, xbytes);
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
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds);   // byte offset of the carve-up
  auto dma_x = [&](int j, unsigned rdst_float_off) {
    w2_bufdma16_soff(xvoff[j], xsrd[j], (unsigned)fx_sc * 16u, lds0 + (rdst_float_off + (wave + NWV * j) * 256) * 4);
  };
  int fw_item = item, fw_sc = 0;
  auto cog_of = [&](int it) { return __builtin_amdgcn_readfirstlane(it - fdiv(it, rNCOG) * ncog); };
  int fw_cog = cog_of(fw_item);
  const int cog_step = cog_of(istride);
  auto fw_src = [&]() { return wp + ((i64)fw_cog * AB + (fw_sc >> 1)) * (48 * 256) + (fw_sc & 1) * W2_W; };
  auto fw_advance = [&]() {
    ++fw_sc;
    if (fw_sc == NSC) {
      fw_sc = 0;
      if (fw_item + istride < ilimit) {
        fw_item += istride;
        fw_cog += cog_step;
        if (fw_cog >= ncog) fw_cog -= ncog;
      }
    }
  };
  const unsigned w_lane_off = lane * 16;
  auto dma_w = [&](int j, const float* wsrc, unsigned wdst_float_off) {
    const int piece = wave + NWV * j;
    w2_glds16_sbase_at(wsrc + piece * 256, w_lane_off, lds0 + (wdst_float_off + piece * 256) * 4);
  };
  // RAW -> T (signs of column px = 3 / row py = 3 as in the tile kernel)
  f32x2 rd[2][4];
  f32x2 dxp[4][4];
  auto tr_read = [&](const float* rw, int r) {
#pragma unroll
    for (int k = 0; k < 4; ++k)
      rd[r & 1][k] = w2_lds_read_b64(rw + ((k & 1) ? t_srcO[r >> 1] : t_srcE[r >> 1]) + r * (C4_H * 4) + 4 * k);
  };
  auto tr_x = [&](int r) {
    const f32x2* d = rd[r & 1];
    dxp[r][0] = w2_pk_sub(d[0], d[2]);
    dxp[r][1] = w2_pk_add(d[1], d[2]);
    dxp[r][2] = w2_pk_sub(d[2], d[1]);
    dxp[r][3] = w2_pk_sub(d[3], d[1]);
  };
  auto tr_y = [&](float* tdst, int px) {
    float* dst = tdst + t_dst + px * PSTR;   // point p = 4 py + px at p * PSTR
    *reinterpret_cast<f32x2*>(dst + 0 * 4 * PSTR) = w2_pk_sub(dxp[0][px], dxp[2][px]);
    *reinterpret_cast<f32x2*>(dst + 1 * 4 * PSTR) = w2_pk_add(dxp[1][px], dxp[2][px]);
    *reinterpret_cast<f32x2*>(dst + 2 * 4 * PSTR) = w2_pk_sub(dxp[2][px], dxp[1][px]);
    *reinterpret_cast<f32x2*>(dst + 3 * 4 * PSTR) = w2_pk_sub(dxp[3][px], dxp[1][px]);
  };

  fx_setup(item);
  {
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, 0);
    fx_advance();
    const float* w0 = fw_src();
#pragma unroll
    for (int j = 0; j < WPW; ++j) dma_w(j, w0, 2 * W2_RAW + 2 * C4_T);
    fw_advance();
#pragma unroll
    for (int j = 0; j < W2_XPW; ++j) dma_x(j, W2_RAW);
    fx_advance();
  }
  w2_dma_wait();
  __syncthreads();
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    tr_read(raw, r);
    tr_x(r);
  }
#pragma unroll
  for (int px = 0; px < 4; ++px) tr_y(timg, px);
  __syncthreads();

  // accumulators [point]: D[co = 16 hh + 4 kq + r][column l16]; bias and addend enter as in the tile kernel
  f32x4 acc[16];
  // this wave's cell of an item: first voxel of the lane's quad, statistics slot, is it a real cell
  auto my_cell = [&](int it, int& vq, int& co, int& slot, bool& valid) __attribute__((always_inline)) {
    int cog;
    const int grp = group_of(it, cog);
    int n, z0, y0, x0, cis;
    cell_origin(NC * grp + wc, n, z0, y0, x0, cis);
    valid = NC * grp + wc < ncells;
    vq = ((n * D + z0 + (l16 >> 2)) * H + y0 + 2 * ((l16 >> 1) & 1)) * W + x0 + 2 * (l16 & 1);
    co = cog * 32 + 16 * hh + 4 * kq;
    slot = ((n * cps + cis) * ncog + cog) * 2 + hh;
  };
  auto acc_init = [&](int vq, int co) __attribute__((always_inline)) {
    f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
    if (BIAS) b4 = w2_bias_quad(bias, co - 4 * kq, kq);
#pragma unroll
    for (int p = 0; p < 16; ++p) {
      if (ADD && (p == 0 || p == 3 || p == 12 || p == 15)) {
        const int k = (p >= 12 ? 2 : 0) + (p & 1);   // p = 0, 3, 12, 15 -> output (i, j) = (k >> 1, k & 1)
        acc[p] = *reinterpret_cast<const f32x4*>(addend + (i64)(vq + (k >> 1) * W + (k & 1)) * Cout + co);
      } else {
        acc[p] = (BIAS && p == 5) ? b4 : f32x4{0.f, 0.f, 0.f, 0.f};
      }
    }
  };
  int cur_vq, cur_co, cur_slot;
  bool cur_valid;
  my_cell(item, cur_vq, cur_co, cur_slot, cur_valid);
  acc_init(cur_vq, cur_co);

  int ci_ = 0;
  // operands read ahead as in the tile kernel: the A word of four steps one word ahead, B W2_PF steps ahead; the steps W2_HEAD .. 47
  // of a chunk run behind its barrier (all their operands in registers), in front of them the first reads of the next chunk
  f32x4 a4[12];
  float bv[48];
  auto ldb_at = [&](int ci, int s1) __attribute__((always_inline)) {
    return w2_lds_read_b32(timg + ci * C4_T + bbase + seg3d_w2_point(seg3d_w2_step_pi(s1)) * PSTR + seg3d_w2_step_kz(s1) * 8);
  };
  auto preload = [&](int ci) __attribute__((always_inline)) {
    a4[0] = *reinterpret_cast<const f32x4*>(wbuf + ci * W2_W + abase);
#pragma unroll
    for (int st = 0; st < W2_PF; ++st) bv[st] = ldb_at(ci, st);
  };
  preload(0);
  for (;;) {
    const int next_item = item + istride;
    const bool more_items = next_item < ilimit;
    for (int sc = 0; sc < NSC; ++sc) {
      const float* ws = wbuf + ci_ * W2_W;
      const unsigned wdst1 = 2 * W2_RAW + 2 * C4_T + (ci_ ^ 1) * W2_W;   // (DMA destinations: float offsets inside the carve-up)
      float* tdst1 = timg + (ci_ ^ 1) * C4_T;
      const float* rsrc1 = raw + (ci_ ^ 1) * W2_RAW;
      const unsigned rdst2 = ci_ * W2_RAW;
      const float* wsrc1 = fw_src();
      auto lda4 = [&](int g) { return *reinterpret_cast<const f32x4*>(ws + g * 512 + abase); };   // steps 4 g .. 4 g + 3
      auto step = [&](auto st_c) __attribute__((always_inline)) {
        constexpr int st = decltype(st_c)::value;
        constexpr int pt = seg3d_w2_point(seg3d_w2_step_pi(st));
        if ((st & 3) == 0 && st + 4 < 48) a4[(st + 4) >> 2 < 12 ? (st + 4) >> 2 : 0] = lda4((st + 4) >> 2);
        if (st < W2_HEAD && st + W2_PF < 48) bv[(st + W2_PF) % 48] = ldb_at(ci_, st + W2_PF);
        if (st < WPW) dma_w(st, wsrc1, wdst1);
        else if (st - WPW < W2_XPW) dma_x(st - WPW, rdst2);
        if (st >= W2_TR0 && st <= W2_TR0 + 6 && ((st - W2_TR0) & 1) == 0) tr_read(rsrc1, (st - W2_TR0) >> 1);
        if (st >= W2_TR0 + 3 && st <= W2_TR0 + 9 && ((st - W2_TR0) & 1) == 1) tr_x((st - W2_TR0 - 3) >> 1);
        if (st >= W2_TR0 + 12 && st <= W2_TR0 + 21 && (st - W2_TR0 - 12) % 3 == 0) tr_y(tdst1, (st - W2_TR0 - 12) / 3);
        __builtin_amdgcn_sched_barrier(0);
        acc[pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a4[st >> 2][st & 3], bv[st], acc[pt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      };
      w2_steps<0>(step, std::make_integer_sequence<int, W2_HEAD>{});
#pragma unroll
      for (int s1 = W2_HEAD + W2_PF; s1 < 48; ++s1) bv[s1] = ldb_at(ci_, s1);   // the B operands of the steps behind the barrier
      w2_dma_wait();
      __syncthreads();
      ci_ ^= 1;
      {
        const f32x4 a0n = *reinterpret_cast<const f32x4*>(wbuf + ci_ * W2_W + abase);   // (a4[0] / bv[0 ..] are dead by now: W2_HEAD > 4 + W2_PF)
        float b0n[W2_PF];
#pragma unroll
        for (int st = 0; st < W2_PF; ++st) b0n[st] = ldb_at(ci_, st);
        w2_steps<W2_HEAD>(step, std::make_integer_sequence<int, 48 - W2_HEAD>{});
        a4[0] = a0n;
#pragma unroll
        for (int st = 0; st < W2_PF; ++st) bv[st] = b0n[st];
      }
      fw_advance();
      fx_advance();
    }

    // ---- output transform (signs as in the tile kernel) + epilogue: the quad's 2 x 2 voxels x 4 channels ----
    int nvq = cur_vq, nco = cur_co, nslot = cur_slot;
    bool nvalid = cur_valid;
    if (more_items) my_cell(next_item, nvq, nco, nslot, nvalid);
    f32x2 v[2][2][2];   // [i][j][channel pair]
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      f32x2 r0[4], r1[4];
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        const f32x2 m0 = {acc[px][2 * h2], acc[px][2 * h2 + 1]}, m1 = {acc[4 + px][2 * h2], acc[4 + px][2 * h2 + 1]};
        const f32x2 m2 = {acc[8 + px][2 * h2], acc[8 + px][2 * h2 + 1]}, m3 = {acc[12 + px][2 * h2], acc[12 + px][2 * h2 + 1]};
        r0[px] = w2_pk_add(w2_pk_add(m0, m1), m2);
        r1[px] = w2_pk_add(w2_pk_sub(m1, m2), m3);
      }
      v[0][0][h2] = w2_pk_add(w2_pk_add(r0[0], r0[1]), r0[2]);
      v[0][1][h2] = w2_pk_add(w2_pk_sub(r0[1], r0[2]), r0[3]);
      v[1][0][h2] = w2_pk_add(w2_pk_add(r1[0], r1[1]), r1[2]);
      v[1][1][h2] = w2_pk_add(w2_pk_sub(r1[1], r1[2]), r1[3]);
    }
    if (more_items) acc_init(nvq, nco);   // (the next item's addend: requested in front of this item's stores)
    f32x2 s0p = {0.f, 0.f}, s1p = {0.f, 0.f};
    if (cur_valid) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const f32x2 lo = v[i][j][0], hi = v[i][j][1];
          s0p = w2_pk_add(s0p, w2_pk_add(lo, hi));
          s1p = w2_pk_fma(lo, lo, s1p);
          s1p = w2_pk_fma(hi, hi, s1p);
          *reinterpret_cast<f32x4*>(y + (i64)(cur_vq + i * W + j) * Cout + cur_co) = f32x4{lo[0], lo[1], hi[0], hi[1]};
        }
      if (stats) {
        const float s0 = wave_sum(s0p[0] + s0p[1]), s1 = wave_sum(s1p[0] + s1p[1]);
        if (lane == 0) {
          float* dst = stats + (i64)cur_slot * 2;
          dst[0] = s0;
          dst[1] = s1;
        }
      }
    }
    if (!more_items) break;
    item = next_item;
    cur_vq = nvq, cur_co = nco, cur_slot = nslot, cur_valid = nvalid;
  }
  w2_dma_wait();
}

// shapes these kernels take: whole 8 x 8 x 8 tiles (the tile kernel) or, failing that, whole 4 x 4 x 4 cells (the cell kernel);
// channel blocks of 8 / 32
static bool w2_tiles(int D, int H, int W) { return !(D % W2_TS) && !(H % W2_TS) && !(W % W2_TS); }
static bool w2_cells(int D, int H, int W) { return !(D % 4) && !(H % 4) && !(W % 4); }
// cells per item of the cell kernel: four where that makes at least 192 items, else two
static int w2_cells_per_item(int N, int D, int H, int W, int Cout) {
  return (((long long)N * (D / 4) * (H / 4) * (W / 4) + 3) / 4) * (Cout / 32) >= 192 ? 4 : 2;
}
static long long w2_items(int N, int D, int H, int W, int Cout) {   // (tile | group of cells, column block) items
  if (w2_tiles(D, H, W)) return (long long)N * (D / W2_TS) * (H / W2_TS) * (W / W2_TS) * (Cout / 32);
  const int nc = w2_cells_per_item(N, D, H, W, Cout);
  return (((long long)N * (D / 4) * (H / 4) * (W / 4) + nc - 1) / nc) * (Cout / 32);
}

extern "C" int seg3d_conv3d_k3_wino2d_supported(int N, int D, int H, int W, int Cin, int Cout) {
  if (N <= 0 || D <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0) return 0;
  if (!w2_cells(D, H, W) || (Cin & 7) || (Cout & 31)) return 0;
  if (w2_items(N, D, H, W, Cout) >= (1 << 20)) return 0;
  if ((long long)N * D * H * W * (Cin > Cout ? Cin : Cout) >= (1ll << 31)) return 0;
  if ((long long)D * H * W * Cin * 4 >= (1ll << 31)) return 0;   // the input comes through one buffer resource per SAMPLE: 32-bit byte offsets
  return 1;
}

// ... and where it is the faster choice: enough items to fill the 256 CUs
extern "C" int seg3d_conv3d_k3_wino2d_preferred(int N, int D, int H, int W, int Cin, int Cout) {
  if (!seg3d_conv3d_k3_wino2d_supported(N, D, H, W, Cin, Cout)) return 0;
  return w2_items(N, D, H, W, Cout) >= 192;
}

// GroupNorm partial (sum, sumsq) slots per sample: one per (tile, column block, wave) / per (cell, column block, channel half)
extern "C" long long seg3d_conv3d_k3_wino2d_stats_count(int N, int D, int H, int W, int Cin, int Cout) {
  (void)N; (void)Cin;
  if (w2_tiles(D, H, W)) return (long long)(D / W2_TS) * (H / W2_TS) * (W / W2_TS) * (Cout / 32) * W2_NW;
  return (long long)(D / 4) * (H / 4) * (W / 4) * (Cout / 32) * 2;
}

// stream-K for the tile kernel: chunks per workgroup (0 = off).  Used when the plain walk would leave more than 6 % of the
// CU-rounds empty (432 items on 256 CUs: the 24^3 level of the train step; 864: the same at inference batches, 48^3 32 -> 32)
// and every workgroup's range is at least one item long (an item is then cut at most once)
#ifndef W2_SK_FILL
#define W2_SK_FILL 0.94
#endif
static int w2_sk_per(int N, int D, int H, int W, int Cin, int Cout) {
  if (!w2_tiles(D, H, W)) return 0;
  const long long nitems = w2_items(N, D, H, W, Cout), G = seg3d_device_cus();
  const int NSC = Cin / 4;
  if (nitems < G || NSC < 8 || nitems * NSC >= SEG3D_FDIV_MAX) return 0;
  const long long rounds = (nitems + G - 1) / G;
  if ((double)nitems / (double)(rounds * G) > W2_SK_FILL) return 0;
  // ranges of a multiple of NSC / 4 chunks: they start at four distinct chunk phases at most.  In the whole-item walk all workgroups
  // are at the same chunk of their items at any time, so an XCD's L2 holds ONE weight chunk per column block for 32 workgroups;
  // ranges of 54 chunks (4 x 24^3 128 -> 128) start at 16 phases, the 3 MB of weight images no longer stay in the 4-MB L2 beside the
  // streaming input, and FETCH_SIZE rose from 45 to 262 MB per launch -- 149 MB with 56, at the same launch time
  long long per = (nitems * NSC + G - 1) / G;
  const long long al = NSC / 4;
  per = (per + al - 1) / al * al;
  return per >= NSC ? (int)per : 0;
}
// floats of workspace seg3d_conv3d_k3_wino2d_fwd_ws wants for this shape (two 8^3 x 32 slabs per workgroup; 0: none)
extern "C" long long seg3d_conv3d_k3_wino2d_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout) {
  if (!seg3d_conv3d_k3_wino2d_supported(N, D, H, W, Cin, Cout)) return 0;
  return w2_sk_per(N, D, H, W, Cin, Cout) > 0 ? 2ll * seg3d_device_cus() * (W2_TS * W2_TS * W2_TS * 32) : 0;
}

// x [N][D][H][W][Cin], wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 48) (the F(2x2, 3x3) image), y [N][D][H][W][Cout];
// bias, addend, stats as seg3d_conv3d_k3_mfma_fwd; workspace: seg3d_conv3d_k3_wino2d_fwd_workspace_floats floats, or null (no stream-K)
extern "C" int seg3d_conv3d_k3_wino2d_fwd_ws(const float* x, const float* wp, const float* bias, const float* addend, float* y,
                                             float* stats, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                                             void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_wino2d_fwd: null pointer");
  SEG3D_REQUIRE(seg3d_conv3d_k3_wino2d_supported(N, D, H, W, Cin, Cout),
                "seg3d_conv3d_k3_wino2d_fwd: shape not supported (whole 4^3 cells, Cin %% 8 == 0, Cout %% 32 == 0)");
  const int ncog = Cout / 32;
  const int nitems = (int)w2_items(N, D, H, W, Cout);
  const int sk_per = workspace ? w2_sk_per(N, D, H, W, Cin, Cout) : 0;
  dim3 grid(sk_per > 0 ? (unsigned)seg3d_device_cus() : seg3d_persistent_grid(nitems), 1, 1);
  if (w2_tiles(D, H, W)) {
    static Seg3dOncePerDevice configured[4];
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<false, false>), configured[0], "conv3d_k3_wino2d")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<true, false>), configured[1], "conv3d_k3_wino2d")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<false, true>), configured[2], "conv3d_k3_wino2d")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_kernel<true, true>), configured[3], "conv3d_k3_wino2d")) return rc;
    const int ntz = D / W2_TS, nty = H / W2_TS, ntx = W / W2_TS;
#define W2_LAUNCH(B_, A_)                                                                                                       \
  hipLaunchKernelGGL((conv3d_k3_wino2d_kernel<B_, A_>), grid, dim3(64 * W2_NW), (size_t)W2_LDS_FLOATS * 4, (hipStream_t)stream, x, \
                     wp, bias, y, stats, N, D, H, W, Cin, Cout, ntz, nty, ntx, ncog, nitems, addend, sk_per, workspace)
    if (bias) {
      if (addend) W2_LAUNCH(true, true); else W2_LAUNCH(true, false);
    } else {
      if (addend) W2_LAUNCH(false, true); else W2_LAUNCH(false, false);
    }
#undef W2_LAUNCH
    if (sk_per > 0) {
      SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_fwd");
      hipLaunchKernelGGL(conv3d_k3_wino2d_sk_finish_kernel, dim3(grid.x - 1), dim3(512), 0, (hipStream_t)stream, workspace, y, stats, D,
                         H, W, Cout, ntz, nty, ntx, ncog, nitems, Cin / 4, sk_per);
    }
  } else {
    static Seg3dOncePerDevice configured[8];
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<false, false, 4>), configured[0], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<true, false, 4>), configured[1], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<false, true, 4>), configured[2], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<true, true, 4>), configured[3], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<false, false, 2>), configured[4], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<true, false, 2>), configured[5], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<false, true, 2>), configured[6], "conv3d_k3_wino2d_c4")) return rc;
    if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wino2d_c4_kernel<true, true, 2>), configured[7], "conv3d_k3_wino2d_c4")) return rc;
    const int ncz = D / 4, ncy = H / 4, ncx = W / 4, ncells = N * ncz * ncy * ncx;
    const int nc = w2_cells_per_item(N, D, H, W, Cout);
#define W2_LAUNCH(B_, A_, NC_)                                                                                                       \
  hipLaunchKernelGGL((conv3d_k3_wino2d_c4_kernel<B_, A_, NC_>), grid, dim3(128 * NC_), (size_t)C4_LDS_FLOATS * 4, (hipStream_t)stream, \
                     x, wp, bias, y, stats, N, D, H, W, Cin, Cout, ncz, ncy, ncx, ncells, ncog, nitems, addend)
    if (nc == 4) {
      if (bias) {
        if (addend) W2_LAUNCH(true, true, 4); else W2_LAUNCH(true, false, 4);
      } else {
        if (addend) W2_LAUNCH(false, true, 4); else W2_LAUNCH(false, false, 4);
      }
    } else {
      if (bias) {
        if (addend) W2_LAUNCH(true, true, 2); else W2_LAUNCH(true, false, 2);
      } else {
        if (addend) W2_LAUNCH(false, true, 2); else W2_LAUNCH(false, false, 2);
      }
    }
#undef W2_LAUNCH
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_wino2d_fwd");
  return SEG3D_OK;
}

// the same without a workspace (every item whole in one workgroup)
extern "C" int seg3d_conv3d_k3_wino2d_fwd(const float* x, const float* wp, const float* bias, const float* addend, float* y,
                                          float* stats, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  return seg3d_conv3d_k3_wino2d_fwd_ws(x, wp, bias, addend, y, stats, nullptr, N, D, H, W, Cin, Cout, stream);
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
// between two tiles all 256 threads transform the NEW planes of RAW x into T[p][plane slot][quad][32 ci] (a ring of six plane slots
// along z: a workgroup walks its tiles z fastest, planes 4, 5 of a tile are planes 0, 1 of the next; two barriers per tile); E is formed from the raw
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
  // GX + g' is dy piece min(wave + 4 g', YPC - 1); lane -> voxel 8 p + (lane >> 3), channels 4 (lane & 7)..+3.
  // The pieces come through buffer resources of the SAMPLE (w2_bufdma16_if): pconst = the lane's byte offset relative to the
  // tile origin (negative for halo voxels in front of it), OOB for channels past the tensor's; z padding = the range check
  // (plane -1 wraps, plane D is past the end), (y, x) padding = pyx, the table below refreshed once per tile column.
  const int lv = lane >> 3, lq = lane & 7;
  const unsigned OOB = 0x80000000u;                    // stays out of range after any tile offset (sample bytes < 2^31)
  unsigned pconst[G2_NG], pyx[G2_GX];
  int pflag[G2_GX];
#pragma unroll
  for (int g = 0; g < G2_NG; ++g) {
    if (g < G2_GX) {
      const int p = wave + 4 * g < G2_XPC ? wave + 4 * g : G2_XPC - 1;
      const int v = p * 8 + lv;
      const int t = fdiv(v, 1.0f / 6.0f);
      const int hx = v - t * 6;
      const int hz = fdiv(t, 1.0f / 6.0f);
      const int hy = t - hz * 6;
      pconst[g] = ci0 + 4 * lq < Cin ? (unsigned)(((((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 4 * lq) * 4) : OOB;
      pflag[g] = (hy == 0 ? 4 : 0) | (hy == 5 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == 5 ? 32 : 0);
      pyx[g] = pconst[g];
    } else {
      const int p = wave + 4 * (g - G2_GX) < G2_YPC ? wave + 4 * (g - G2_GX) : G2_YPC - 1;
      const int v = p * 8 + lv;
      const int co = co0 + 4 * lq;
      pconst[g] = co < Cout ? (unsigned)(((((v >> 4) * H + ((v >> 2) & 3)) * W + (v & 3)) * Cout + co) * 4) : OOB;
    }
  }
  const unsigned xbytes = (unsigned)D * H * W * Cin * 4u, ybytes = (unsigned)D * H * W * Cout * 4u;   // one sample
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds);
  int tn = 0, tz0 = 0, ty0 = 0, tx0 = 0;  // origin of the tile being fetched
  // tiles are numbered z FASTEST and a workgroup walks a contiguous range of them: consecutive tiles are neighbours in z (until
  // the column ends), and two of the six T planes of a tile are the last two of its predecessor (see the plane ring below)
  // the walk is contiguous, so the coordinates of the current tile and of the two fetched ahead are carried as wave-uniform
  // counters (z, x, y, n in tiles) and stepped with scalar instructions; only the first tile is decoded by division
  struct TileAt { int z, x, y, n; };
  auto tile_at = [&](int tile) {
    int b = tile;
    int q = fdiv(b, rNTZ);
    TileAt t;
    t.z = __builtin_amdgcn_readfirstlane(b - q * ntz);
    b = q;
    q = fdiv(b, rNTX);
    t.x = __builtin_amdgcn_readfirstlane(b - q * ntx);
    b = q;
    q = fdiv(b, rNTY);
    t.y = __builtin_amdgcn_readfirstlane(b - q * nty);
    t.n = __builtin_amdgcn_readfirstlane(q);
    return t;
  };
  auto step = [&](TileAt t) {
    if (++t.z == ntz) {
      t.z = 0;
      if (++t.x == ntx) {
        t.x = 0;
        if (++t.y == nty) t.y = 0, ++t.n;
      }
    }
    return t;
  };
  auto use_tile = [&](const TileAt& t) { tn = t.n, tz0 = t.z * 4, ty0 = t.y * 4, tx0 = t.x * 4; };
  // T plane RING: halo plane k (0..5) of the tile at z index tiz lives in slot (4 tiz + k) % 6 of T[p][slot][quad][32 ci]; planes
  // 4, 5 of a tile are planes 0, 1 of the next tile of the column and are not transformed again
  auto slot_of = [](int tiz, int k) { return (4 * tiz + k) % 6; };
  // (y, x) padding of the x pieces for the column of the tile being fetched: refreshed only when that tile starts a column
  auto column_table = [&]() {
    const int fyx = (ty0 == 0 ? 4 : 0) | (ty0 + 4 >= H ? 8 : 0) | (tx0 == 0 ? 16 : 0) | (tx0 + 4 >= W ? 32 : 0);
#pragma unroll
    for (int g = 0; g < G2_GX; ++g) pyx[g] = (pflag[g] & fyx) ? OOB : pconst[g];
  };
  // piece g of the tile at (tn, tz0, ty0, tx0) into the buffers at the float offsets xdst / ydst of the carve-up.  A tile that
  // CONTINUES a column needs no halo planes 0, 1 (they are planes 4, 5 of its predecessor, already in the T ring): its x pieces
  // 0 .. 8 (= voxels 0 .. 71 = those two planes) are not fetched at all
  auto issue_piece = [&](int g, unsigned xdst, unsigned ydst, w2_srd xsrd, w2_srd ysrd, unsigned xoff, unsigned yoff, int column_start) {
    if (g < G2_GX) {
      const int piece = wave + 4 * g < G2_XPC ? wave + 4 * g : G2_XPC - 1;
      const int pred = (4 * g + 3 < 9) ? column_start : ((4 * g < 9) ? (column_start | (piece >= 9)) : 1);   // compile-time for g >= 3
      w2_bufdma16_if(pred, pyx[g] + xoff, xsrd, lds0 + (xdst + piece * 256) * 4);
    } else {
      const int piece = wave + 4 * (g - G2_GX) < G2_YPC ? wave + 4 * (g - G2_GX) : G2_YPC - 1;
      w2_bufdma16_if(1, pconst[g] + yoff, ysrd, lds0 + (ydst + piece * 256) * 4);
    }
  };
  // RAW x -> T: V = B^T d B per (plane, quad), 16 points, as HALF tasks (plane, quad, channel pair): 256 of them for the four new
  // planes of a tile that continues a column -- exactly one per thread, 32 packed adds -- and 384 for the six planes of a tile
  // that starts one (threads 0 .. 127 take a second task).  (Rounds 2-3: 192 four-channel tasks of 64 packed adds on three of
  // the four waves, all six planes for every tile.)
  auto transform_task = [&](const float* rx, int t, int kbase, int tiz) {
    const int kpl = __builtin_amdgcn_readfirstlane(kbase + (t >> 6));       // halo plane (wave-uniform)
    const int t_q = (t >> 4) & 3, t_c2 = t & 15;
    const float* src = rx + ((kpl * 6 + 2 * (t_q >> 1)) * 6 + 2 * (t_q & 1)) * 32 + 2 * t_c2;
    float* dst = timg + (slot_of(tiz, kpl) * 4 + t_q) * 32 + 2 * t_c2;
    f32x2 dx[4][4];   // [row][px]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float* sp = src + r * (6 * 32);
      const f32x2 d0 = *reinterpret_cast<const f32x2*>(sp), d1 = *reinterpret_cast<const f32x2*>(sp + 32);
      const f32x2 d2 = *reinterpret_cast<const f32x2*>(sp + 64), d3 = *reinterpret_cast<const f32x2*>(sp + 96);
      dx[r][0] = w2_pk_sub(d0, d2);
      dx[r][1] = w2_pk_add(d1, d2);
      dx[r][2] = w2_pk_sub(d2, d1);
      dx[r][3] = w2_pk_sub(d1, d3);
    }
#pragma unroll
    for (int px = 0; px < 4; ++px) {
      *reinterpret_cast<f32x2*>(dst + (0 * 4 + px) * (24 * 32)) = w2_pk_sub(dx[0][px], dx[2][px]);
      *reinterpret_cast<f32x2*>(dst + (1 * 4 + px) * (24 * 32)) = w2_pk_add(dx[1][px], dx[2][px]);
      *reinterpret_cast<f32x2*>(dst + (2 * 4 + px) * (24 * 32)) = w2_pk_sub(dx[2][px], dx[1][px]);
      *reinterpret_cast<f32x2*>(dst + (3 * 4 + px) * (24 * 32)) = w2_pk_sub(dx[1][px], dx[3][px]);
    }
  };
  auto transform = [&](const float* rx, int tiz, bool column_start) {
    if (column_start) {   // wave-uniform
      transform_task(rx, tid, 0, tiz);                    // planes 0 .. 3
      if (tid < 128) transform_task(rx, tid, 4, tiz);     // planes 4, 5
    } else {
      transform_task(rx, tid, 2, tiz);                    // planes 2 .. 5; 0, 1 are the previous tile's 4, 5
    }
  };

  // tile walk: a contiguous range per workgroup (neighbouring ranges on one XCD when the slab count allows)
  const int per_wg = (ntiles + slabs - 1) / slabs;
  const int ord = (slabs & 7) == 0 ? (slab & 7) * (slabs >> 3) + (slab >> 3) : slab;
  int tile = ord * per_wg;
  const int tstride = 1;
  const int tlimit = tile + per_wg < ntiles ? tile + per_wg : ntiles;
  auto fetch = [&](const TileAt& t, unsigned xdst, unsigned ydst) {   // all pieces of tile t at once (prologue)
    use_tile(t);
    column_table();
    const unsigned origin = (unsigned)((tz0 * H + ty0) * W + tx0);   // voxels inside the sample
    const w2_srd xsrd = w2_make_srd(x + (i64)tn * D * H * W * Cin, xbytes), ysrd = w2_make_srd(dy + (i64)tn * D * H * W * Cout, ybytes);
#pragma unroll
    for (int g = 0; g < G2_NG; ++g) issue_piece(g, xdst, ydst, xsrd, ysrd, origin * Cin * 4u, origin * Cout * 4u, 1);
  };
  int xi = 0, yi = 0;   // buffers of the current tile: RAW x (already transformed) xi, dy yi
  TileAt cur = tile_at(tile < ntiles ? tile : 0), nx1 = cur, nx2 = cur;   // this tile, the next one, the one after
  if (tile < tlimit) {
    nx1 = tile + 1 < tlimit ? step(cur) : cur;        // past the end of the range: this tile once more (idle buffers)
    nx2 = tile + 2 < tlimit ? step(nx1) : cur;
    fetch(cur, 0, 2 * G2_XS + G2_TS);
    fetch(nx1, G2_XS, 2 * G2_XS + G2_TS + G2_YS);
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G2_NG) : "memory");   // the first tile landed, the second may be in flight
    __syncthreads();
    transform(rawx, cur.z, true);
    __syncthreads();
  }
  for (; tile < tlimit; tile += tstride) {
    const float* ycur = rawy + yi * G2_YS;
    const int y2 = yi == 0 ? 2 : yi - 1;                 // (yi + 2) % 3
    const unsigned ynxt2 = 2 * G2_XS + G2_TS + y2 * G2_YS;   // dy buffer of the tile after next (float offset in the carve-up)
    const unsigned xnxt2 = xi * G2_XS;                       // RAW x of the current tile is dead (transformed): reuse it
    const bool more = tile + tstride < tlimit;
    // the tile after next -- past the end: this tile once more, into idle buffers (keeps the loop uniform)
    use_tile(nx2);
    const int f_start = nx2.z == 0;                          // it starts a column: new (y, x) padding, all six halo planes
    if (f_start) column_table();
    const unsigned origin = (unsigned)((tz0 * H + ty0) * W + tx0);
    const w2_srd xsrd = w2_make_srd(x + (i64)tn * D * H * W * Cin, xbytes), ysrd = w2_make_srd(dy + (i64)tn * D * H * W * Cout, ybytes);
    const unsigned fxoff = origin * Cin * 4u, fyoff = origin * Cout * 4u;
    // K step k: quads 2k (lane half 0) and 2k + 1 (half 1) = (z, qy) = (k >> 1, k & 1), qx = lane half
    const int tiz = cur.z;
    const float* ta = timg + (wave * 4) * (24 * 32) + lane;        // + (px * 24 + slot(z + kz) * 4 + 2 qy) * 32
    const float* tk[6];                                            // per halo plane: its ring slot
#pragma unroll
    for (int kp = 0; kp < 6; ++kp) tk[kp] = ta + slot_of(tiz, kp) * (4 * 32);
    const float* yb = ycur + lh * 64 + li;                         // + ((z * 4 + 2 qy) * 4) * 32; j: + 32, i: + 128
    auto lda = [&](int k, int kz, int px) { return tk[(k >> 1) + kz][(px * 24 + 2 * (k & 1)) * 32]; };
    auto yoff = [](int k) { return (((k >> 1) * 4 + 2 * (k & 1)) * 4) * 32; };
    float a1[12], g00, g01, g10, g11;
#pragma unroll
    for (int j = 0; j < 12; ++j) a1[j] = lda(0, j >> 2, j & 3);
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
        for (int j = 0; j < 12; ++j) a1[j] = lda(k + 1, j >> 2, j & 3);
        g00 = yb[yoff(k + 1) + g0off];
        g01 = yb[yoff(k + 1) + g0off + 32];
        g10 = yb[yoff(k + 1) + 128];
        g11 = yb[yoff(k + 1) + 128 + 32];
      }
#pragma unroll
      for (int g = 0; g < G2_NG; ++g)
#ifndef G2_EXP_NODMA
        if (g % G2_KS == k) issue_piece(g, xnxt2, ynxt2, xsrd, ysrd, fxoff, fyoff, f_start);
#endif
      __builtin_amdgcn_sched_barrier(0);   // keeps hipcc from sinking the reads above down to their first use
#pragma unroll
      for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], e[j & 3], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    // the NEXT tile landed (fetched one tile ago): all but the pieces of the latest fetch -- at least G2_NG - 3 of them (a tile
    // that continues a column skips two or three x pieces per wave) -- so the wait asks for that many outstanding at most
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G2_NG - 3) : "memory");
    __syncthreads();     // everyone done with T
    xi ^= 1;
    yi = yi == 2 ? 0 : yi + 1;
    if (more) {
#ifndef G2_EXP_NOTR    // (diagnostic builds: what do the exposed transform and its barrier cost?)
      {                               // the next tile's RAW x -> T
        transform(rawx + xi * G2_XS, nx1.z, nx1.z == 0);
      }
#endif
#ifndef G2_EXP_NOBAR2
      __syncthreads();
#endif
    }
    cur = nx1;
    nx1 = nx2;
    nx2 = tile + 3 < tlimit ? step(nx2) : nx2;   // (tile + 3 = the tile two ahead of the NEXT iteration's tile)
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

#ifndef G2_INLOOP
#define G2_INLOOP 0   // 1: build and use conv3d_k3_wgrad_wino2d_il_kernel (measured 4 % SLOWER: kept as a reproducible negative result)
#endif
#if G2_INLOOP
// The same kernel with the RAW x -> T transform INSIDE the MFMA loop (round 4, conv3d_k3_wgrad_wino2d_il_kernel).  Above, the
// transform of a tile's four new halo planes is a phase of its own between two barriers (14 % of the kernel with the matrix pipe
// idle); here the planes of the NEXT step are transformed during the current step's eight K steps -- one task per thread: four
// row stages at K steps 0..3, four column stages at 4..7 -- into ring slots the current step does not read, and a step ends with
// ONE barrier.  What makes it fit:
//   * T is a ring of TEN plane slots (80 KB): a tile reads its six planes at slots tb .. tb + 5, the next step's planes 2..5 are
//     written to tb + 6 .. tb + 9, and tb advances by four per step;
//   * a RAW entry holds only the planes 2..5 of its step (18 KB): three entries (54 KB, what the two 27-KB tiles took) let the
//     fetch run THREE steps ahead, so the step being transformed landed a whole step ago (a tile is 2.6 us: too short for a fetch
//     issued in the same tile); dy stays two ahead in three buffers.  LDS: 54 + 80 + 24 = 158 KB;
//   * a tile that STARTS a column (or a workgroup's range) has no predecessor whose planes 4, 5 are its planes 0, 1: a PHANTOM
//     step in front of it -- the geometry of the tile one further down in z (planes out of the volume are zeros by the
//     resource's range check), transform and fetches as usual, no MFMAs -- provides them, and the pipeline has no special case.
// One phantom step per column costs what the exposed transform cost per TILE; columns of 3 tiles (the 12^3 level) are a draw.
// RESULT (tools/bench_k3.py, same box, passes test_conv3d_k3_winograd_wgrad and the random-shape test): 4 x 96^3 32->32 0.832 ->
// 0.868 ms, 48^3 64->64 0.424 -> 0.445, 24^3 128->128 0.224 -> 0.238: 4-6 % SLOWER.  With one wave per SIMD the transform's vector
// instructions are not hidden by being placed between MFMAs -- fp32 VALU and fp32 MFMA are one datapath, so its 32 packed
// adds cost the same cycles inside the loop as outside -- and what the in-loop form saves (one barrier and the LDS round trip of
// the exposed phase, ~ 400 of 7 250 cycles per tile) is less than what it adds (a uniform branch around every K step's MFMAs,
// the phantom steps, fetch descriptors three steps ahead).  Not compiled by default (-DG2_INLOOP=1 builds and uses it).
#define G3_XV 144                                 // voxels of a RAW entry: halo planes 2 .. 5 (4 x 6 x 6)
#define G3_XS (G3_XV * 32)
#define G3_NSLOT 10
#define G3_PS (G3_NSLOT * 4 * 32)                 // floats of one Winograd point of T: [slot][quad][32 ci]
#define G3_TS (16 * G3_PS)
#define G3_YS (64 * 32)
#define G3_LDS_FLOATS (3 * G3_XS + G3_TS + 3 * G3_YS)   // 40 448 floats = 161 792 bytes
#define G3_XPC 18                                 // 1-KiB DMA pieces of a RAW entry
#define G3_GX 5                                   // x piece groups per wave (pieces wave + 4 g, clamped: duplicates are harmless)
#define G3_GY 2
#define G3_NG (G3_GX + G3_GY)                     // LDS-DMA instructions per wave and step: the end-of-step wait leaves exactly these

__global__ __launch_bounds__(256, 1) void conv3d_k3_wgrad_wino2d_il_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                            float* __restrict__ part, int N, int D, int H, int W,
                                                                            int Cin, int Cout, int ntz, int nty, int ntx, int ntiles,
                                                                            int slabs, int COB32) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* rawx = lds;                        // [3][144][32]
  float* timg = lds + 3 * G3_XS;            // [16 p][10 slots][4 quads][32 ci]
  float* rawy = timg + G3_TS;               // [3][64][32]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = point row py
  const int li = lane & 31, lh = lane >> 5;
  const int slab = blockIdx.x % slabs;
  const int pg = blockIdx.x / slabs;                   // (ci block, co block)
  const int cib = pg / COB32, cob = pg % COB32;
  const int ci0 = cib * 32, co0 = cob * 32;
  const float rNTX = 1.0f / (float)ntx, rNTY = 1.0f / (float)nty, rNTZ = 1.0f / (float)ntz;
  auto fdiv = [](int v, float r) { return (int)(((float)v + 0.5f) * r); };
  const float e1 = wave == 1 ? 1.f : (wave == 2 ? -1.f : 0.f);   // row combination of the dy quad for py = wave (see above)
  const int g0off = wave == 3 ? 4 * 32 : 0;

  f32x16 acc[12];   // [kz][px]
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

  // DMA pieces of this wave: x piece min(wave + 4 g, 17) of the entry (voxel 8 p + (lane >> 3) of the 144, channels 4 (lane & 7)..),
  // dy piece min(wave + 4 g', 7); offsets relative to the step's tile origin, (y, x) padding from the per-column table
  const int lv = lane >> 3, lq = lane & 7;
  const unsigned OOB = 0x80000000u;
  unsigned pconst[G3_NG], pyx[G3_GX];
  int pflag[G3_GX];
#pragma unroll
  for (int g = 0; g < G3_NG; ++g) {
    if (g < G3_GX) {
      const int p = wave + 4 * g < G3_XPC ? wave + 4 * g : G3_XPC - 1;
      const int v = p * 8 + lv;
      const int pl = fdiv(v, 1.0f / 36.0f);              // plane of the entry: halo plane 2 + pl
      const int r = v - pl * 36;
      const int hy = fdiv(r, 1.0f / 6.0f);
      const int hx = r - hy * 6;
      pconst[g] = ci0 + 4 * lq < Cin ? (unsigned)(((((pl + 1) * H + (hy - 1)) * W + (hx - 1)) * Cin + ci0 + 4 * lq) * 4) : OOB;
      pflag[g] = (hy == 0 ? 4 : 0) | (hy == 5 ? 8 : 0) | (hx == 0 ? 16 : 0) | (hx == 5 ? 32 : 0);
      pyx[g] = pconst[g];
    } else {
      const int p = wave + 4 * (g - G3_GX) < 8 ? wave + 4 * (g - G3_GX) : 7;
      const int v = p * 8 + lv;
      const int co = co0 + 4 * lq;
      pconst[g] = co < Cout ? (unsigned)(((((v >> 4) * H + ((v >> 2) & 3)) * W + (v & 3)) * Cout + co) * 4) : OOB;
    }
  }
  const unsigned xbytes = (unsigned)D * H * W * Cin * 4u, ybytes = (unsigned)D * H * W * Cout * 4u;   // one sample
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(w2_lds_float*)lds);

  // steps: wave-uniform descriptors.  ph = 1: the phantom step in front of tile (z, x, y, n); ok = 0: past the end of the range
  struct StepAt { int z, x, y, n, tile, ph, ok; };
  const int per_wg = (ntiles + slabs - 1) / slabs;
  const int ord = (slabs & 7) == 0 ? (slab & 7) * (slabs >> 3) + (slab >> 3) : slab;
  const int tile0 = ord * per_wg;
  const int tlimit = tile0 + per_wg < ntiles ? tile0 + per_wg : ntiles;
  auto first_step = [&]() {
    StepAt t;
    int b = tile0 < ntiles ? tile0 : 0;
    int q = fdiv(b, rNTZ);
    t.z = __builtin_amdgcn_readfirstlane(b - q * ntz);
    b = q;
    q = fdiv(b, rNTX);
    t.x = __builtin_amdgcn_readfirstlane(b - q * ntx);
    b = q;
    q = fdiv(b, rNTY);
    t.y = __builtin_amdgcn_readfirstlane(b - q * nty);
    t.n = __builtin_amdgcn_readfirstlane(q);
    t.tile = tile0, t.ph = 1, t.ok = tile0 < tlimit ? 1 : 0;
    return t;
  };
  auto next_step = [&](StepAt t) {
    if (!t.ok) return t;
    if (t.ph) {
      t.ph = 0;
      return t;
    }
    if (t.tile + 1 >= tlimit) {
      t.ok = 0;            // (keeps the geometry of the last tile: fetches and transforms past the end go to idle buffers / slots)
      return t;
    }
    ++t.tile;
    if (++t.z == ntz) {
      t.z = 0;
      if (++t.x == ntx) {
        t.x = 0;
        if (++t.y == nty) t.y = 0, ++t.n;
      }
    }
    t.ph = t.z == 0 ? 1 : 0;
    return t;
  };
  // fetch of step t into RAW entry xe / dy buffer ye: piece group g of this wave.  The (y, x) padding table belongs to the column
  // of the step being fetched (fetches walk the columns in order: refreshed when a fetched step is a phantom, i.e. starts one)
  auto column_table = [&](const StepAt& t) {
    const int ty0 = t.y * 4, tx0 = t.x * 4;
    const int fyx = (ty0 == 0 ? 4 : 0) | (ty0 + 4 >= H ? 8 : 0) | (tx0 == 0 ? 16 : 0) | (tx0 + 4 >= W ? 32 : 0);
#pragma unroll
    for (int g = 0; g < G3_GX; ++g) pyx[g] = (pflag[g] & fyx) ? OOB : pconst[g];
  };
  struct FetchAt { w2_srd xsrd, ysrd; unsigned xoff, yoff, xdst, ydst; };
  auto fetch_of = [&](const StepAt& t, int xe, int ye) {
    FetchAt f;
    const int tz0 = (t.z - t.ph) * 4;                    // a phantom step has the geometry of the tile one further down (may be -4)
    const int origin = (tz0 * H + t.y * 4) * W + t.x * 4;
    f.xsrd = w2_make_srd(x + (i64)t.n * D * H * W * Cin, xbytes);
    f.ysrd = w2_make_srd(dy + (i64)t.n * D * H * W * Cout, ybytes);
    f.xoff = (unsigned)origin * (unsigned)Cin * 4u;      // (mod 2^32: planes in front of the sample wrap out of the resource's range)
    f.yoff = (unsigned)origin * (unsigned)Cout * 4u;
    f.xdst = xe * G3_XS;
    f.ydst = 3 * G3_XS + G3_TS + ye * G3_YS;
    return f;
  };
  auto issue_piece = [&](int g, const FetchAt& f) {
    if (g < G3_GX) {
      const int piece = wave + 4 * g < G3_XPC ? wave + 4 * g : G3_XPC - 1;
      w2_bufdma16_if(1, pyx[g] + f.xoff, f.xsrd, lds0 + (f.xdst + piece * 256) * 4);
    } else {
      const int piece = wave + 4 * (g - G3_GX) < 8 ? wave + 4 * (g - G3_GX) : 7;
      w2_bufdma16_if(1, pconst[g] + f.yoff, f.ysrd, lds0 + (f.ydst + piece * 256) * 4);
    }
  };
  // transform task of this thread: (plane 2 + (tid >> 6), quad (tid >> 4) & 3, channel pair tid & 15) of a RAW entry
  const int t_pl = __builtin_amdgcn_readfirstlane(tid >> 6), t_q = (tid >> 4) & 3, t_c2 = tid & 15;
  const int t_src = ((t_pl * 6 + 2 * (t_q >> 1)) * 6 + 2 * (t_q & 1)) * 32 + 2 * t_c2;
  const int t_dst = t_q * 32 + 2 * t_c2;     // + (slot * 4) * 32 + point * G3_PS
  f32x2 dx[4][4];   // [row][px]
  f32x2 dr[4];      // the row read at this K step, combined behind its MFMAs
  auto tr_read = [&](const float* rx, int r) {
    const float* sp = rx + t_src + r * (6 * 32);
    dr[0] = *reinterpret_cast<const f32x2*>(sp), dr[1] = *reinterpret_cast<const f32x2*>(sp + 32);
    dr[2] = *reinterpret_cast<const f32x2*>(sp + 64), dr[3] = *reinterpret_cast<const f32x2*>(sp + 96);
  };
  auto tr_math = [&](int r) {
    dx[r][0] = w2_pk_sub(dr[0], dr[2]);
    dx[r][1] = w2_pk_add(dr[1], dr[2]);
    dx[r][2] = w2_pk_sub(dr[2], dr[1]);
    dx[r][3] = w2_pk_sub(dr[1], dr[3]);
  };
  auto tr_row = [&](const float* rx, int r) { tr_read(rx, r), tr_math(r); };
  auto tr_col = [&](float* dst, int px) {   // dst = timg + slot * 128 + t_dst
    *reinterpret_cast<f32x2*>(dst + (0 * 4 + px) * G3_PS) = w2_pk_sub(dx[0][px], dx[2][px]);
    *reinterpret_cast<f32x2*>(dst + (1 * 4 + px) * G3_PS) = w2_pk_add(dx[1][px], dx[2][px]);
    *reinterpret_cast<f32x2*>(dst + (2 * 4 + px) * G3_PS) = w2_pk_sub(dx[2][px], dx[1][px]);
    *reinterpret_cast<f32x2*>(dst + (3 * 4 + px) * G3_PS) = w2_pk_sub(dx[1][px], dx[3][px]);
  };
  auto slot_mod = [](int v) { return v >= G3_NSLOT ? v - G3_NSLOT : v; };

  StepAt cur = first_step();   // (an empty range -- more slabs than tile triples -- still writes its zero slab below)
  StepAt nx1 = next_step(cur), nx2 = next_step(nx1), nx3 = next_step(nx2);
  int tb = 0;          // ring slot of the current step's halo plane 0
  int xe = 0, ye = 0;  // RAW entry / dy buffer of the current step
  if (cur.ok) {   // prologue: steps 0, 1, 2 (x) and 0, 1 (dy) at once; step 0's planes 2 .. 5 transformed here, exposed
    column_table(cur);
    const FetchAt f0 = fetch_of(cur, 0, 0);
#pragma unroll
    for (int g = 0; g < G3_NG; ++g) issue_piece(g, f0);
    if (nx1.ph) column_table(nx1);
    const FetchAt f1 = fetch_of(nx1, 1, 1);
#pragma unroll
    for (int g = 0; g < G3_NG; ++g) issue_piece(g, f1);
    if (nx2.ph) column_table(nx2);
    const FetchAt f2 = fetch_of(nx2, 2, 2);
#pragma unroll
    for (int g = 0; g < G3_GX; ++g) issue_piece(g, f2);
    w2_dma_wait();
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) tr_row(rawx, r);
    float* d0 = timg + slot_mod(tb + 2 + t_pl) * 128 + t_dst;
#pragma unroll
    for (int px = 0; px < 4; ++px) tr_col(d0, px);
    __syncthreads();
  }
  for (; cur.ok;) {
    // this step: T planes at slots tb + kp, dy in buffer ye; next step's RAW entry xe + 1 -> T slots tb + 6 .. 9;
    // fetched: x of the step three ahead -> entry xe (dead), dy of the step two ahead -> buffer ye + 2
    const int xe1 = xe == 2 ? 0 : xe + 1, ye2 = ye == 0 ? 2 : ye - 1;
    if (nx3.ph) column_table(nx3);
    FetchAt f = fetch_of(nx3, xe, ye2);
    {
      const FetchAt fy = fetch_of(nx2, xe, ye2);   // (dy runs two ahead)
      f.ysrd = fy.ysrd, f.yoff = fy.yoff;
    }
    const float* rx1 = rawx + xe1 * G3_XS;
    float* tdst = timg + slot_mod(slot_mod(tb + 6) + t_pl) * 128 + t_dst;
    const float* ycur = rawy + ye * G3_YS;
    const float* ta = timg + (wave * 4) * G3_PS + lane;          // + px * G3_PS + (slot * 4 + 2 qy) * 32
    const float* tk[6];
#pragma unroll
    for (int kp = 0; kp < 6; ++kp) tk[kp] = ta + slot_mod(tb + kp) * (4 * 32);
    const float* yb = ycur + lh * 64 + li;
    auto lda = [&](int k, int kz, int px) { return tk[(k >> 1) + kz][px * G3_PS + (2 * (k & 1)) * 32]; };
    auto yoff = [](int k) { return (((k >> 1) * 4 + 2 * (k & 1)) * 4) * 32; };
    const int real = cur.ph ? 0 : 1;
    float a1[12], g00, g01, g10, g11;
#pragma unroll
    for (int j = 0; j < 12; ++j) a1[j] = lda(0, j >> 2, j & 3);
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
        for (int j = 0; j < 12; ++j) a1[j] = lda(k + 1, j >> 2, j & 3);
        g00 = yb[yoff(k + 1) + g0off];
        g01 = yb[yoff(k + 1) + g0off + 32];
        g10 = yb[yoff(k + 1) + 128];
        g11 = yb[yoff(k + 1) + 128 + 32];
      }
      if (k < G3_NG) issue_piece(k, f);
      if (k < 4) tr_read(rx1, k);        // (its four reads fly while this K step's MFMAs run; combined behind them)
      __builtin_amdgcn_sched_barrier(0);
      if (real) {
#pragma unroll
        for (int j = 0; j < 12; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], e[j & 3], acc[j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (k < 4) tr_math(k); else tr_col(tdst, k - 4);
      __builtin_amdgcn_sched_barrier(0);
    }
    // everything but this step's own G3_NG fetches has landed: the next step's dy, the RAW entry of the step after it
    asm volatile("s_waitcnt vmcnt(%0)" : : "n"(G3_NG) : "memory");
    __syncthreads();     // T of the next step complete; everyone done with this step's T planes, dy buffer and the dead RAW entry
    tb = slot_mod(tb + 4);
    xe = xe1;
    ye = ye == 2 ? 0 : ye + 1;
    cur = nx1;
    nx1 = nx2;
    nx2 = nx3;
    nx3 = next_step(nx3);
  }
  w2_dma_wait();   // nothing in flight when the workgroup's LDS is released

  float* dst = part + ((i64)slab * (COB32 * ((Cin + 31) / 32)) + cib * COB32 + cob) * 48 * 1024;
#pragma unroll
  for (int j = 0; j < 12; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      dst[((j >> 2) * 16 + wave * 4 + (j & 3)) * 1024 + w2_mfma_row(r, lh) * 32 + li] = acc[j][r];
}

#endif   // G2_INLOOP

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
  if ((long long)D * H * W * (Cin > Cout ? Cin : Cout) * 4 >= (1ll << 31)) return 0;   // one SAMPLE per buffer resource: 32-bit byte offsets
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
#if G2_INLOOP
  static Seg3dOncePerDevice configured_il;
  if (int rc = seg3d_allow_full_lds(reinterpret_cast<const void*>(&conv3d_k3_wgrad_wino2d_il_kernel), configured_il, "conv3d_k3_wgrad_wino2d_il")) return rc;
#endif
  const int slabs = g2_slabs(N, D, H, W, Cin, Cout);
  const int CIB32 = (Cin + 31) / 32, COB32 = (Cout + 31) / 32, npairs = CIB32 * COB32;
  const int ntz = D / 4, nty = H / 4, ntx = W / 4;
  const int ntiles = N * ntz * nty * ntx;
  hipStream_t s = (hipStream_t)stream;
#if G2_INLOOP
  if (ntz >= 4)   // columns of four tiles and more (below, a phantom step per column costs as much as the exposed transforms)
    hipLaunchKernelGGL(conv3d_k3_wgrad_wino2d_il_kernel, dim3((unsigned)(slabs * npairs)), dim3(256), (size_t)G3_LDS_FLOATS * 4, s, x,
                       dy, workspace, N, D, H, W, Cin, Cout, ntz, nty, ntx, ntiles, slabs, COB32);
  else
#endif
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
