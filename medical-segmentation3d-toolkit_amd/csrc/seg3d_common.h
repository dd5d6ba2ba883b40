// seg3d_common.h -- shared helpers for the gfx950 (MI355X / CDNA4) segmentation kernels.
// All kernels in this library work on NDHWC fp32 activations ("voxel rows" of C floats) unless a
// function says otherwise; see DESIGN.md for the HBM layout of every buffer.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#define SEG3D_OK 0
#define SEG3D_ERR_INVALID (-1)
#define SEG3D_ERR_LAUNCH (-2)
#define SEG3D_ERR_UNSUPPORTED (-3)

// thread-local error string, read back through seg3d_last_error()
void seg3d_set_error(const char* fmt, ...);

#define SEG3D_REQUIRE(cond, ...)            \
  do {                                      \
    if (!(cond)) {                          \
      seg3d_set_error(__VA_ARGS__);         \
      return SEG3D_ERR_INVALID;             \
    }                                       \
  } while (0)

#define SEG3D_UNSUPPORTED(...)              \
  do {                                      \
    seg3d_set_error(__VA_ARGS__);           \
    return SEG3D_ERR_UNSUPPORTED;           \
  } while (0)

#define SEG3D_LAUNCH_CHECK(name)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      seg3d_set_error("%s: kernel launch failed: %s", name, hipGetErrorString(e__)); \
      return SEG3D_ERR_LAUNCH;                                                       \
    }                                                                                \
  } while (0)

typedef long long i64;

// MI355X in SPX mode: 256 CUs.  The persistent grids ("one workgroup per CU") and the slab counts of the weight-gradient
// kernels -- and therefore the workspace-size queries, which are pure functions of the shape -- are sized for it.  The
// kernels are correct for any CU count (items / slabs are walked with a stride of gridDim.x); on a partitioned device
// (CPX: 32 CUs) they merely run in more rounds.
#define SEG3D_NUM_CUS 256

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is per DEVICE: several devices in one process are a supported flow
// (load_single_model / TrainStep select their device), so the "already configured" flag is kept per device ordinal.
struct Seg3dOncePerDevice {
  bool done[64] = {};
};
static inline int seg3d_allow_full_lds(const void* kernel, Seg3dOncePerDevice& once, const char* name) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) dev = 0;
  const bool tracked = dev >= 0 && dev < 64;
  if (tracked && once.done[dev]) return SEG3D_OK;
  hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)(160 * 1024));
  if (e != hipSuccess) {
    seg3d_set_error("%s: hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed on device %d: %s", name, dev, hipGetErrorString(e));
    return SEG3D_ERR_LAUNCH;
  }
  if (tracked) once.done[dev] = true;
  return SEG3D_OK;
}

// compute units of the current device (hipDeviceAttributeMultiprocessorCount, cached per device ordinal): the size of a
// persistent "one workgroup per CU" grid.  Workspace and statistics-slot counts do NOT depend on it (they are pure functions of
// the shape, see SEG3D_NUM_CUS); the kernels walk their items with a stride of gridDim.x, so any grid size is correct.
static inline int seg3d_device_cus() {
  static int cached[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SEG3D_NUM_CUS;
  if (dev >= 0 && dev < 64 && cached[dev] > 0) return cached[dev];
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = SEG3D_NUM_CUS;
  if (dev >= 0 && dev < 64) cached[dev] = n;
  return n;
}
static inline unsigned seg3d_persistent_grid(long long nitems) {
  const long long cus = seg3d_device_cus();
  return (unsigned)(nitems < cus ? nitems : cus);
}

static inline int seg3d_cdiv(i64 a, i64 b) { return (int)((a + b - 1) / b); }
static inline int seg3d_round_up(int a, int b) { return ((a + b - 1) / b) * b; }

// number of workgroups for a grid-stride elementwise kernel: enough to fill 256 CUs x 8 blocks
static inline int seg3d_ew_grid(i64 work_items, int block) {
  i64 g = (work_items + block - 1) / block;
  if (g > 8192) g = 8192;
  if (g < 1) g = 1;
  return (int)g;
}

#ifdef __HIPCC__
// ---- K-chunk step order of the F(2x2, 3x3) forward kernels (conv_wino2d.hip), shared with the T = 48 weight pack (layout.hip),
// whose image holds the A operands of four consecutive steps next to each other (one 16-byte LDS read per lane and four steps).
// 48 (kz, point) steps: the 16 Winograd points in PAIRS (a, b), each pair's six steps in the order a0 b0 a1 b1 a2 b2 (digit = kz),
// so that the operands of one point are read once and used for all three kz; the twelve ordinary points first, the four
// corner points 0, 3, 12, 15 (whose accumulators may still be waiting for a fused addend) in the last two pairs.
__host__ __device__ constexpr int seg3d_w2_point(int i) {   // i-th point of the order
  return i < 12 ? i + 1 + (i >= 2) + (i >= 10) : ((i - 12) >> 1) * 12 + ((i - 12) & 1) * 3;
}
__host__ __device__ constexpr int seg3d_w2_step_pi(int s) { return 2 * (s / 6) + ((s % 6) & 1); }   // index into the point order
__host__ __device__ constexpr int seg3d_w2_step_kz(int s) { return (s % 6) >> 1; }
__host__ __device__ constexpr int seg3d_w2_step_t(int s) { return seg3d_w2_step_kz(s) * 16 + seg3d_w2_point(seg3d_w2_step_pi(s)); }   // t = kz * 16 + py * 4 + px

// ---- wave64 / workgroup reductions -------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;  // valid in lane 0
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// v / d for 0 <= v < 2^22 (any divisor 1 <= d) without the ~35-instruction integer division: r = 1.0f / d, and
// (v + 0.5) * r is at least 0.5 / d away from an integer boundary while its float error (two roundings,
// <= (v / d) * 2^-23) stays below that for v < 2^22, so the truncation is exact.  Index decoding in kernels whose
// prologue is not hidden by other waves uses this; launchers check SEG3D_FDIV_MAX.
#define SEG3D_FDIV_MAX (1 << 22)
__device__ __forceinline__ int seg3d_fdiv(int v, float r) { return (int)(((float)v + 0.5f) * r); }

// bf16 storage helpers (bf16 mode: conv INPUTS and packed weights are bf16, accumulation and conv outputs fp32).
// A plain cast compiles to v_cvt_pk_bf16_f32 (round to nearest even, NaN stays NaN).
typedef unsigned short seg3d_bf16;  // raw bits at the C ABI
__device__ __forceinline__ seg3d_bf16 seg3d_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float seg3d_bf2f(seg3d_bf16 b) { return __uint_as_float((unsigned)b << 16); }
__device__ __forceinline__ unsigned seg3d_pack2bf(float lo, float hi) {
  return (unsigned)seg3d_f2bf(lo) | ((unsigned)seg3d_f2bf(hi) << 16);
}

// Four consecutive channels of an activation tensor that is fp32 (BF = false) or bf16 (BF = true).  `raw` is what a
// load returns (kept in staging registers untouched, so that loads stay back to back); cvt() widens it to fp32.
typedef float seg3d_f32x4 __attribute__((ext_vector_type(4)));
template <bool BF> struct Seg3dQuad;
template <> struct Seg3dQuad<false> {
  typedef seg3d_f32x4 raw;
  static __device__ __forceinline__ raw load(const void* p, i64 elem) {
    return *reinterpret_cast<const seg3d_f32x4*>(reinterpret_cast<const float*>(p) + elem);
  }
  static __device__ __forceinline__ seg3d_f32x4 cvt(raw r) { return r; }
  static __device__ __forceinline__ void store(void* p, i64 elem, seg3d_f32x4 v) {
    *reinterpret_cast<seg3d_f32x4*>(reinterpret_cast<float*>(p) + elem) = v;
  }
};
template <> struct Seg3dQuad<true> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw load(const void* p, i64 elem) {
    return *reinterpret_cast<const uint2*>(reinterpret_cast<const seg3d_bf16*>(p) + elem);
  }
  static __device__ __forceinline__ seg3d_f32x4 cvt(raw r) {
    seg3d_f32x4 v;
    v[0] = __uint_as_float(r.x << 16);
    v[1] = __uint_as_float(r.x & 0xffff0000u);
    v[2] = __uint_as_float(r.y << 16);
    v[3] = __uint_as_float(r.y & 0xffff0000u);
    return v;
  }
  static __device__ __forceinline__ void store(void* p, i64 elem, seg3d_f32x4 v) {
    uint2 o;
    o.x = seg3d_pack2bf(v[0], v[1]);
    o.y = seg3d_pack2bf(v[2], v[3]);
    *reinterpret_cast<uint2*>(reinterpret_cast<seg3d_bf16*>(p) + elem) = o;
  }
};

// XCD-contiguous block order for one-tile-per-workgroup kernels whose neighbouring tiles share halo voxels: workgroup b
// runs on XCD b % 8 (round-robin dispatch), so it takes tile (b % 8) * ceil(n / 8) + b / 8 -- XCD j covers the contiguous
// eighth j of the tile list and the halos of neighbours come through one L2.  Returns -1 for the padding blocks of the
// last eighth (n not a multiple of 8: launch seg3d_xcd_grid(n) blocks).
#ifndef SEG3D_XCD_TILES
#define SEG3D_XCD_TILES 1
#endif
static inline int seg3d_xcd_grid(int n) { return SEG3D_XCD_TILES ? ((n + 7) / 8) * 8 : n; }
__device__ __forceinline__ int seg3d_xcd_tile(int b, int n) {
  if (!SEG3D_XCD_TILES) return b;
  const int per = (n + 7) >> 3;
  const int t = (b & 7) * per + (b >> 3);
  return ((b >> 3) < per && t < n) ? t : -1;
}

// Sum NV values over a 256-thread workgroup. Result valid in thread 0. `red` must hold 4*NV floats.
template <int NV>
__device__ __forceinline__ void block_sum_256(float (&v)[NV], float* red) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < NV; ++k) v[k] = wave_sum(v[k]);
  __syncthreads();
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) red[wave * NV + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < NV; ++k) v[k] = red[k] + red[NV + k] + red[2 * NV + k] + red[3 * NV + k];
  }
}
#endif
