// conv_thin.hip -- the two "thin" 3x3x3 layers at full resolution, which are HBM-bound, not FLOP-bound
// (stem AI 12.7 FLOP/B, head AI 25 FLOP/B):
//   stem  InputBlock.conv   Conv3d(in_channels 1..4 -> 16, k3 p1)      network/module/vnet_inblock.py:9
//   head  OutputBlock.conv1 Conv3d(32 -> num_classes, k3 p1)           network/module/vnet_outblock.py:13
// One side of each has only a handful of channels, so the generic implicit GEMM of conv_mfma.hip would waste
// 8-16x of its MFMA rows/columns.  Three kernels cover forward, data-gradient and weight-gradient of both:
//
//   thin-in  (conv3d_k3_thin_in_kernel, MFMA):   y[v][b] = bias[b] + sum_{t, a < CT} x[v + t][a] W(a, b, t)
//       K = 27*CT <= 216 is folded into ONE GEMM K dimension: lane l supplies A[voxel][k] = x[v + off(t_k)][a_k] with a
//       per-lane LDS offset, the weights for all K live in registers (<= 108 VGPRs).  = stem forward, head dgrad.
//   thin-out (conv3d_k3_thin_out_kernel, VALU):  y[v][b < CO] = bias[b] + sum_{t, a} x[v + t][a] W(a, b, t)
//       LDS-tiled direct form, 2 voxels x CO outputs per thread, weights broadcast from LDS.  = head forward
//       (and stem dgrad).  12 GFLOP over 453 MB: the VALU keeps up with HBM here.
//   thin wgrad (k3_thin_wgrad_kernel, MFMA):     G[t][ct][cf] = sum_u fat[u][cf] * thin[u + off(t)][ct]
//       rows = (tap, thin channel) gathered per lane, columns = fat channels, K = voxels split over the 4 waves.
//       = stem wgrad (thin = x, fat = dy) and head wgrad (thin = dy, fat = x, taps reversed).
// All three emit / consume NDHWC fp32 and, where they produce a conv output, the GroupNorm partial statistics.
#include "seg3d_common.h"
#include "seg3d_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 to_bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int thin_row(int reg, int half) { return (reg & 3) + 8 * (reg >> 2) + 4 * half; }

// fixed output tile of the thin kernels: 4 x 8 x 8 = 256 voxels (2 MFMA row blocks per wave), halo tile 6 x 10 x 10
#define TH_TZ 4
#define TH_TY 8
#define TH_TX 8
#define TH_MT (TH_TZ * TH_TY * TH_TX)
#define TH_HY (TH_TY + 2)
#define TH_HX (TH_TX + 2)
#define TH_NV ((TH_TZ + 2) * TH_HY * TH_HX)  // 600

// ---- weight packer for the thin-in kernel: wp[bb][p][h][j] = W(a = k % CT, b = bb*32 + j, t = k / CT), k = 2p + h ----
__global__ __launch_bounds__(256) void pack_thin_in_kernel(const float* __restrict__ w, float* __restrict__ wp, int CT,
                                                             int B, int BB, int KP, i64 sa, i64 sb, int flip) {
  const i64 total = (i64)BB * KP * 64;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int j = (int)(idx & 31);
    const int h = (int)((idx >> 5) & 1);
    const i64 r = idx >> 6;
    const int p = (int)(r % KP);
    const int bb = (int)(r / KP);
    const int k = 2 * p + h;
    const int t = k / CT, a = k % CT;
    const int b = bb * 32 + j;
    float v = 0.f;
    if (t < 27 && b < B) v = w[a * sa + b * sb + (flip ? 26 - t : t)];
    wp[idx] = v;
  }
}

extern "C" long long seg3d_packed_thin_in_floats(int CT, int B) {
  return (long long)((B + 31) / 32) * ((27 * CT + 1) / 2) * 64;
}

extern "C" int seg3d_pack_weights_thin_in(const float* w, float* wp, int CT, int B, long long sa, long long sb, int flip,
                                          void* stream) {
  SEG3D_REQUIRE(w && wp && CT >= 1 && CT <= 8 && B > 0, "seg3d_pack_weights_thin_in: bad arguments (1 <= CT <= 8)");
  const int KP = (27 * CT + 1) / 2, BB = (B + 31) / 32;
  const i64 total = (i64)BB * KP * 64;
  hipLaunchKernelGGL(pack_thin_in_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, CT, B,
                     BB, KP, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_thin_in");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// thin-in forward
// ---------------------------------------------------------------------------------------------------------------
template <int CT>
__device__ __forceinline__ int thin_koff(int k) {  // LDS float offset of GEMM-k inside the halo tile
  const int t = k / CT, a = k % CT;
  const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
  return t < 27 ? ((kz * TH_HY + ky) * TH_HX + kx) * CT + a : 0;
}

// OUT_BF (bf16 mode): y is bf16 storage -- the stem's conv output and the head's data-gradient; statistics from fp32 values
template <int CT, bool OUT_BF>
__device__ __forceinline__ void conv3d_k3_thin_in_body(const float* __restrict__ x,
                                                                     const float* __restrict__ wp,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ stats,
                                                                     int N, int D, int H, int W, int Cout, int ntz,
                                                                     int nty, int ntx) {
  constexpr int KP = (27 * CT + 1) / 2;
  __shared__ __attribute__((aligned(16))) float xs[TH_NV * CT + 4];
  __shared__ int voff[TH_MT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cob = blockIdx.y;
  int b = blockIdx.x;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int z0 = tiz * TH_TZ, y0 = tiy * TH_TY, x0 = tix * TH_TX;

  // weights of this lane's column for every k-pair, kept in registers
  float bw[KP];
  {
    const float* wsrc = wp + ((i64)cob * KP * 2 + lh) * 32 + li;
#pragma unroll
    for (int p = 0; p < KP; ++p) bw[p] = wsrc[p * 64];
  }
  for (int e = tid; e < TH_NV * CT; e += 256) {
    const int v = e / CT, a = e % CT;
    const int hx = v % TH_HX;
    const int t = v / TH_HX;
    const int hy = t % TH_HY;
    const int hz = t / TH_HY;
    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
    float val = 0.f;
    if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
      val = x[((((i64)n * D + gz) * H + gy) * W + gx) * CT + a];
    xs[e] = val;
  }
  {
    const int tx = tid % TH_TX;
    const int t = tid / TH_TX;
    const int ty = t % TH_TY;
    const int tz = t / TH_TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    voff[tid] = (gz < D && gy < H && gx < W) ? ((n * D + gz) * H + gy) * W + gx : -1;
  }
  int abase[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int idx = (wave + 4 * m) * 32 + li;
    const int tx = idx % TH_TX;
    const int t = idx / TH_TX;
    const int ty = t % TH_TY;
    const int tz = t / TH_TY;
    abase[m] = ((tz * TH_HY + ty) * TH_HX + tx) * CT;
  }
  __syncthreads();
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
  for (int p = 0; p < KP; ++p) {
    const int koff = lh ? thin_koff<CT>(2 * p + 1) : thin_koff<CT>(2 * p);
    // weights are the ROW operand: the result puts a voxel in every lane (column li) and four consecutive output
    // channels in every register quad (rows 8 g + 4 lh ..), so the epilogue stores 16 (fp32) / 8 (bf16) contiguous bytes
    // per quad instead of one element per lane -- with 16 output channels the element stores also left half the lanes idle
#pragma unroll
    for (int m = 0; m < 2; ++m) acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[p], xs[abase[m] + koff], acc[m], 0, 0, 0);
  }

  float s[2] = {0.f, 0.f};
  const bool quad_ok = (Cout & 3) == 0;
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int vo = voff[(wave + 4 * m) * 32 + li];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co0 = cob * 32 + 8 * g + 4 * lh;
      float val[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const bool ok = vo >= 0 && co0 + j < Cout;
        val[j] = acc[m][4 * g + j] + ((bias && co0 + j < Cout) ? bias[co0 + j] : 0.f);
        const float sv = ok ? val[j] : 0.f;
        s[0] += sv;
        s[1] += sv * sv;
      }
      if (vo < 0 || co0 >= Cout) continue;
      const i64 o = (i64)vo * Cout + co0;
      if (quad_ok) {
        if (OUT_BF) {
          uint2 pk;
          pk.x = seg3d_pack2bf(val[0], val[1]);
          pk.y = seg3d_pack2bf(val[2], val[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<seg3d_bf16*>(y) + o) = pk;
        } else {
          const f32x4 v4 = {val[0], val[1], val[2], val[3]};
          *reinterpret_cast<f32x4*>(y + o) = v4;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
          if (co0 + j < Cout) {
            if (OUT_BF) reinterpret_cast<seg3d_bf16*>(y)[o + j] = seg3d_f2bf(val[j]);
            else y[o + j] = val[j];
          }
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

template <int CT>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_in_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ wp,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ stats,
                                                                     int N, int D, int H, int W, int Cout, int ntz,
                                                                     int nty, int ntx) {
  conv3d_k3_thin_in_body<CT, false>(x, wp, bias, y, stats, N, D, H, W, Cout, ntz, nty, ntx);
}

template <int CT>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_in_bf16out_kernel(const float* __restrict__ x,
                                                                     const float* __restrict__ wp,
                                                                     const float* __restrict__ bias,
                                                                     float* __restrict__ y, float* __restrict__ stats,
                                                                     int N, int D, int H, int W, int Cout, int ntz,
                                                                     int nty, int ntx) {
  conv3d_k3_thin_in_body<CT, true>(x, wp, bias, y, stats, N, D, H, W, Cout, ntz, nty, ntx);
}

// ---------------------------------------------------------------------------------------------------------------
// thin-in forward on the bf16 matrix cores (bf16 mode: stem forward CT = 1, head data-gradient CT = 2; bf16 output)
// ---------------------------------------------------------------------------------------------------------------
// The fp32 kernel above issues 14 / 27 fp32 MFMAs of 64 cycles per row block and is bound by instruction issue.  Here
// a K-step is 16 reduction entries of one v_mfma_f32_32x32x16_bf16, and both operands are bf16 hi + lo pairs (three
// MFMAs per K-step: hi.hi + hi.lo + lo.hi, exact to 2^-16): the stem and the head stay fp32-grade.  A lane's operand
// is 8 reduction entries of ITS voxel = four dwords, each ONE aligned LDS read:
//   CT = 2: k = 2 tap + channel; the halo tile holds one dword (ch0 | ch1 << 16) per voxel; K = 54 -> 4 K-steps;
//   CT = 1: k = 4 (kz 3 + ky) + slot, slot = kx 0, 1, 2 and one zero-weight filler; a dword is the x pair (hx, hx + 1),
//           kept in two copies (pairs starting at even / odd hx) so that either parity of the lane's x is an aligned
//           read; K = 36 -> 3 K-steps.
// Weights: the ROW operand, in registers (wq[hi, lo][K-step][lane][8]); result and epilogue as in the fp32 kernel.
template <int CT> struct ThinIn16 {
  static constexpr int KS = CT == 1 ? 3 : 4;
  static constexpr int ROWD = 6;                                        // CT = 1: dwords per halo row (12 elements)
  static constexpr int PLANE = CT == 1 ? (TH_TZ + 2) * TH_HY * ROWD : TH_NV;   // dwords of one LDS image
  static constexpr int IMAGES = CT == 1 ? 4 : 2;                        // CT = 1: even hi, odd hi, even lo, odd lo
};

// k of K-step ks, lane half h, entry j -> (tap, channel) or (-1) for a zero weight
template <int CT>
__host__ __device__ __forceinline__ int thin16_tap(int ks, int h, int j, int* ch) {
  const int k = 16 * ks + 8 * h + j;
  if (CT == 2) {
    *ch = k & 1;
    return (k >> 1) < 27 ? (k >> 1) : -1;
  }
  *ch = 0;
  const int row = k >> 2, slot = k & 3;
  return (row < 9 && slot < 3) ? row * 3 + slot : -1;
}

template <int CT>
__global__ __launch_bounds__(256) void pack_thin_in16_kernel(const float* __restrict__ w, seg3d_bf16* __restrict__ wq, int B,
                                                               int BB, i64 sa, i64 sb, int flip) {
  constexpr int KS = ThinIn16<CT>::KS;
  const int total = BB * KS * 512;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int j = idx & 7, l = (idx >> 3) & 63, r = idx >> 9;
    const int ks = r % KS, bb = r / KS;
    int a;
    const int t = thin16_tap<CT>(ks, l >> 5, j, &a);
    const int b = bb * 32 + (l & 31);
    float v = 0.f;
    if (t >= 0 && b < B) v = w[a * sa + b * sb + (flip ? 26 - t : t)];
    const seg3d_bf16 hi = seg3d_f2bf(v);
    wq[idx] = hi;
    wq[total + idx] = seg3d_f2bf(v - seg3d_bf2f(hi));
  }
}

extern "C" int seg3d_conv3d_k3_thin_in_mfma16_supported(int CT, int Cout) { return (CT == 1 || CT == 2) && Cout > 0 && (Cout % 4) == 0; }

extern "C" long long seg3d_packed_thin_in16_elems(int CT, int B) {
  return 2ll * ((B + 31) / 32) * (CT == 1 ? 3 : 4) * 512;
}

extern "C" int seg3d_pack_weights_thin_in16(const float* w, void* wq_bf16, int CT, int B, long long sa, long long sb, int flip,
                                            void* stream) {
  SEG3D_REQUIRE(w && wq_bf16 && (CT == 1 || CT == 2) && B > 0, "seg3d_pack_weights_thin_in16: bad arguments (CT in {1, 2})");
  const int BB = (B + 31) / 32;
  const int total = BB * (CT == 1 ? 3 : 4) * 512;
  seg3d_bf16* wq = reinterpret_cast<seg3d_bf16*>(wq_bf16);
  if (CT == 1)
    hipLaunchKernelGGL(pack_thin_in16_kernel<1>, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wq, B, BB,
                       (i64)sa, (i64)sb, flip);
  else
    hipLaunchKernelGGL(pack_thin_in16_kernel<2>, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wq, B, BB,
                       (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_thin_in16");
  return SEG3D_OK;
}

template <int CT>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_in_mfma16_kernel(const float* __restrict__ x,
                                                                           const seg3d_bf16* __restrict__ wq,
                                                                           const float* __restrict__ bias,
                                                                           seg3d_bf16* __restrict__ y, float* __restrict__ stats,
                                                                           int N, int D, int H, int W, int Cout, int ntz,
                                                                           int nty, int ntx) {
  typedef ThinIn16<CT> TI;
  constexpr int KS = TI::KS, PL = TI::PLANE;
  __shared__ __attribute__((aligned(16))) unsigned img[TI::IMAGES * PL + 64];   // (+ the block_sum scratch fits as well)
  __shared__ int voff[TH_MT];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, lh = lane >> 5;
  const int cob = blockIdx.y;
  int b = blockIdx.x;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int z0 = tiz * TH_TZ, y0 = tiy * TH_TY, x0 = tix * TH_TX;

  f32x4 whi[KS], wlo[KS];
  {
    const seg3d_bf16* wsrc = wq + ((i64)cob * KS * 64 + lane) * 8;
    const i64 lo_off = (i64)gridDim.y * KS * 512;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      whi[ks] = *reinterpret_cast<const f32x4*>(wsrc + ks * 512);
      wlo[ks] = *reinterpret_cast<const f32x4*>(wsrc + lo_off + ks * 512);
    }
  }
  seg3d_bf16* img16 = reinterpret_cast<seg3d_bf16*>(img);
  if (CT == 1) {   // the filler elements (hx 10, 11) are read with zero weights: they must be finite
    for (int e = tid; e < TI::IMAGES * PL; e += 256) img[e] = 0u;
    __syncthreads();
  }
  for (int e = tid; e < TH_NV * CT; e += 256) {
    const int v = e / CT, a = e % CT;
    const int hx = v % TH_HX;
    const int row = v / TH_HX;           // hz * TH_HY + hy
    const int hy = row % TH_HY;
    const int hz = row / TH_HY;
    const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
    float val = 0.f;
    if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
      val = x[((((i64)n * D + gz) * H + gy) * W + gx) * CT + a];
    const seg3d_bf16 hi = seg3d_f2bf(val);
    const seg3d_bf16 lo = seg3d_f2bf(val - seg3d_bf2f(hi));
    if (CT == 2) {
      img16[2 * v + a] = hi;
      img16[2 * (PL + v) + a] = lo;
    } else {
      // even image: pair (2 i, 2 i + 1) in dword i; odd image: pair (2 i + 1, 2 i + 2) in dword i
      const int ev = 2 * (row * TI::ROWD) + hx;
      img16[ev] = hi;
      img16[2 * 2 * PL + ev] = lo;
      if (hx >= 1) {
        const int od = 2 * (PL + row * TI::ROWD) + hx - 1;
        img16[od] = hi;
        img16[2 * 2 * PL + od] = lo;
      }
    }
  }
  {
    const int tx = tid % TH_TX;
    const int t = tid / TH_TX;
    const int ty = t % TH_TY;
    const int tz = t / TH_TY;
    const int gz = z0 + tz, gy = y0 + ty, gx = x0 + tx;
    voff[tid] = (gz < D && gy < H && gx < W) ? ((n * D + gz) * H + gy) * W + gx : -1;
  }
  // dword offset of this lane's voxel (tap 0) in the hi image, per row block
  int abase[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int idx = (wave + 4 * m) * 32 + li;
    const int tx = idx % TH_TX;
    const int t = idx / TH_TX;
    const int ty = t % TH_TY;
    const int tz = t / TH_TY;
    if (CT == 2) abase[m] = (tz * TH_HY + ty) * TH_HX + tx;
    else abase[m] = (tx & 1) * PL + (tz * TH_HY + ty) * TI::ROWD + (tx >> 1);
  }
  __syncthreads();
  f32x16 acc[2];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
  constexpr int LO = CT == 1 ? 2 * PL : PL;   // dword distance hi image -> lo image
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      unsigned xh[4], xl[4];
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        int off0, off1;   // dword offsets for lane half 0 / 1 (compile-time)
        if (CT == 2) {
          const int t0 = 8 * ks + d, t1 = 8 * ks + 4 + d;
          off0 = t0 < 27 ? ((t0 / 9) * TH_HY + (t0 / 3) % 3) * TH_HX + t0 % 3 : 0;
          off1 = t1 < 27 ? ((t1 / 9) * TH_HY + (t1 / 3) % 3) * TH_HX + t1 % 3 : 0;
        } else {
          const int r0 = 4 * ks + (d >> 1), r1 = 4 * ks + 2 + (d >> 1);
          off0 = r0 < 9 ? ((r0 / 3) * TH_HY + r0 % 3) * TI::ROWD + (d & 1) : 0;
          off1 = r1 < 9 ? ((r1 / 3) * TH_HY + r1 % 3) * TI::ROWD + (d & 1) : 0;
        }
        const int off = abase[m] + (lh ? off1 : off0);
        xh[d] = img[off];
        xl[d] = img[LO + off];
      }
      const f32x4 bh = {__uint_as_float(xh[0]), __uint_as_float(xh[1]), __uint_as_float(xh[2]), __uint_as_float(xh[3])};
      const f32x4 bl = {__uint_as_float(xl[0]), __uint_as_float(xl[1]), __uint_as_float(xl[2]), __uint_as_float(xl[3])};
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, whi[ks]), __builtin_bit_cast(to_bf16x8, bh),
                                                       acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, whi[ks]), __builtin_bit_cast(to_bf16x8, bl),
                                                       acc[m], 0, 0, 0);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, wlo[ks]), __builtin_bit_cast(to_bf16x8, bh),
                                                       acc[m], 0, 0, 0);
    }
  }

  float s[2] = {0.f, 0.f};
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int vo = voff[(wave + 4 * m) * 32 + li];
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int co0 = cob * 32 + 8 * g + 4 * lh;    // Cout % 4 == 0: a quad is inside or outside
      if (co0 >= Cout) continue;
      float val[4];
      const f32x4 bq = bias ? *reinterpret_cast<const f32x4*>(bias + co0) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        val[j] = acc[m][4 * g + j] + bq[j];
        const float sv = vo >= 0 ? val[j] : 0.f;
        s[0] += sv;
        s[1] += sv * sv;
      }
      if (vo < 0) continue;
      uint2 pk;
      pk.x = seg3d_pack2bf(val[0], val[1]);
      pk.y = seg3d_pack2bf(val[2], val[3]);
      *reinterpret_cast<uint2*>(y + (i64)vo * Cout + co0) = pk;
    }
  }
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, reinterpret_cast<float*>(img));
    if (tid == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + (((i64)n * tiles_per_sample + tile) * gridDim.y + cob) * 2;
      dst[0] = s[0];
      dst[1] = s[1];
    }
  }
}

// x fp32 [N][D][H][W][CT] (CT in {1, 2}), wq = seg3d_pack_weights_thin_in16, y bf16 [N][D][H][W][Cout] (Cout % 4 == 0);
// stats as seg3d_conv3d_k3_thin_in_fwd
extern "C" int seg3d_conv3d_k3_thin_in_mfma16_fwd(const float* x, const void* wq_bf16, const float* bias, void* y_bf16,
                                                  float* stats, int N, int D, int H, int W, int CT, int Cout, void* stream) {
  SEG3D_REQUIRE(x && wq_bf16 && y_bf16, "seg3d_conv3d_k3_thin_in_mfma16_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "seg3d_conv3d_k3_thin_in_mfma16_fwd: bad dims");
  SEG3D_REQUIRE(seg3d_conv3d_k3_thin_in_mfma16_supported(CT, Cout),
                "seg3d_conv3d_k3_thin_in_mfma16_fwd: need CT in {1, 2} and Cout %% 4 == 0");
  SEG3D_REQUIRE((i64)N * D * H * W * Cout < (1ll << 31), "seg3d_conv3d_k3_thin_in_mfma16_fwd: tensor exceeds 2^31 elements");
  const int ntz = seg3d_cdiv(D, TH_TZ), nty = seg3d_cdiv(H, TH_TY), ntx = seg3d_cdiv(W, TH_TX);
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  hipStream_t s = (hipStream_t)stream;
  const seg3d_bf16* wq = reinterpret_cast<const seg3d_bf16*>(wq_bf16);
  seg3d_bf16* y = reinterpret_cast<seg3d_bf16*>(y_bf16);
  if (CT == 1)
    hipLaunchKernelGGL((conv3d_k3_thin_in_mfma16_kernel<1>), grid, dim3(256), 0, s, x, wq, bias, y, stats, N, D, H, W, Cout, ntz,
                       nty, ntx);
  else
    hipLaunchKernelGGL((conv3d_k3_thin_in_mfma16_kernel<2>), grid, dim3(256), 0, s, x, wq, bias, y, stats, N, D, H, W, Cout, ntz,
                       nty, ntx);
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_in_mfma16_fwd");
  return SEG3D_OK;
}

extern "C" long long seg3d_conv3d_k3_thin_stats_count(int D, int H, int W, int Cout_blocks) {
  return (long long)seg3d_cdiv(D, TH_TZ) * seg3d_cdiv(H, TH_TY) * seg3d_cdiv(W, TH_TX) * Cout_blocks;
}

template <int CT>
static void launch_thin_in(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                           int W, int Cout, hipStream_t s, int out_bf16) {
  const int ntz = seg3d_cdiv(D, TH_TZ), nty = seg3d_cdiv(H, TH_TY), ntx = seg3d_cdiv(W, TH_TX);
  dim3 grid((unsigned)(N * ntz * nty * ntx), (unsigned)((Cout + 31) / 32));
  if (out_bf16)
    hipLaunchKernelGGL((conv3d_k3_thin_in_bf16out_kernel<CT>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H, W, Cout,
                       ntz, nty, ntx);
  else
    hipLaunchKernelGGL((conv3d_k3_thin_in_kernel<CT>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H, W, Cout, ntz,
                       nty, ntx);
}

// x [N][D][H][W][CT] (CT <= 8), wp = seg3d_pack_weights_thin_in, y [N][D][H][W][Cout];
// stats (optional): [N][seg3d_conv3d_k3_thin_stats_count(D,H,W,ceil(Cout/32))][2]
static int thin_in_launch(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                          int W, int CT, int Cout, int out_bf16, void* stream);

extern "C" int seg3d_conv3d_k3_thin_in_fwd(const float* x, const float* wp, const float* bias, float* y, float* stats, int N,
                                           int D, int H, int W, int CT, int Cout, void* stream) {
  return thin_in_launch(x, wp, bias, y, stats, N, D, H, W, CT, Cout, 0, stream);
}

// bf16 mode: y is bf16 storage (stem conv output / head data-gradient); x, weights, bias and statistics fp32
extern "C" int seg3d_conv3d_k3_thin_in_bf16out_fwd(const float* x, const float* wp, const float* bias, void* y_bf16,
                                                   float* stats, int N, int D, int H, int W, int CT, int Cout, void* stream) {
  return thin_in_launch(x, wp, bias, reinterpret_cast<float*>(y_bf16), stats, N, D, H, W, CT, Cout, 1, stream);
}

static int thin_in_launch(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D, int H,
                          int W, int CT, int Cout, int out_bf16, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_thin_in_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, "seg3d_conv3d_k3_thin_in_fwd: bad dims");
  SEG3D_REQUIRE(CT >= 1 && CT <= 8, "seg3d_conv3d_k3_thin_in_fwd: thin channel count %d not in [1, 8]", CT);
  SEG3D_REQUIRE((i64)N * D * H * W * Cout < (1ll << 31), "seg3d_conv3d_k3_thin_in_fwd: tensor exceeds 2^31 elements");
  hipStream_t s = (hipStream_t)stream;
  switch (CT) {
    case 1: launch_thin_in<1>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 2: launch_thin_in<2>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 3: launch_thin_in<3>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 4: launch_thin_in<4>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 5: launch_thin_in<5>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 6: launch_thin_in<6>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 7: launch_thin_in<7>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
    default: launch_thin_in<8>(x, wp, bias, y, stats, N, D, H, W, Cout, s, out_bf16); break;
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_in_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// thin-out forward (VALU): tile 4 x 8 x 16 = 512 voxels, 2 voxels per thread, 8 input channels per LDS chunk
// ---------------------------------------------------------------------------------------------------------------
#define TO_TZ 8
#define TO_TY 8
#define TO_TX 16
#define TO_HY (TO_TY + 2)
#define TO_HX (TO_TX + 2)
#define TO_NV ((TO_TZ + 2) * TO_HY * TO_HX)  // 1080
#define TO_E ((2 * TO_NV + 255) / 256)       // 9 float4 per thread per chunk

// weights: wq[cib][tap][8][CO] (tap-major inside an 8-channel chunk), zero padded
__global__ __launch_bounds__(256) void pack_thin_out_kernel(const float* __restrict__ w, float* __restrict__ wq, int A, int B,
                                                              int CO, int AB, i64 sa, i64 sb, int flip) {
  const i64 total = (i64)AB * 27 * 8 * CO;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int b = (int)(idx % CO);
    i64 r = idx / CO;
    const int a8 = (int)(r % 8);
    r /= 8;
    const int t = (int)(r % 27);
    const int ab = (int)(r / 27);
    const int a = ab * 8 + a8;
    float v = 0.f;
    if (a < A && b < B) v = w[a * sa + b * sb + (flip ? 26 - t : t)];
    wq[idx] = v;
  }
}

extern "C" int seg3d_pack_weights_thin_out(const float* w, float* wq, int A, int B, int CO, long long sa, long long sb,
                                           int flip, void* stream) {
  SEG3D_REQUIRE(w && wq && A > 0 && B > 0 && B <= CO && (CO == 2 || CO == 4 || CO == 8),
                "seg3d_pack_weights_thin_out: bad arguments (CO in {2,4,8}, B <= CO)");
  const int AB = (A + 7) / 8;
  const i64 total = (i64)AB * 27 * 8 * CO;
  hipLaunchKernelGGL(pack_thin_out_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wq, A, B,
                     CO, AB, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_thin_out");
  return SEG3D_OK;
}

// Tile 8 x 8 x 16 outputs; a thread owns FOUR consecutive x outputs of one (z, y) row: per (kz, ky) it reads the six
// input voxels of that row once (12 ds_read_b128 for 8 channels) and uses them for the three kx taps of all four
// outputs (1/16 LDS read per FMA).  The halo row stride is padded to 19 voxels so that the 64 lanes of a ds_read_b128
// hit 64 distinct banks (row stride 76 dwords, x-group stride 16 dwords).  Weights are wave-uniform: scalar loads.
// Measured (round 1): 369 us for the 96^3 32 -> 2 head, 4x its HBM time.  PMC shows 2.8 GB fetched for a 453 MB input:
// the 8-channel chunks read every 128-byte voxel row in four 32-byte passes, and with ~230 KB of halo rows per
// workgroup x 64 workgroups per XCD the rows are evicted from the 4 MB L2 between passes.  Halving the LDS reads
// (this version) therefore changed little; staging whole rows once is the fix that is still open.
#define TO_PX (TO_TX + 3)                        // padded halo row: 19 voxels
#define TO_NVP ((TO_TZ + 2) * TO_HY * TO_PX)     // 1900 padded halo voxels
// XBF (bf16 mode): x is bf16, widened when the staged chunk is written to LDS
template <int CO, bool XBF>
__device__ __forceinline__ void conv3d_k3_thin_out_body(const void* __restrict__ x, const float* __restrict__ wq,
                                                        const float* __restrict__ bias, float* __restrict__ y,
                                                        float* __restrict__ stats, int N, int D, int H, int W, int Cin,
                                                        int Cout, int ntz, int nty, int ntx) {
  __shared__ __attribute__((aligned(16))) float xs[8 * TO_NVP];      // [2][NVP][4]  (60.8 KB)
  const int tid = threadIdx.x;
  int b = seg3d_xcd_tile(blockIdx.x, N * ntz * nty * ntx);   // neighbouring tiles (shared halo rows) on one XCD
  if (b < 0) return;
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  const int z0 = tiz * TO_TZ, y0 = tiy * TO_TY, x0 = tix * TO_TX;
  const int CIB = (Cin + 7) >> 3;
  const int hh = tid & 1;
  // staged float4 e = tid + 256 k: voxel v = e >> 1 of the UNPADDED halo (TO_HX = 18 per row), half hh
  int goff[TO_E], loff[TO_E];
#pragma unroll
  for (int e = 0; e < TO_E; ++e) {
    const int eidx = tid + e * 256;
    goff[e] = -1;
    loff[e] = -1;
    if (eidx < 2 * TO_NV) {
      const int v = eidx >> 1;
      const int t = seg3d_fdiv(v, 1.0f / (float)TO_HX);
      const int hx = v - t * TO_HX;
      const int hz = seg3d_fdiv(t, 1.0f / (float)TO_HY);
      const int hy = t - hz * TO_HY;
      loff[e] = (hh * TO_NVP + (hz * TO_HY + hy) * TO_PX + hx) * 4;
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      if (gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W)
        goff[e] = (((n * D + gz) * H + gy) * W + gx) * Cin + hh * 4;
    }
  }
  // this thread's outputs: (tz, ty, 4 xg .. 4 xg + 3)
  const int xg = tid & 3, ty = (tid >> 2) & 7, tz = tid >> 5;
  const int vb = ((tz * TO_HY + ty) * TO_PX + 4 * xg) * 4;
  float acc[4][CO];
#pragma unroll
  for (int o = 0; o < 4; ++o)
#pragma unroll
    for (int c = 0; c < CO; ++c) acc[o][c] = 0.f;

  typename Seg3dQuad<XBF>::raw xst[TO_E];
  auto load_chunk = [&](int cib) {
    const bool half_ok = cib * 8 + hh * 4 < Cin;
#pragma unroll
    for (int e = 0; e < TO_E; ++e) {
      const bool ok = goff[e] >= 0 && half_ok;
      xst[e] = Seg3dQuad<XBF>::load(x, ok ? (i64)goff[e] + cib * 8 : (i64)0);
    }
  };
  load_chunk(0);
  for (int cib = 0; cib < CIB; ++cib) {
    __syncthreads();
    {
      const bool half_ok = cib * 8 + hh * 4 < Cin;
      const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int e = 0; e < TO_E; ++e)
        if (loff[e] >= 0)
          *reinterpret_cast<f32x4*>(xs + loff[e]) = (goff[e] >= 0 && half_ok) ? Seg3dQuad<XBF>::cvt(xst[e]) : zero;
    }
    __syncthreads();
    if (cib + 1 < CIB) load_chunk(cib + 1);
    const float* __restrict__ wchunk = wq + (i64)cib * 27 * 8 * CO;
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int ky = 0; ky < 3; ++ky) {
        const int rowoff = vb + ((kz * TO_HY + ky) * TO_PX) * 4;
        f32x4 lo[6], hi[6];   // channels 0-3 / 4-7 of the six input voxels of this row
#pragma unroll
        for (int i = 0; i < 6; ++i) {
          lo[i] = *reinterpret_cast<const f32x4*>(xs + rowoff + 4 * i);
          hi[i] = *reinterpret_cast<const f32x4*>(xs + TO_NVP * 4 + rowoff + 4 * i);
        }
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const float* wt = wchunk + ((kz * 3 + ky) * 3 + kx) * 8 * CO;   // wave-uniform address: scalar loads
#pragma unroll
          for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int c = 0; c < CO; ++c) {
              const float w0 = wt[a * CO + c], w1 = wt[(4 + a) * CO + c];
#pragma unroll
              for (int o = 0; o < 4; ++o) {
                acc[o][c] = fmaf(lo[o + kx][a], w0, acc[o][c]);
                acc[o][c] = fmaf(hi[o + kx][a], w1, acc[o][c]);
              }
            }
        }
      }
  }
  float s[2] = {0.f, 0.f};
  const int gz = z0 + tz, gy = y0 + ty;
  if (gz < D && gy < H) {
#pragma unroll
    for (int o = 0; o < 4; ++o) {
      const int gx = x0 + 4 * xg + o;
      if (gx < W) {
        float* yp = y + ((((i64)n * D + gz) * H + gy) * W + gx) * Cout;
#pragma unroll
        for (int c = 0; c < CO; ++c) {
          if (c < Cout) {
            const float val = acc[o][c] + (bias ? bias[c] : 0.f);
            yp[c] = val;
            s[0] += val;
            s[1] += val * val;
          }
        }
      }
    }
  }
  if (stats) {
    __syncthreads();
    block_sum_256<2>(s, xs);
    if (tid == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + ((i64)n * tiles_per_sample + tile) * 2;
      dst[0] = s[0];
      dst[1] = s[1];
    }
  }
}

template <int CO>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_out_kernel(const float* __restrict__ x,
                                                                      const float* __restrict__ wq,
                                                                      const float* __restrict__ bias,
                                                                      float* __restrict__ y, float* __restrict__ stats,
                                                                      int N, int D, int H, int W, int Cin, int Cout,
                                                                      int ntz, int nty, int ntx) {
  conv3d_k3_thin_out_body<CO, false>(x, wq, bias, y, stats, N, D, H, W, Cin, Cout, ntz, nty, ntx);
}

template <int CO>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_out_bf16_kernel(const void* __restrict__ x,
                                                                           const float* __restrict__ wq,
                                                                           const float* __restrict__ bias,
                                                                           float* __restrict__ y, float* __restrict__ stats,
                                                                           int N, int D, int H, int W, int Cin, int Cout,
                                                                           int ntz, int nty, int ntx) {
  conv3d_k3_thin_out_body<CO, true>(x, wq, bias, y, stats, N, D, H, W, Cin, Cout, ntz, nty, ntx);
}

extern "C" long long seg3d_conv3d_k3_thin_out_stats_count(int D, int H, int W) {
  return (long long)seg3d_cdiv(D, TO_TZ) * seg3d_cdiv(H, TO_TY) * seg3d_cdiv(W, TO_TX);
}

// x [N][D][H][W][Cin] (Cin % 4 == 0), wq = seg3d_pack_weights_thin_out(CO), y [N][D][H][W][Cout], Cout <= CO <= 8
static int thin_out_launch(const void* xv, int x_bf16, const float* wq, const float* bias, float* y, float* stats, int N,
                           int D, int H, int W, int Cin, int Cout, int CO, void* stream);

extern "C" int seg3d_conv3d_k3_thin_out_fwd(const float* x, const float* wq, const float* bias, float* y, float* stats,
                                            int N, int D, int H, int W, int Cin, int Cout, int CO, void* stream) {
  return thin_out_launch(x, 0, wq, bias, y, stats, N, D, H, W, Cin, Cout, CO, stream);
}

// bf16 mode: x bf16, everything else as seg3d_conv3d_k3_thin_out_fwd
extern "C" int seg3d_conv3d_k3_thin_out_bf16_fwd(const void* x_bf16, const float* wq, const float* bias, float* y,
                                                 float* stats, int N, int D, int H, int W, int Cin, int Cout, int CO,
                                                 void* stream) {
  return thin_out_launch(x_bf16, 1, wq, bias, y, stats, N, D, H, W, Cin, Cout, CO, stream);
}

static int thin_out_launch(const void* xv, int x_bf16, const float* wq, const float* bias, float* y, float* stats, int N,
                           int D, int H, int W, int Cin, int Cout, int CO, void* stream) {
  const float* x = reinterpret_cast<const float*>(xv);
  SEG3D_REQUIRE(x && wq && y, "seg3d_conv3d_k3_thin_out_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "seg3d_conv3d_k3_thin_out_fwd: bad dims");
  SEG3D_REQUIRE((Cin % 4) == 0 && Cout <= CO, "seg3d_conv3d_k3_thin_out_fwd: need Cin %% 4 == 0 and Cout <= CO");
  SEG3D_REQUIRE((i64)N * D * H * W * Cin < (1ll << 31), "seg3d_conv3d_k3_thin_out_fwd: tensor exceeds 2^31 elements");
  const int ntz = seg3d_cdiv(D, TO_TZ), nty = seg3d_cdiv(H, TO_TY), ntx = seg3d_cdiv(W, TO_TX);
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "seg3d_conv3d_k3_thin_out_fwd: more than 2^22 tiles");
  dim3 grid((unsigned)seg3d_xcd_grid(N * ntz * nty * ntx));
  hipStream_t s = (hipStream_t)stream;
  if (x_bf16) {
    if (CO == 2)
      hipLaunchKernelGGL((conv3d_k3_thin_out_bf16_kernel<2>), grid, dim3(256), 0, s, xv, wq, bias, y, stats, N, D, H, W, Cin,
                         Cout, ntz, nty, ntx);
    else if (CO == 4)
      hipLaunchKernelGGL((conv3d_k3_thin_out_bf16_kernel<4>), grid, dim3(256), 0, s, xv, wq, bias, y, stats, N, D, H, W, Cin,
                         Cout, ntz, nty, ntx);
    else if (CO == 8)
      hipLaunchKernelGGL((conv3d_k3_thin_out_bf16_kernel<8>), grid, dim3(256), 0, s, xv, wq, bias, y, stats, N, D, H, W, Cin,
                         Cout, ntz, nty, ntx);
    else
      SEG3D_UNSUPPORTED("seg3d_conv3d_k3_thin_out_fwd: CO must be 2, 4 or 8 (got %d)", CO);
  } else if (CO == 2) {
    hipLaunchKernelGGL((conv3d_k3_thin_out_kernel<2>), grid, dim3(256), 0, s, x, wq, bias, y, stats, N, D, H, W, Cin, Cout, ntz,
                       nty, ntx);
  } else if (CO == 4) {
    hipLaunchKernelGGL((conv3d_k3_thin_out_kernel<4>), grid, dim3(256), 0, s, x, wq, bias, y, stats, N, D, H, W, Cin, Cout, ntz,
                       nty, ntx);
  } else if (CO == 8) {
    hipLaunchKernelGGL((conv3d_k3_thin_out_kernel<8>), grid, dim3(256), 0, s, x, wq, bias, y, stats, N, D, H, W, Cin, Cout, ntz,
                       nty, ntx);
  } else {
    SEG3D_UNSUPPORTED("seg3d_conv3d_k3_thin_out_fwd: CO must be 2, 4 or 8 (got %d)", CO);
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_out_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// Thin-output head on the matrix cores (bf16 activations).  The VALU kernel above spends 27 * Cin * Cout FMAs per voxel
// and is VALU bound (215 us for the 96^3 32 -> 2 head on bf16 input, its HBM time is ~50 us).  Here the x taps go into
// the reduction dimension and the (kz, ky) taps into the output dimension of ONE small GEMM per row block:
//     P[v][(kz, ky, co)] = sum_{kx, ci} x[v + (kx - 1)][ci] * w[kz][ky][kx][ci][co]          (K = 3 Cin, 9 Cout <= 32 columns)
//     y[z][y][x][co]     = sum_{kz, ky} P[(z + kz - 1, y + ky - 1, x)][(kz, ky, co)]           (9 Cout adds per voxel)
// NDHWC makes x[v - 1 .. v + 1][0 .. Cin) ONE contiguous run, so a lane's 8-channel operand chunk of any K-step is a
// 16-byte piece of global memory: the kernel loads the voxel itself once and takes its x neighbours from the
// neighbouring lanes (DPP row shifts) -- the activations never pass through LDS.  The weights are the other MFMA
// operand, resident in registers for the whole kernel, as a bf16 hi + lo pair (two MFMAs per K-step) so the head keeps
// fp32-grade weights (2^-17) like the rest of the 2..5-channel tail.  Only P goes through LDS, wave-private and one halo
// plane at a time.  ONE WAVE owns one 8 x 8 x 16 tile (the tile and statistics slot of the VALU kernel; a workgroup is
// four independent waves, no barrier anywhere): it walks the 10 halo planes in z, 5 row blocks (2 rows x 16 x) each,
// and after each plane adds the plane's 3 x 3 shifted P entries into three rolling accumulator sets -- output planes
// p, p - 1, p - 2 take the plane's kz = 0, 1, 2 taps -- then stores the finished plane p - 2 (16 bytes per lane) and
// rotates.  Per output voxel: 50 / 1024 row blocks (z halo 10 / 8, y halo 10 / 8), 18 LDS floats read, 18 adds.

// wp[2 (hi, lo)][KS][64 lanes][8]: lane l supplies row n = l & 31 = (kz * 3 + ky) * COUT + co, k = 16 ks + 8 (l >> 5) + j
__global__ __launch_bounds__(256) void pack_thin_out_mfma_kernel(const float* __restrict__ w, seg3d_bf16* __restrict__ wp,
                                                                   int A, int B, int COUT, i64 sa, i64 sb, int flip) {
  const int KS = 3 * A / 16;
  const int total = KS * 64 * 8;
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    const int j = idx & 7, l = (idx >> 3) & 63, ks = idx >> 9;
    const int nrow = l & 31, k = ks * 16 + 8 * (l >> 5) + j;
    const int kx = k / A, ci = k - kx * A;
    const int t9 = nrow / COUT, co = nrow - t9 * COUT;
    float v = 0.f;
    if (t9 < 9 && co < B) {
      const int t = t9 * 3 + kx;
      v = w[ci * sa + co * sb + (flip ? 26 - t : t)];
    }
    const seg3d_bf16 hi = seg3d_f2bf(v);
    wp[idx] = hi;
    wp[total + idx] = seg3d_f2bf(v - seg3d_bf2f(hi));
  }
}

extern "C" int seg3d_conv3d_k3_thin_out_mfma_supported(int Cin, int Cout) {
  return (Cin == 16 || Cin == 32) && Cout >= 1 && Cout <= 3;
}

extern "C" long long seg3d_thin_out_mfma_packed_elems(int Cin) { return 2ll * (3 * Cin / 16) * 64 * 8; }

extern "C" int seg3d_pack_weights_thin_out_mfma(const float* w, void* wp_bf16, int A, int B, long long sa, long long sb,
                                                int flip, void* stream) {
  SEG3D_REQUIRE(w && wp_bf16 && seg3d_conv3d_k3_thin_out_mfma_supported(A, B),
                "seg3d_pack_weights_thin_out_mfma: need Cin in {16, 32} and Cout <= 3");
  const int total = (3 * A / 16) * 64 * 8;
  hipLaunchKernelGGL(pack_thin_out_mfma_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     reinterpret_cast<seg3d_bf16*>(wp_bf16), A, B, B <= 2 ? 2 : 3, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_thin_out_mfma");
  return SEG3D_OK;
}

#ifndef SEG3D_HEAD_RING
#define SEG3D_HEAD_RING 2
#endif
template <int CIN, int COUT>
__global__ __launch_bounds__(256, COUT == 2 ? 3 : 2) void conv3d_k3_thin_out_mfma_kernel(const seg3d_bf16* __restrict__ x,
                                                                                         const seg3d_bf16* __restrict__ wp,
                                                                                         const float* __restrict__ bias,
                                                                                         float* __restrict__ y,
                                                                                         float* __restrict__ stats, int N,
                                                                                         int D, int H, int W, int Cout,
                                                                                         int ntz, int nty, int ntx) {
  constexpr int KS = 3 * CIN / 16;             // K-steps of 16: ks = kx * KC + j, channels 16 j + 8 half ..
  constexpr int KC = CIN / 16;
  constexpr int NCOL = 9 * COUT;               // used rows of the 32-row MFMA result
  constexpr int STR = (NCOL + 3) & ~3;         // floats per voxel of a P plane (20 / 28)
  constexpr int ROW = TO_TX * STR + 4;         // floats per halo row (pad: the 8 ty rows of a read start 4 banks apart)
  constexpr int PLANE = TO_HY * ROW;           // one halo plane of one wave (12.7 / 17.7 KB)
  constexpr int NBLK = 5 * (TO_TZ + 2);        // row blocks of a tile
  constexpr int RING = SEG3D_HEAD_RING;        // row blocks in flight (registers)
  static_assert(NBLK % RING == 0, "the row blocks of a tile are walked in groups of RING");
  __shared__ __attribute__((aligned(16))) float pl[4 * PLANE];
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, lh = lane >> 5, r = lane & 31;
  const int ntiles = N * ntz * nty * ntx;
  const int wg = seg3d_xcd_tile(blockIdx.x, (ntiles + 3) >> 2);   // neighbouring tiles on one XCD
  if (wg < 0) return;
  int b = wg * 4 + wave;                       // the wave's tile (x fastest)
  if (b >= ntiles) return;
  int qd = seg3d_fdiv(b, 1.0f / (float)ntx);
  const int tix = b - qd * ntx; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)nty);
  const int tiy = b - qd * nty; b = qd;
  qd = seg3d_fdiv(b, 1.0f / (float)ntz);
  const int tiz = b - qd * ntz;
  const int n = qd;
  const int z0 = tiz * TO_TZ, y0 = tiy * TO_TY, x0 = tix * TO_TX;
  float* __restrict__ pw = pl + wave * PLANE;

  f32x4 whi[KS], wlo[KS];
#pragma unroll
  for (int ks = 0; ks < KS; ++ks) {
    whi[ks] = *reinterpret_cast<const f32x4*>(wp + ((i64)ks * 64 + lane) * 8);
    wlo[ks] = *reinterpret_cast<const f32x4*>(wp + ((i64)(KS + ks) * 64 + lane) * 8);
  }

  // row block i = 5 p + mb: halo plane p (gz = z0 + p - 1), halo rows 2 mb, 2 mb + 1; lane row r is the voxel
  // (row 2 mb + (r >> 4), x = r & 15), lane half lh its channels 16 j + 8 lh ..
  const int gx = x0 + (r & 15);
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  auto block_voxel = [&](int i, bool& rowok) {
    const int p = i / 5, mb = i - 5 * p;
    const int gz = z0 + p - 1;
    const int gy = y0 + 2 * mb + (r >> 4) - 1;
    rowok = gz >= 0 && gz < D && gy >= 0 && gy < H;
    return ((n * D + gz) * H + gy) * W + gx;
  };
  // operands of a row block: ONE load of the voxel's own channels (the kx = 1 K-steps); the kx = 0 / kx = 2 operands are
  // the same registers of the neighbouring lane (DPP row shift inside the 16-lane x row), and only the two end lanes of
  // a row need their outside neighbour (x0 - 1 / x0 + 16).  Loads are unconditional (an invalid voxel reads the tensor's
  // first bytes and is zeroed when it is consumed): with no branch around them the compiler can count them, so the wait
  // in front of a row block is a counted vmcnt that leaves the later blocks of the ring in flight.
  const int edge_dx = (r & 15) == 0 ? -1 : ((r & 15) == 15 ? 1 : 0);
  const int edge_gx = gx + edge_dx;
  const bool centre_in = gx < W, edge_in = edge_dx != 0 && edge_gx >= 0 && edge_gx < W;
  auto load_centre = [&](int vox, bool rowok, int j) {
    return *reinterpret_cast<const f32x4*>(x + ((rowok && centre_in) ? (i64)vox * CIN : (i64)0) + j * 16 + 8 * lh);
  };
  auto load_edge = [&](int vox, bool rowok, int j) {
    return *reinterpret_cast<const f32x4*>(x + ((rowok && edge_in) ? (i64)(vox + edge_dx) * CIN : (i64)0) + j * 16 + 8 * lh);
  };
  auto shift_x = [&](const f32x4& cen, const f32x4& edge, f32x4& left, f32x4& right) {
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int cv = __float_as_int(cen[d]), ev = __float_as_int(edge[d]);
      left[d] = __int_as_float(__builtin_amdgcn_update_dpp(ev, cv, 0x111, 0xf, 0xf, false));    // row_shr:1: lane i <- i - 1
      right[d] = __int_as_float(__builtin_amdgcn_update_dpp(ev, cv, 0x101, 0xf, 0xf, false));   // row_shl:1: lane i <- i + 1
    }
  };

  // outputs of a lane: (ty, x = 2 xp, 2 xp + 1) of the planes in flight; acc[k] is output plane p - k (taps kz = k)
  const int ty = lane >> 3, xp = lane & 7;
  float acc[3][2][COUT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < COUT; ++c) acc[k][o][c] = 0.f;
  float bv[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) bv[c] = (bias && c < Cout) ? bias[c] : 0.f;
  float s[2] = {0.f, 0.f};
  const int oy = y0 + ty, ox = x0 + 2 * xp;

  f32x4 cc[RING][KC], ee[RING][KC];
#pragma unroll
  for (int sl = 0; sl < RING; ++sl) {
    bool ok0;
    const int v0 = block_voxel(sl, ok0);
#pragma unroll
    for (int j = 0; j < KC; ++j) {
      cc[sl][j] = load_centre(v0, ok0, j);
      ee[sl][j] = load_edge(v0, ok0, j);
    }
  }
#pragma unroll 1
  for (int i0 = 0; i0 < NBLK; i0 += RING) {
#pragma unroll
    for (int sl = 0; sl < RING; ++sl) {
      const int i = i0 + sl;
      const int p = i / 5, mb = i - 5 * p;
      bool cok, nok;
      block_voxel(i, cok);
      const int nvox = block_voxel(i + RING < NBLK ? i + RING : i, nok);
      f32x4 a[KS];
#pragma unroll
      for (int j = 0; j < KC; ++j) {
        const f32x4 cen = (cok && centre_in) ? cc[sl][j] : zero4;
        const f32x4 edg = (cok && edge_in) ? ee[sl][j] : zero4;
        shift_x(cen, edg, a[j], a[2 * KC + j]);
        a[KC + j] = cen;
      }
#pragma unroll
      for (int j = 0; j < KC; ++j) {   // block i + RING (the last RING blocks re-request themselves: no branch around a load)
        cc[sl][j] = load_centre(nvox, nok, j);
        ee[sl][j] = load_edge(nvox, nok, j);
      }
      f32x16 c;
#pragma unroll
      for (int v = 0; v < 16; ++v) c[v] = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, whi[ks]),
                                                    __builtin_bit_cast(to_bf16x8, a[ks]), c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, wlo[ks]),
                                                    __builtin_bit_cast(to_bf16x8, a[ks]), c, 0, 0, 0);
      }
      // result register 4 g + j is row 8 g + 4 lh + j (the (kz, ky, co) column of P) of voxel r
      float* dst = pw + (2 * mb + (r >> 4)) * ROW + (r & 15) * STR;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        if (8 * g < NCOL) {
          const int row0 = 8 * g + 4 * lh;
          if (row0 < NCOL) {
            const f32x4 v = {c[4 * g], c[4 * g + 1], c[4 * g + 2], c[4 * g + 3]};
            *reinterpret_cast<f32x4*>(dst + row0) = v;
          }
        }
      }
      if (mb == 4) {   // halo plane p is complete in LDS (wave-private: ordering inside the wave is all that is needed)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int kz = 0; kz < 3; ++kz) {
          if (p - kz >= 0 && p - kz < TO_TZ) {   // wave-uniform
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
              const float* src = pw + (ty + ky) * ROW + (2 * xp) * STR + (kz * 3 + ky) * COUT;
#pragma unroll
              for (int o = 0; o < 2; ++o)
#pragma unroll
                for (int cq = 0; cq < COUT; ++cq) acc[kz][o][cq] += src[o * STR + cq];
            }
          }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int oz = z0 + p - 2;   // output plane p - 2 has all three kz taps now
        if (p >= 2 && oz < D && oy < H) {
          float* yp = y + ((((i64)n * D + oz) * H + oy) * W + ox) * Cout;
          float val[2][COUT];
#pragma unroll
          for (int o = 0; o < 2; ++o)
#pragma unroll
            for (int cq = 0; cq < COUT; ++cq) val[o][cq] = acc[2][o][cq] + bv[cq];
          if (COUT == 2 && Cout == 2 && ox + 1 < W && (W & 1) == 0) {   // 16 bytes, aligned (even row length, even x)
            const f32x4 v = {val[0][0], val[0][COUT - 1], val[1][0], val[1][COUT - 1]};
            *reinterpret_cast<f32x4*>(yp) = v;
#pragma unroll
            for (int o = 0; o < 2; ++o)
#pragma unroll
              for (int cq = 0; cq < COUT; ++cq) {
                s[0] += val[o][cq];
                s[1] += val[o][cq] * val[o][cq];
              }
          } else {
#pragma unroll
            for (int o = 0; o < 2; ++o)
              if (ox + o < W) {
#pragma unroll
                for (int cq = 0; cq < COUT; ++cq)
                  if (cq < Cout) {
                    yp[o * Cout + cq] = val[o][cq];
                    s[0] += val[o][cq];
                    s[1] += val[o][cq] * val[o][cq];
                  }
              }
          }
        }
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int cq = 0; cq < COUT; ++cq) {
            acc[2][o][cq] = acc[1][o][cq];
            acc[1][o][cq] = acc[0][o][cq];
            acc[0][o][cq] = 0.f;
          }
      }
    }
  }

  if (stats) {   // one slot per tile = per wave
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      s[0] += __shfl_xor(s[0], m, 64);
      s[1] += __shfl_xor(s[1], m, 64);
    }
    if (lane == 0) {
      const int tiles_per_sample = ntz * nty * ntx;
      const int tile = (tiz * nty + tiy) * ntx + tix;
      float* dst = stats + ((i64)n * tiles_per_sample + tile) * 2;
      dst[0] = s[0];
      dst[1] = s[1];
    }
  }
}

// x bf16 [N][D][H][W][Cin], wp = seg3d_pack_weights_thin_out_mfma, y fp32 [N][D][H][W][Cout]; stats as the VALU kernel
extern "C" int seg3d_conv3d_k3_thin_out_mfma_fwd(const void* x_bf16, const void* wp_bf16, const float* bias, float* y,
                                                 float* stats, int N, int D, int H, int W, int Cin, int Cout, void* stream) {
  SEG3D_REQUIRE(x_bf16 && wp_bf16 && y, "seg3d_conv3d_k3_thin_out_mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "seg3d_conv3d_k3_thin_out_mfma_fwd: bad dims");
  SEG3D_REQUIRE(seg3d_conv3d_k3_thin_out_mfma_supported(Cin, Cout),
                "seg3d_conv3d_k3_thin_out_mfma_fwd: need Cin in {16, 32} and Cout <= 3");
  SEG3D_REQUIRE((i64)N * D * H * W * Cin < (1ll << 31), "seg3d_conv3d_k3_thin_out_mfma_fwd: tensor exceeds 2^31 elements");
  const int ntz = seg3d_cdiv(D, TO_TZ), nty = seg3d_cdiv(H, TO_TY), ntx = seg3d_cdiv(W, TO_TX);
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "seg3d_conv3d_k3_thin_out_mfma_fwd: more than 2^22 tiles");
  dim3 grid((unsigned)seg3d_xcd_grid((N * ntz * nty * ntx + 3) / 4));   // one tile per wave, four waves per workgroup
  hipStream_t s = (hipStream_t)stream;
  const seg3d_bf16* xp = reinterpret_cast<const seg3d_bf16*>(x_bf16);
  const seg3d_bf16* wq = reinterpret_cast<const seg3d_bf16*>(wp_bf16);
#define SEG3D_TO_MFMA(CI, CO)                                                                                              \
  hipLaunchKernelGGL((conv3d_k3_thin_out_mfma_kernel<CI, CO>), grid, dim3(256), 0, s, xp, wq, bias, y, stats, N, D, H, W, \
                     Cout, ntz, nty, ntx)
  if (Cin == 32 && Cout <= 2) SEG3D_TO_MFMA(32, 2);
  else if (Cin == 32) SEG3D_TO_MFMA(32, 3);
  else if (Cout <= 2) SEG3D_TO_MFMA(16, 2);
  else SEG3D_TO_MFMA(16, 3);
#undef SEG3D_TO_MFMA
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_out_mfma_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// thin weight gradient: G[t][ct][cf] = sum_u fat[u][cf] * thin[u + off(t)][ct]
// ---------------------------------------------------------------------------------------------------------------
// FAT_BF (bf16 mode, head weight gradient): the fat operand (the unit's bf16 input) is widened at the LDS store
// Round 2: fp32 MFMA and the VALU share one datapath, and this kernel spent ~820 vector instructions per
// tile and wave beside 32..64 MFMAs (index arithmetic of the staging loads and of the K loop).  Now every per-lane quantity
// that does not depend on the tile is computed once: halo / tile coordinates of the staged entries (packed), LDS offsets of
// the K steps (the loop is fully unrolled: the offsets are immediates), and rows beyond 27 CT read tap 0 instead of being
// masked by a multiply (the reduce kernel never looks at them).
template <int CT, bool FAT_BF>
__device__ __forceinline__ void k3_thin_wgrad_body(const float* __restrict__ thin, const void* __restrict__ fat,
                                                                 float* __restrict__ part, int N, int D, int H, int W,
                                                                 int CF, int ntz, int nty, int ntx, int ntiles) {
  constexpr int ROWS = 27 * CT;
  constexpr int RB = (ROWS + 31) / 32;
  __shared__ __attribute__((aligned(16))) float fs[TH_MT * 32];         // fat tile [256][32]; reused for the wave reduce
  __shared__ __attribute__((aligned(16))) float ts[TH_NV * CT + 4];     // thin halo tile [600][CT]
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int cf0 = blockIdx.y * 32;
  // per-lane row descriptors: LDS offset of row i = (tap, thin channel) relative to the voxel's own halo position
  int roff[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int i = rb * 32 + li;
    if (i >= ROWS) i = 0;
    const int t = i / CT, a = i % CT;
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    roff[rb] = ((kz * TH_HY + ky) * TH_HX + kx) * CT + a;
  }
  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  const int q = tid & 7;
  const bool fq_ok = cf0 + 4 * q < CF;

  // staged entries of this thread (tile-invariant): thin halo entry k -> (hz, hy, hx, channel), fat entry k -> (tz, ty, tx);
  // trel / frel = element offset from the tile's first voxel; tface = halo faces the entry lies on (bit 0..5 = z lo, z hi,
  // y lo, y hi, x lo, x hi; bit 6 = no such entry).  When the level is a whole number of tiles (`regular`) an entry is
  // padding exactly when one of its faces is outside the volume for this tile: no per-entry coordinate arithmetic.
  constexpr int TE = (TH_NV * CT + 255) / 256, FE = (TH_MT * 8) / 256;
  static_assert(TE + FE <= 32, "okmask is 32 bits");
  const bool regular = (D % TH_TZ) == 0 && (H % TH_TY) == 0 && (W % TH_TX) == 0;
  int tpos[TE], fpos[FE], trel[TE], frel[FE], tface[TE];   // pos: hz << 20 | hy << 10 | hx (| channel << 28), -1: no entry
#pragma unroll
  for (int k = 0; k < TE; ++k) {
    const int e = tid + k * 256;
    const int v = e / CT, a = e % CT;
    const int hx = v % TH_HX;
    const int t = v / TH_HX;
    const int hy = t % TH_HY, hz = t / TH_HY;
    const bool have = e < TH_NV * CT;
    tpos[k] = have ? ((a << 28) | (hz << 20) | (hy << 10) | hx) : -1;
    trel[k] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * CT + a;
    tface[k] = have ? ((hz == 0 ? 1 : 0) | (hz == TH_TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TH_TY + 1 ? 8 : 0) |
                       (hx == 0 ? 16 : 0) | (hx == TH_TX + 1 ? 32 : 0)) : 64;
  }
#pragma unroll
  for (int k = 0; k < FE; ++k) {
    const int v = (tid + k * 256) >> 3;
    const int t = v / TH_TX;
    fpos[k] = ((t / TH_TY) << 20) | ((t % TH_TY) << 10) | (v % TH_TX);
    frel[k] = (((t / TH_TY) * H + (t % TH_TY)) * W + (v % TH_TX)) * CF + cf0 + 4 * q;
  }
  // K steps: wave w takes voxels [64 w, 64 w + 64); step kp, lane half lh is voxel 64 w + 2 kp + lh = (tz = w,
  // ty = kp >> 2, tx = 2 (kp & 3) + lh): everything but the lane's own (lh, li) part is a compile-time offset
  const int ts_lane = (wave * TH_HY * TH_HX + lh) * CT;
  const int fs_lane = (wave * 64 + lh) * 32 + li;

  // register-prefetch pipeline over the tile loop (the next tile's loads stay in flight behind the MFMA block of the
  // current one)
  float tst[TE];
  typename Seg3dQuad<FAT_BF>::raw fst[FE];
  unsigned okmask = 0;  // zero-select deferred to store_tile: a select right at the load would serialise the loads
  auto load_tile = [&](int tile) {   // tile is wave-uniform: the decode below is scalar work
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * TH_TZ, y0 = tiy * TH_TY, x0 = tix * TH_TX;
    const i64 vox0 = (i64)((n * D + z0) * H + y0) * W + x0;     // the tile's first voxel
    okmask = 0;
    if (regular) {
      const int faces = 64 | (z0 == 0 ? 1 : 0) | (z0 + TH_TZ >= D ? 2 : 0) | (y0 == 0 ? 4 : 0) | (y0 + TH_TY >= H ? 8 : 0) |
                        (x0 == 0 ? 16 : 0) | (x0 + TH_TX >= W ? 32 : 0);
      const float* tbase = thin + vox0 * CT;
      const float* fbase = reinterpret_cast<const float*>(fat);   // element offsets below are in fat's own element type
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const bool ok = (tface[k] & faces) == 0;
        tst[k] = ok ? tbase[trel[k]] : thin[0];
        okmask |= (ok ? 1u : 0u) << k;
      }
      (void)fbase;
#pragma unroll
      for (int k = 0; k < FE; ++k) {
        fst[k] = Seg3dQuad<FAT_BF>::load(fat, fq_ok ? vox0 * CF + frel[k] : (i64)0);
        okmask |= (fq_ok ? 1u : 0u) << (TE + k);
      }
    } else {
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const int hz = (tpos[k] >> 20) & 255, hy = (tpos[k] >> 10) & 1023, hx = tpos[k] & 1023;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = tpos[k] >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        tst[k] = thin[ok ? vox0 * CT + trel[k] : (i64)0];
        okmask |= (ok ? 1u : 0u) << k;
      }
#pragma unroll
      for (int k = 0; k < FE; ++k) {
        const int tz = fpos[k] >> 20, ty = (fpos[k] >> 10) & 1023, tx = fpos[k] & 1023;
        const bool ok = fq_ok && z0 + tz < D && y0 + ty < H && x0 + tx < W;
        fst[k] = Seg3dQuad<FAT_BF>::load(fat, ok ? vox0 * CF + frel[k] : (i64)0);
        okmask |= (ok ? 1u : 0u) << (TE + k);
      }
    }
  };
  auto store_tile = [&]() {
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < TE; ++k) {
      const int e = tid + k * 256;
      if (e < TH_NV * CT) ts[e] = ((okmask >> k) & 1u) ? tst[k] : 0.f;
    }
#pragma unroll
    for (int k = 0; k < FE; ++k)
      *reinterpret_cast<f32x4*>(fs + ((tid + k * 256) >> 3) * 32 + 4 * q) =
          ((okmask >> (TE + k)) & 1u) ? Seg3dQuad<FAT_BF>::cvt(fst[k]) : zero;
  };
  const int tile0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x), tstep = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  if (tile0 < ntiles) load_tile(tile0);
  for (int tile = tile0; tile < ntiles; tile += tstep) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + tstep < ntiles) load_tile(tile + tstep);
#pragma unroll
    for (int kp = 0; kp < 32; ++kp) {
      const int ub = ts_lane + ((kp >> 2) * TH_HX + 2 * (kp & 3)) * CT;
      const float bvv = fs[fs_lane + kp * 64];
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) acc[rb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ts[ub + roff[rb]], bvv, acc[rb], 0, 0, 0);
    }
  }
  // reduce the 4 waves' accumulators through LDS (fixed order) and write this workgroup's slab [RB*32][32]
  float* dst = part + ((i64)blockIdx.x * gridDim.y + blockIdx.y) * RB * 1024;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) fs[wave * 1024 + thin_row(r, lh) * 32 + li] = acc[rb][r];
    __syncthreads();
    for (int k = tid; k < 1024; k += 256) dst[rb * 1024 + k] = (fs[k] + fs[1024 + k]) + (fs[2048 + k] + fs[3072 + k]);
  }
}

// bf16 mode, head weight gradient on the bf16 matrix cores: same tiles, slabs and reduce as above, but the voxel (K)
// dimension is fed 16 per v_mfma_f32_32x32x16_bf16 instead of 2 per fp32 MFMA (the fp32 version is bound by its 64
// MFMAs + 96 scalar LDS reads per wave and tile).  A lane's operand is 8 consecutive voxels = one x row of the 4 x 8 x 8
// tile, so both tiles sit in LDS with x fastest:
//   fat  (bf16 input):   fsT[cf][256 voxels], transposed in registers -- a thread loads the 8 voxels of one row for its 4
//                        channels and writes four 16-byte rows;
//   thin (fp32 dy):      three x-shifted planar copies tsP[kx][ct][halo row][8] so that every tap's row starts 16-byte
//                        aligned, as a bf16 hi + lo pair (two MFMAs into the same accumulator): the 2..5-channel tail
//                        keeps fp32-grade operands (2^-17), the fat operand is bf16 in memory already.
// Rows of the last row block beyond 27 CT read tap 0 again; the reduce kernel never looks at them.
#define THB_FS (TH_MT + 8)   // fsT row stride in bf16 (528 bytes: 16-byte reads of 16 lanes cover all banks)
template <int CT>
__global__ __launch_bounds__(256, 2) void k3_thin_wgrad_bf16_mfma_kernel(const float* __restrict__ thin,
                                                                          const seg3d_bf16* __restrict__ fat,
                                                                          float* __restrict__ part, int N, int D, int H, int W,
                                                                          int CF, int ntz, int nty, int ntx, int ntiles) {
  constexpr int ROWS = 27 * CT;
  constexpr int RB = (ROWS + 31) / 32;
  constexpr int HR = (TH_TZ + 2) * TH_HY;           // 60 halo rows
  constexpr int TSP = 3 * CT * HR * 8;              // bf16 elements of one (hi or lo) set of shifted planar copies
  constexpr int LDS_A = 32 * THB_FS * 2 + 2 * TSP * 2;   // bytes: fsT + tsP(hi, lo)
  constexpr int LDS_B = 4 * 1024 * 4;                    // bytes: the final 4-wave reduce
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[LDS_A > LDS_B ? LDS_A : LDS_B];
  seg3d_bf16* fsT = reinterpret_cast<seg3d_bf16*>(lds_raw);
  seg3d_bf16* tsP = fsT + 32 * THB_FS;
  float* red = reinterpret_cast<float*>(lds_raw);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int cf0 = blockIdx.y * 32;
  int roff[RB];   // start of the row's x row 0 in tsP (bf16 elements)
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    int i = rb * 32 + li;
    if (i >= ROWS) i = 0;
    const int t = i / CT, a = i % CT;
    const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
    roff[rb] = ((kx * CT + a) * HR + kz * TH_HY + ky) * 8;
  }
  f32x16 acc[RB];
#pragma unroll
  for (int rb = 0; rb < RB; ++rb)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[rb][r] = 0.f;
  const int q = tid & 7, frow = tid >> 3;           // fat: channels cf0 + 4 q .., x row frow = tz * 8 + ty, voxels tx = k
  const bool fq_ok = cf0 + 4 * q < CF;

  constexpr int TE = (TH_NV * CT + 255) / 256, FE = TH_TX;
  float tst[TE];
  uint2 fst[FE];
  static_assert(TE + FE <= 32, "okmask is 32 bits");
  unsigned okmask = 0;
  auto load_tile = [&](int tile) {
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * TH_TZ, y0 = tiy * TH_TY, x0 = tix * TH_TX;
    okmask = 0;
#pragma unroll
    for (int k = 0; k < TE; ++k) {
      const int e = tid + k * 256;
      const int v = e / CT, a = e % CT;
      const int hx = v % TH_HX;
      const int t = v / TH_HX;
      const int hy = t % TH_HY;
      const int hz = t / TH_HY;
      const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
      const bool ok = e < TH_NV * CT && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
      tst[k] = thin[ok ? ((((i64)n * D + gz) * H + gy) * W + gx) * CT + a : (i64)0];
      okmask |= (ok ? 1u : 0u) << k;
    }
    const int gz = z0 + (frow >> 3), gy = y0 + (frow & 7);
    const bool rok = fq_ok && gz < D && gy < H;
    const i64 rbase = ((((i64)n * D + gz) * H + gy) * W + x0) * CF + cf0 + 4 * q;
#pragma unroll
    for (int k = 0; k < FE; ++k) {
      const bool ok = rok && x0 + k < W;
      fst[k] = *reinterpret_cast<const uint2*>(fat + (ok ? rbase + (i64)k * CF : (i64)0));
      okmask |= (ok ? 1u : 0u) << (TE + k);
    }
  };
  auto store_tile = [&]() {
#pragma unroll
    for (int k = 0; k < TE; ++k) {
      const int e = tid + k * 256;
      if (e < TH_NV * CT) {
        const int v = e / CT, a = e % CT;
        const int hx = v % TH_HX, hrow = v / TH_HX;
        const float val = ((okmask >> k) & 1u) ? tst[k] : 0.f;
        const seg3d_bf16 hi = seg3d_f2bf(val);
        const seg3d_bf16 lo = seg3d_f2bf(val - seg3d_bf2f(hi));
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
          const int txp = hx - kx;
          if (txp >= 0 && txp < TH_TX) {
            const int idx = ((kx * CT + a) * HR + hrow) * 8 + txp;
            tsP[idx] = hi;
            tsP[TSP + idx] = lo;
          }
        }
      }
    }
    // 8 voxels x 4 channels -> 4 channels x 8 voxels (bf16 pairs): out[c] dword j = voxel 2j | voxel 2j + 1 << 16
    unsigned d0[FE], d1[FE];
#pragma unroll
    for (int k = 0; k < FE; ++k) {
      const bool ok = (okmask >> (TE + k)) & 1u;
      d0[k] = ok ? fst[k].x : 0u;
      d1[k] = ok ? fst[k].y : 0u;
    }
    uint4 o[4];
    unsigned* ow = reinterpret_cast<unsigned*>(o);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      ow[0 * 4 + j] = (d0[2 * j] & 0xffffu) | (d0[2 * j + 1] << 16);
      ow[1 * 4 + j] = (d0[2 * j] >> 16) | (d0[2 * j + 1] & 0xffff0000u);
      ow[2 * 4 + j] = (d1[2 * j] & 0xffffu) | (d1[2 * j + 1] << 16);
      ow[3 * 4 + j] = (d1[2 * j] >> 16) | (d1[2 * j + 1] & 0xffff0000u);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<uint4*>(fsT + (4 * q + c) * THB_FS + frow * 8) = o[c];
  };
  if ((int)blockIdx.x < ntiles) load_tile(blockIdx.x);
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    __syncthreads();
    store_tile();
    __syncthreads();
    if (tile + (int)gridDim.x < ntiles) load_tile(tile + gridDim.x);
    // K (voxels) is split over the waves: wave w takes plane tz = w, K-step s its x rows ty = 2 s + lh
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const int ty = 2 * st + lh;
      const f32x4 bq = *reinterpret_cast<const f32x4*>(fsT + li * THB_FS + (wave * 8 + ty) * 8);
      const int ub = (wave * TH_HY + ty) * 8;
#pragma unroll
      for (int rb = 0; rb < RB; ++rb) {
        const f32x4 ah = *reinterpret_cast<const f32x4*>(tsP + roff[rb] + ub);
        const f32x4 al = *reinterpret_cast<const f32x4*>(tsP + TSP + roff[rb] + ub);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, ah), __builtin_bit_cast(to_bf16x8, bq),
                                                          acc[rb], 0, 0, 0);
        acc[rb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(to_bf16x8, al), __builtin_bit_cast(to_bf16x8, bq),
                                                          acc[rb], 0, 0, 0);
      }
    }
  }
  // reduce the 4 waves' accumulators through LDS (fixed order) and write this workgroup's slab [RB*32][32]
  float* dst = part + ((i64)blockIdx.x * gridDim.y + blockIdx.y) * RB * 1024;
#pragma unroll
  for (int rb = 0; rb < RB; ++rb) {
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wave * 1024 + thin_row(r, lh) * 32 + li] = acc[rb][r];
    __syncthreads();
    for (int k = tid; k < 1024; k += 256) dst[rb * 1024 + k] = (red[k] + red[1024 + k]) + (red[2048 + k] + red[3072 + k]);
  }
}

// dw[ct*s_ct + cf*s_cf + (flip ? 26 - t : t)] = sum_slab part[slab][cfb][row = t*CT + ct][cf % 32]
// 4 outputs x 64 slab groups per workgroup (the output is tiny: the reduction over <= 512 slabs is the work, and it sits
// at the very end of backward where nothing overlaps it: 16 x 16 left each thread 32 dependent-latency loads), combined
// through LDS in a fixed order.
__global__ __launch_bounds__(256) void k3_thin_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                                     int slabs, int CT, int CF, int CFB, int RB, i64 s_ct,
                                                                     i64 s_cf, int flip, int accumulate) {
  __shared__ float red[256];
  const i64 total = (i64)27 * CT * CF;
  const i64 idx = (i64)blockIdx.x * 4 + (threadIdx.x & 3);
  const int g = threadIdx.x >> 2;
  float s = 0.f;
  int t = 0, ct = 0, cf = 0;
  if (idx < total) {
    cf = (int)(idx % CF);
    const int row = (int)(idx / CF);
    t = row / CT;
    ct = row % CT;
    const float* p = part + ((i64)(cf >> 5) * RB * 1024) + (i64)row * 32 + (cf & 31);
    for (int k = g; k < slabs; k += 64) s += p[(i64)k * CFB * RB * 1024];
  }
  red[threadIdx.x] = s;
  __syncthreads();
  if (g == 0 && idx < total) {
    float v = 0.f;
#pragma unroll
    for (int k = 0; k < 64; ++k) v += red[k * 4 + threadIdx.x];
    float* d = dw + ct * s_ct + cf * s_cf + (flip ? 26 - t : t);
    *d = accumulate ? *d + v : v;
  }
}

template <int CT>
__global__ __launch_bounds__(256, 2) void k3_thin_wgrad_kernel(const float* __restrict__ thin, const float* __restrict__ fat,
                                                                 float* __restrict__ part, int N, int D, int H, int W,
                                                                 int CF, int ntz, int nty, int ntx, int ntiles) {
  k3_thin_wgrad_body<CT, false>(thin, fat, part, N, D, H, W, CF, ntz, nty, ntx, ntiles);
}

static int thin_wgrad_slabs(int N, int D, int H, int W) {
  const int ntiles = N * seg3d_cdiv(D, TH_TZ) * seg3d_cdiv(H, TH_TY) * seg3d_cdiv(W, TH_TX);
  const int want = (ntiles + 7) / 8;  // >= 8 tiles per workgroup keeps the slab count (and the serial reduce) small
  return want < 512 ? (want < 1 ? 1 : want) : 512;
}

extern "C" long long seg3d_k3_thin_wgrad_workspace_floats(int N, int D, int H, int W, int CT, int CF) {
  const int RB = (27 * CT + 31) / 32, CFB = (CF + 31) / 32;
  return (long long)thin_wgrad_slabs(N, D, H, W) * CFB * RB * 1024;
}

template <int CT>
static void launch_thin_wgrad(const float* thin, const void* fat, int fat_bf16, float* part, int N, int D, int H, int W,
                              int CF, int slabs, hipStream_t s) {
  const int ntz = seg3d_cdiv(D, TH_TZ), nty = seg3d_cdiv(H, TH_TY), ntx = seg3d_cdiv(W, TH_TX);
  if (fat_bf16)
    hipLaunchKernelGGL((k3_thin_wgrad_bf16_mfma_kernel<CT>), dim3(slabs, (CF + 31) / 32), dim3(256), 0, s, thin,
                       reinterpret_cast<const seg3d_bf16*>(fat), part, N, D, H, W, CF, ntz, nty, ntx, N * ntz * nty * ntx);
  else
    hipLaunchKernelGGL((k3_thin_wgrad_kernel<CT>), dim3(slabs, (CF + 31) / 32), dim3(256), 0, s, thin,
                       reinterpret_cast<const float*>(fat), part, N, D, H, W, CF, ntz, nty, ntx, N * ntz * nty * ntx);
}

// thin [N][D][H][W][CT] (CT <= 8), fat [N][D][H][W][CF] (CF % 4 == 0); dw[ct*s_ct + cf*s_cf + tap]
// stem wgrad: thin = x, fat = dy, s_ct = 27, s_cf = Cin*27, flip = 0;  head wgrad: thin = dy, fat = x,
// s_ct = Cin*27, s_cf = 27, flip = 1.
static int thin_wgrad_launch(const float* thin, const void* fat, int fat_bf16, float* dw, float* workspace, int N, int D,
                             int H, int W, int CT, int CF, long long s_ct, long long s_cf, int flip, int accumulate,
                             void* stream);

extern "C" int seg3d_k3_thin_wgrad(const float* thin, const float* fat, float* dw, float* workspace, int N, int D, int H,
                                   int W, int CT, int CF, long long s_ct, long long s_cf, int flip, int accumulate,
                                   void* stream) {
  return thin_wgrad_launch(thin, fat, 0, dw, workspace, N, D, H, W, CT, CF, s_ct, s_cf, flip, accumulate, stream);
}

// bf16 mode: the fat operand is bf16 (head weight gradient: thin = fp32 dy of the 2..5-channel head, fat = bf16 input)
extern "C" int seg3d_k3_thin_wgrad_fatbf16(const float* thin, const void* fat_bf16, float* dw, float* workspace, int N, int D,
                                           int H, int W, int CT, int CF, long long s_ct, long long s_cf, int flip,
                                           int accumulate, void* stream) {
  return thin_wgrad_launch(thin, fat_bf16, 1, dw, workspace, N, D, H, W, CT, CF, s_ct, s_cf, flip, accumulate, stream);
}

static int thin_wgrad_launch(const float* thin, const void* fat, int fat_bf16, float* dw, float* workspace, int N, int D,
                             int H, int W, int CT, int CF, long long s_ct, long long s_cf, int flip, int accumulate,
                             void* stream) {
  SEG3D_REQUIRE(thin && fat && dw && workspace, "seg3d_k3_thin_wgrad: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "seg3d_k3_thin_wgrad: bad dims");
  SEG3D_REQUIRE(CT >= 1 && CT <= 8 && CF > 0 && (CF % 4) == 0, "seg3d_k3_thin_wgrad: need 1 <= CT <= 8 and CF %% 4 == 0");
  const int slabs = thin_wgrad_slabs(N, D, H, W);
  const int RB = (27 * CT + 31) / 32, CFB = (CF + 31) / 32;
  hipStream_t s = (hipStream_t)stream;
  switch (CT) {
    case 1: launch_thin_wgrad<1>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 2: launch_thin_wgrad<2>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 3: launch_thin_wgrad<3>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 4: launch_thin_wgrad<4>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 5: launch_thin_wgrad<5>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 6: launch_thin_wgrad<6>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    case 7: launch_thin_wgrad<7>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
    default: launch_thin_wgrad<8>(thin, fat, fat_bf16, workspace, N, D, H, W, CF, slabs, s); break;
  }
  SEG3D_LAUNCH_CHECK("seg3d_k3_thin_wgrad");
  const i64 total = (i64)27 * CT * CF;
  hipLaunchKernelGGL(k3_thin_wgrad_reduce_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, workspace, dw, slabs, CT, CF,
                     CFB, RB, (i64)s_ct, (i64)s_cf, flip, accumulate);
  SEG3D_LAUNCH_CHECK("seg3d_k3_thin_wgrad(reduce)");
  return SEG3D_OK;
}
