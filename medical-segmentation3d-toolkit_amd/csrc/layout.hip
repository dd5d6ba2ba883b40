// layout.hip -- NCDHW <-> NDHWC conversion, channel-slice copies and weight packers.
//
// The reference keeps tensors NCDHW (utils/image_tools.py:274-294 builds [C,z,y,x] from sitk arrays) and
// conv weights as [Cout,Cin,kD,kH,kW] (ConvTranspose3d: [Cin,Cout,kD,kH,kW]; network/module/vnet_upblock.py:11).
// The HIP engine computes in NDHWC with tap-major packed weights; these kernels are the bridges. They are
// HBM-bound byte movers: one read + one write per element, 16-byte accesses on the contiguous side.
#include "seg3d_common.h"
#include "seg3d_hip.h"

// ---- NCDHW -> NDHWC : out[n][s][c] = in[n][c][s] ------------------------------------------------
// A 64(s) x 16(c)-ish LDS transpose would be the classic form; C is tiny here (1..5 at the API edge), so each
// thread handles one voxel and loops over channels: reads are coalesced per channel plane, writes are C*4 B per lane.
__global__ __launch_bounds__(256) void ncdhw_to_ndhwc_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int C, i64 S, i64 total_vox) {
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total_vox; v += (i64)gridDim.x * 256) {
    i64 n = v / S, s = v - n * S;
    const float* src = in + n * C * S + s;
    float* dst = out + v * C;
    for (int c = 0; c < C; ++c) dst[c] = src[(i64)c * S];
  }
}

__global__ __launch_bounds__(256) void ndhwc_to_ncdhw_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                               int C, i64 S, i64 total_vox) {
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < total_vox; v += (i64)gridDim.x * 256) {
    i64 n = v / S, s = v - n * S;
    const float* src = in + v * C;
    float* dst = out + n * C * S + s;
    for (int c = 0; c < C; ++c) dst[(i64)c * S] = src[c];
  }
}

extern "C" int seg3d_ncdhw_to_ndhwc(const float* in, float* out, int N, int C, long long S, void* stream) {
  SEG3D_REQUIRE(in && out && N > 0 && C > 0 && S > 0, "seg3d_ncdhw_to_ndhwc: bad arguments");
  i64 total = (i64)N * S;
  hipLaunchKernelGGL(ncdhw_to_ndhwc_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, C, (i64)S, total);
  SEG3D_LAUNCH_CHECK("seg3d_ncdhw_to_ndhwc");
  return SEG3D_OK;
}

extern "C" int seg3d_ndhwc_to_ncdhw(const float* in, float* out, int N, int C, long long S, void* stream) {
  SEG3D_REQUIRE(in && out && N > 0 && C > 0 && S > 0, "seg3d_ndhwc_to_ncdhw: bad arguments");
  i64 total = (i64)N * S;
  hipLaunchKernelGGL(ndhwc_to_ncdhw_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, in,
                     out, C, (i64)S, total);
  SEG3D_LAUNCH_CHECK("seg3d_ndhwc_to_ncdhw");
  return SEG3D_OK;
}

// ---- channel-slice copy: dst[v][dst_off + c] = src[v][src_off + c], c < C -----------------------
// Used for torch.cat((up, skip), 1) (network/module/vnet_upblock.py:21) and its backward split.
__global__ __launch_bounds__(256) void copy_channels_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                              i64 nvox, int C, int src_ld, int src_off, int dst_ld,
                                                              int dst_off) {
  if ((C & 3) == 0 && (src_ld & 3) == 0 && (dst_ld & 3) == 0 && (src_off & 3) == 0 && (dst_off & 3) == 0) {
    const int CQ = C >> 2;
    i64 total = nvox * CQ;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
      i64 v = idx / CQ;
      int q = (int)(idx - v * CQ);
      float4 t = *reinterpret_cast<const float4*>(src + v * src_ld + src_off + 4 * q);
      *reinterpret_cast<float4*>(dst + v * dst_ld + dst_off + 4 * q) = t;
    }
  } else {
    i64 total = nvox * C;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
      i64 v = idx / C;
      int c = (int)(idx - v * C);
      dst[v * dst_ld + dst_off + c] = src[v * src_ld + src_off + c];
    }
  }
}

extern "C" int seg3d_copy_channels(const float* src, float* dst, long long nvox, int C, int src_ld, int src_off,
                                   int dst_ld, int dst_off, void* stream) {
  SEG3D_REQUIRE(src && dst && nvox > 0 && C > 0, "seg3d_copy_channels: bad arguments");
  SEG3D_REQUIRE(src_off + C <= src_ld && dst_off + C <= dst_ld, "seg3d_copy_channels: slice exceeds row pitch");
  i64 work = (i64)nvox * ((C & 3) ? C : C / 4);
  hipLaunchKernelGGL(copy_channels_kernel, dim3(seg3d_ew_grid(work, 256)), dim3(256), 0, (hipStream_t)stream, src, dst,
                     (i64)nvox, C, src_ld, src_off, dst_ld, dst_off);
  SEG3D_LAUNCH_CHECK("seg3d_copy_channels");
  return SEG3D_OK;
}

// ---- weight packers ------------------------------------------------------------------------------
// A weight tensor is described as W(a, b, t) = w[a*sa + b*sb + t] with a = reduction channel (A of them),
// b = output channel (B of them), t = tap (T of them); `flip` reads tap T-1-t (dgrad of a 3x3x3 conv).
//   Conv3d fwd      w[co][ci][t] : a=ci sa=T       b=co sb=Cin*T
//   Conv3d k3 dgrad             : a=co sa=Cin*T   b=ci sb=T        flip=1
//   ConvT fwd       w[ci][co][t] : a=ci sa=Cout*T  b=co sb=T
// tap-major pack (direct kernels):  wp[t][a][bp]            bp padded to BP (multiple of 4), zeros beyond B
__global__ __launch_bounds__(256) void pack_tapmajor_kernel(const float* __restrict__ w, float* __restrict__ wp, int A,
                                                              int B, int BP, int T, i64 sa, i64 sb, int flip) {
  i64 total = (i64)T * A * BP;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    int b = (int)(idx % BP);
    i64 r = idx / BP;
    int a = (int)(r % A);
    int t = (int)(r / A);
    float v = 0.f;
    if (b < B) v = w[a * sa + b * sb + (flip ? T - 1 - t : t)];
    wp[idx] = v;
  }
}

extern "C" int seg3d_pack_weights_tapmajor(const float* w, float* wp, int A, int B, int BP, int T, long long sa,
                                           long long sb, int flip, void* stream) {
  SEG3D_REQUIRE(w && wp && A > 0 && B > 0 && T > 0 && BP >= B && (BP % 4) == 0, "seg3d_pack_weights_tapmajor: bad arguments");
  i64 total = (i64)T * A * BP;
  hipLaunchKernelGGL(pack_tapmajor_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, A,
                     B, BP, T, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_tapmajor");
  return SEG3D_OK;
}

// Winograd F(2, 3) along x (conv_wino.hip): T = 36 selects the transformed image of a 27-tap weight,
// t = (kz * 3 + ky) * 4 + p with U_0 = g0, U_1 = (g0 + g1 + g2) / 2, U_2 = (g0 - g1 + g2) / 2, U_3 = g2 over the kx taps
#define SEG3D_WINO_T 36
__device__ __forceinline__ float seg3d_wino_u(float g0, float g1, float g2, int p) {
  return p == 0 ? g0 : p == 1 ? 0.5f * ((g0 + g2) + g1) : p == 2 ? 0.5f * ((g0 + g2) - g1) : g2;
}
// Winograd F(2x2, 3x3) over (y, x) (conv_wino2d.hip): T = 48 selects U = G g G^T per kz, t = kz * 16 + py * 4 + px;
// g = the 27 taps of one (a, b) with stride 1 (flip: read back to front).
// The T = 48 image has its OWN order inside an (32 x 8) chunk (round 4): wp[bb][ab][h 2][g 12][r 4][j 32][s 4] =
// U(a = ab*8 + h*4 + r, b = bb*32 + j, t = seg3d_w2_step_t(4 g + s)) -- one half h = the 24-KB LDS image of a 4-channel K
// chunk of the forward kernels, in which a lane (output channel j, K index r) finds the A operands of the four consecutive
// steps 4 g .. 4 g + 3 in one aligned 16-byte word (ds_read_b128, conflict-free: tools/lds_bank_sim.py).
#define SEG3D_WINO2D_T 48
struct Seg3dW2StepTable {   // t48 of every step of the chunk's step order, built at compile time
  unsigned char t[48];
  constexpr Seg3dW2StepTable() : t() {
    for (int s = 0; s < 48; ++s) t[s] = (unsigned char)seg3d_w2_step_t(s);
  }
};
__device__ const Seg3dW2StepTable seg3d_w2_step_table;
__device__ __forceinline__ void seg3d_wino2d_decode(int i, int& h, int& r, int& j, int& t48) {   // i = offset inside a chunk's 12288 floats
  const int s4 = i & 3, g24 = i >> 9;         // g24 = h * 12 + g
  j = (i >> 2) & 31;
  r = (i >> 7) & 3;
  h = g24 >= 12 ? 1 : 0;
  t48 = seg3d_w2_step_table.t[4 * (g24 - 12 * h) + s4];   // (a table: the div / mod arithmetic of the step order per element made the multi-pack 17 % slower)
}
__device__ __forceinline__ float seg3d_wino2d_u(const float* g, int flip, int t48) {
  const int kz = t48 >> 4, py = (t48 >> 2) & 3, px = t48 & 3;
  float r[3];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int i = (kz * 3 + ky) * 3;
    r[ky] = seg3d_wino_u(g[flip ? 26 - i : i], g[flip ? 25 - i : i + 1], g[flip ? 24 - i : i + 2], px);
  }
  return seg3d_wino_u(r[0], r[1], r[2], py);
}

// MFMA pack (conv_mfma.hip): wp[bb][ab][t][h][j][r] = W(a = ab*8 + h*4 + r, b = bb*32 + j, t), zero padded.
// One (bb, ab) chunk = T*2*32*4 floats is exactly the LDS image a workgroup stages per K-chunk, so staging is
// a straight 16-byte copy.
__global__ __launch_bounds__(256) void pack_mfma_kernel(const float* __restrict__ w, float* __restrict__ wp, int A,
                                                          int B, int AB, int BB, int T, i64 sa, i64 sb, int flip) {
  i64 total = (i64)BB * AB * T * 256;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    int r = (int)(idx & 3);
    int j = (int)((idx >> 2) & 31);
    int h = (int)((idx >> 7) & 1);
    i64 rest = idx >> 8;
    int t = (int)(rest % T);
    rest /= T;
    int ab = (int)(rest % AB);
    int bb = (int)(rest / AB);
    if (T == SEG3D_WINO2D_T) seg3d_wino2d_decode((int)(idx % (48 * 256)), h, r, j, t);
    int a = ab * 8 + h * 4 + r, b = bb * 32 + j;
    float v = 0.f;
    if (a < A && b < B) {
      if (T == SEG3D_WINO_T) {   // t = (kz * 3 + ky) * 4 + p: Winograd F(2, 3) transform along kx of a 27-tap weight
        const int t9 = t >> 2;
        const float* g = w + a * sa + b * sb;
        const float g0 = g[flip ? 26 - 3 * t9 : 3 * t9], g1 = g[flip ? 25 - 3 * t9 : 3 * t9 + 1], g2 = g[flip ? 24 - 3 * t9 : 3 * t9 + 2];
        v = seg3d_wino_u(g0, g1, g2, t & 3);
      } else if (T == SEG3D_WINO2D_T) {
        v = seg3d_wino2d_u(w + a * sa + b * sb, flip, t);
      } else {
        v = w[a * sa + b * sb + (flip ? T - 1 - t : t)];
      }
    }
    wp[idx] = v;
  }
}

extern "C" int seg3d_pack_weights_mfma(const float* w, float* wp, int A, int B, int T, long long sa, long long sb,
                                       int flip, void* stream) {
  SEG3D_REQUIRE(w && wp && A > 0 && B > 0 && T > 0, "seg3d_pack_weights_mfma: bad arguments");
  int AB = (A + 7) / 8, BB = (B + 31) / 32;
  i64 total = (i64)BB * AB * T * 256;
  hipLaunchKernelGGL(pack_mfma_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wp, A, B,
                     AB, BB, T, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_mfma");
  return SEG3D_OK;
}

// bf16 MFMA pack (bf16 conv path): wp[bb][ab][t][h][j][r] = bf16(W(a = ab*16 + h*8 + r, b = bb*32 + j, t)), zero padded:
// the same 1-KiB-per-tap LDS image as the fp32 pack with 8 bf16 channels where that has 4 floats.
__global__ __launch_bounds__(256) void pack_mfma_bf16_kernel(const float* __restrict__ w, seg3d_bf16* __restrict__ wp,
                                                               int A, int B, int AB, int BB, int T, i64 sa, i64 sb,
                                                               int flip) {
  i64 total = (i64)BB * AB * T * 512;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    int r = (int)(idx & 7);
    int j = (int)((idx >> 3) & 31);
    int h = (int)((idx >> 8) & 1);
    i64 rest = idx >> 9;
    int t = (int)(rest % T);
    rest /= T;
    int ab = (int)(rest % AB);
    int bb = (int)(rest / AB);
    int a = ab * 16 + h * 8 + r, b = bb * 32 + j;
    float v = 0.f;
    if (a < A && b < B) v = w[a * sa + b * sb + (flip ? T - 1 - t : t)];
    wp[idx] = seg3d_f2bf(v);
  }
}

extern "C" long long seg3d_packed_mfma_bf16_elems(int A, int B, int T) {
  if (T == SEG3D_WINO_T || T == SEG3D_WINO2D_T) return -1;   // no bf16 Winograd image: callers must not size a buffer for one
  return (long long)((B + 31) / 32) * ((A + 15) / 16) * T * 512;
}

extern "C" int seg3d_pack_weights_mfma_bf16(const float* w, void* wp, int A, int B, int T, long long sa, long long sb,
                                            int flip, void* stream) {
  SEG3D_REQUIRE(w && wp && A > 0 && B > 0 && T > 0, "seg3d_pack_weights_mfma_bf16: bad arguments");
  SEG3D_REQUIRE(T != SEG3D_WINO_T && T != SEG3D_WINO2D_T,
                "seg3d_pack_weights_mfma_bf16: T = 36 / 48 (Winograd images) exist in fp32 only");
  int AB = (A + 15) / 16, BB = (B + 31) / 32;
  i64 total = (i64)BB * AB * T * 512;
  hipLaunchKernelGGL(pack_mfma_bf16_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w,
                     (seg3d_bf16*)wp, A, B, AB, BB, T, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_mfma_bf16");
  return SEG3D_OK;
}

// fp32 <-> bf16 (round to nearest even) over n elements, n % 4 == 0 handled 4 per thread, tail by the last threads
__global__ __launch_bounds__(256) void f32_to_bf16_kernel(const float* __restrict__ src, seg3d_bf16* __restrict__ dst,
                                                            i64 n) {
  const i64 n4 = n >> 2;
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(src)[i];
    uint2 o;
    o.x = seg3d_pack2bf(v.x, v.y);
    o.y = seg3d_pack2bf(v.z, v.w);
    reinterpret_cast<uint2*>(dst)[i] = o;
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) dst[(n4 << 2) + threadIdx.x] = seg3d_f2bf(src[(n4 << 2) + threadIdx.x]);
}

__global__ __launch_bounds__(256) void bf16_to_f32_kernel(const seg3d_bf16* __restrict__ src, float* __restrict__ dst,
                                                            i64 n) {
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) dst[i] = seg3d_bf2f(src[i]);
}

extern "C" int seg3d_f32_to_bf16(const float* src, void* dst, long long n, void* stream) {
  SEG3D_REQUIRE(src && dst && n > 0, "seg3d_f32_to_bf16: bad arguments");
  SEG3D_REQUIRE((((uintptr_t)src) & 15) == 0 && (((uintptr_t)dst) & 7) == 0, "seg3d_f32_to_bf16: unaligned buffers");
  hipLaunchKernelGGL(f32_to_bf16_kernel, dim3(seg3d_ew_grid((n + 3) / 4, 256)), dim3(256), 0, (hipStream_t)stream, src,
                     (seg3d_bf16*)dst, (i64)n);
  SEG3D_LAUNCH_CHECK("seg3d_f32_to_bf16");
  return SEG3D_OK;
}

extern "C" int seg3d_bf16_to_f32(const void* src, float* dst, long long n, void* stream) {
  SEG3D_REQUIRE(src && dst && n > 0, "seg3d_bf16_to_f32: bad arguments");
  hipLaunchKernelGGL(bf16_to_f32_kernel, dim3(seg3d_ew_grid(n, 256)), dim3(256), 0, (hipStream_t)stream,
                     (const seg3d_bf16*)src, dst, (i64)n);
  SEG3D_LAUNCH_CHECK("seg3d_bf16_to_f32");
  return SEG3D_OK;
}

// Many weight tensors in ONE launch (after an optimizer step every conv weight has to be re-packed, for the forward
// and for the data-gradient orientation: 52 tiny launches per V-Net step otherwise).  `jobs` is a device array;
// job k owns the workgroups [first_block[k], first_block[k+1]), one per packed (32 x 8 x T) chunk.
// AW = reduction channels per chunk: 8 (fp32 image) or 16 (bf16 image, 8 per half)
// TC = compile-time tap count (27 / 8; 0 = run-time T): the index arithmetic below is one division per element and
// loop, and with a run-time divisor those divisions WERE the kernel (about 60 VALU instructions per element, 134 us for
// the 2 x 45 M packed elements of a V-Net step; divisions by constants are a multiply and a shift)
// LDS tile element of the multi-pack kernel: the value as it will be stored (fp32 image: float, bf16 image: bf16 bits)
template <bool BF> struct Seg3dPackTile;
template <> struct Seg3dPackTile<false> {
  typedef float type;
  static __device__ __forceinline__ float put(float v) { return v; }
  static __device__ __forceinline__ void store(float* wp, i64 i, float v) { wp[i] = v; }
};
template <> struct Seg3dPackTile<true> {
  typedef seg3d_bf16 type;
  static __device__ __forceinline__ seg3d_bf16 put(float v) { return seg3d_f2bf(v); }
  static __device__ __forceinline__ void store(float* wp, i64 i, seg3d_bf16 v) { reinterpret_cast<seg3d_bf16*>(wp)[i] = v; }
};

// WINO (jb.T = 36 / 48, fp32 images only): the 27 taps are read as usual, the packed chunk holds the WINO transformed "taps"
template <int AW, bool BF, int TC, int WINO = 0>
__device__ __forceinline__ void pack_mfma_chunk(const Seg3dPackJob& jb, int chunk,
                                                typename Seg3dPackTile<BF>::type* __restrict__ tile) {
  typedef Seg3dPackTile<BF> TL;
  const int T = TC ? TC : jb.T;
  const int AB = (jb.A + AW - 1) / AW;
  const int ab = chunk % AB, bb = chunk / AB;
  const int a0 = ab * AW, b0 = bb * 32;
  const int n = AW * 32 * T;                                     // elements of this chunk, tile[(a * 32 + b) * T + t]
  if (T <= 27 && jb.sa == T) {
    // w[b][a][t]: for a fixed output channel b the 8 x T values of this chunk are contiguous
    const int run = AW * T;
#pragma unroll 6
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
      const int b = i / run, r = i - b * run;
      const int a = r / T, t = r - a * T;
      float v = 0.f;
      if (a0 + a < jb.A && b0 + b < jb.B) v = jb.w[(i64)(b0 + b) * jb.sb + (i64)a0 * T + r];
      tile[(a * 32 + b) * T + t] = TL::put(v);
    }
  } else if (T <= 27 && jb.sb == T) {
    // w[a][b][t]: for a fixed reduction channel a the 32 x T values are contiguous
    const int run = 32 * T;
#pragma unroll 6
    for (int i = threadIdx.x; i < AW * run; i += 256) {
      const int a = i / run, r = i - a * run;
      float v = 0.f;
      if (a0 + a < jb.A && b0 + r / T < jb.B) v = jb.w[(i64)(a0 + a) * jb.sa + (i64)b0 * T + r];
      tile[a * run + r] = TL::put(v);
    }
  } else {
    for (int i = threadIdx.x; i < n && T <= 27; i += 256) {
      const int t = i % T, ab_ = i / T;
      const int b = ab_ % 32, a = ab_ / 32;
      float v = 0.f;
      if (a0 + a < jb.A && b0 + b < jb.B) v = jb.w[(a0 + a) * jb.sa + (b0 + b) * jb.sb + t];
      tile[i] = TL::put(v);
    }
  }
  __syncthreads();
  // packed order inside the chunk: [t][h][j][r]  with a = 4 h + r, b = j
  constexpr int HW = AW / 2;   // channels per half
  if constexpr (WINO == SEG3D_WINO2D_T && !BF) {
    // one thread per (h, r, j) = one (a, b) pair of the chunk: its 27 taps once from the tile (stride 27 over j: conflict-free),
    // all 3 x 16 transformed values in registers (the arithmetic of seg3d_wino2d_u, term by term), written as the twelve 16-byte
    // words [g] of the image -- for one g the 32 lanes j of a (h, r) write 512 contiguous bytes.  (Round 4's first form computed
    // every one of the 12 288 outputs on its own: nine tile reads, an index decode and the transform per element, 143 us per step.)
    static_assert(AW == 8, "the T = 48 image is an fp32 image");
    const int j = threadIdx.x & 31, r = (threadIdx.x >> 5) & 3, h = threadIdx.x >> 7;
    const float* g = tile + ((HW * h + r) * 32 + j) * 27;
    float gl[27];
#pragma unroll
    for (int i = 0; i < 27; ++i) gl[i] = g[jb.flip ? 26 - i : i];
    float u[48];
#pragma unroll
    for (int kz = 0; kz < 3; ++kz)
#pragma unroll
      for (int px = 0; px < 4; ++px) {
        float rw[3];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) rw[ky] = seg3d_wino_u(gl[(kz * 3 + ky) * 3], gl[(kz * 3 + ky) * 3 + 1], gl[(kz * 3 + ky) * 3 + 2], px);
#pragma unroll
        for (int py = 0; py < 4; ++py) u[kz * 16 + py * 4 + px] = seg3d_wino_u(rw[0], rw[1], rw[2], py);
      }
    float* dst = jb.wp + (i64)chunk * (AW * 32 * WINO) + ((h * 12) * 4 + r) * 128 + j * 4;
#pragma unroll
    for (int gq = 0; gq < 12; ++gq) {
      seg3d_f32x4 o;
#pragma unroll
      for (int s4 = 0; s4 < 4; ++s4) o[s4] = u[seg3d_w2_step_t(4 * gq + s4)];
      *reinterpret_cast<seg3d_f32x4*>(dst + gq * 512) = o;
    }
    return;
  }
  if constexpr (WINO == SEG3D_WINO_T && !BF) {
    const int nw = AW * 32 * WINO;
    for (int i = threadIdx.x; i < nw; i += 256) {
      const int r = i % HW, j = (i / HW) & 31, h = (i / (HW * 32)) & 1, tw = i / (HW * 64);
      const float* g = tile + ((HW * h + r) * 32 + j) * 27;
      const int t9 = tw >> 2;
      const float g0 = g[jb.flip ? 26 - 3 * t9 : 3 * t9], g1 = g[jb.flip ? 25 - 3 * t9 : 3 * t9 + 1],
                  g2 = g[jb.flip ? 24 - 3 * t9 : 3 * t9 + 2];
      jb.wp[(i64)chunk * nw + i] = seg3d_wino_u(g0, g1, g2, tw & 3);
    }
    return;
  }
#pragma unroll 6
  for (int i = threadIdx.x; i < n; i += 256) {
    const int r = i % HW, j = (i / HW) & 31, h = (i / (HW * 32)) & 1, t = i / (HW * 64);
    const typename TL::type v = tile[((HW * h + r) * 32 + j) * T + (jb.flip ? T - 1 - t : t)];
    TL::store(jb.wp, (i64)chunk * n + i, v);
  }
}

template <int AW, bool BF>
__device__ __forceinline__ void pack_mfma_multi_body(const Seg3dPackJob* __restrict__ jobs, int njobs) {
  // One workgroup per packed chunk (8 reduction channels x 32 output channels x T taps = exactly one LDS image of the
  // conv kernels).  In both reference layouts one of the two channel strides equals T, so the chunk's source elements
  // form long contiguous runs: they are read in memory order (coalesced), transposed through LDS and written in packed
  // order (coalesced).  The first version gathered single floats straight from global memory and moved 5x the bytes.
  __shared__ typename Seg3dPackTile<BF>::type tile[AW * 32 * 27];   // bf16 images keep the tile in bf16: 27 KB, 5 workgroups per CU
  __shared__ int sjob;
  if (threadIdx.x == 0) {
    int lo = 0, hi = njobs - 1;  // last job whose first_block <= blockIdx.x
    while (lo < hi) {
      const int mid = (lo + hi + 1) >> 1;
      if (jobs[mid].first_block <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
    }
    sjob = lo;
  }
  __syncthreads();
  const Seg3dPackJob jb = jobs[sjob];
  const int chunk = (int)((i64)blockIdx.x - jb.first_block);   // = bb * AB + ab
  if (jb.T == 27) pack_mfma_chunk<AW, BF, 27>(jb, chunk, tile);
  else if (jb.T == SEG3D_WINO_T) pack_mfma_chunk<AW, BF, 27, SEG3D_WINO_T>(jb, chunk, tile);
  else if (jb.T == SEG3D_WINO2D_T) pack_mfma_chunk<AW, BF, 27, SEG3D_WINO2D_T>(jb, chunk, tile);
  else if (jb.T == 8) pack_mfma_chunk<AW, BF, 8>(jb, chunk, tile);
  else pack_mfma_chunk<AW, BF, 0>(jb, chunk, tile);
}

__global__ __launch_bounds__(256) void pack_mfma_multi_kernel(const Seg3dPackJob* __restrict__ jobs, int njobs) {
  pack_mfma_multi_body<8, false>(jobs, njobs);
}

// bf16 images (bf16 mode): 16 reduction channels per chunk, the same coalesced read / LDS transpose / coalesced write
__global__ __launch_bounds__(256) void pack_mfma_bf16_multi_kernel(const Seg3dPackJob* __restrict__ jobs, int njobs) {
  pack_mfma_multi_body<16, true>(jobs, njobs);
}

extern "C" long long seg3d_pack_job_blocks(int A, int B, int T) {
  return (long long)((B + 31) / 32) * ((A + 7) / 8);   // one workgroup per (32 outputs x 8 inputs) chunk
}

extern "C" int seg3d_pack_weights_mfma_multi(const Seg3dPackJob* jobs_device, int njobs, long long total_blocks,
                                             void* stream) {
  SEG3D_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31),
                "seg3d_pack_weights_mfma_multi: bad arguments");   // job T: 27 / 8 / 1 taps, or 36 / 48 = Winograd images (fp32 only)
  hipLaunchKernelGGL(pack_mfma_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, jobs_device,
                     njobs);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_mfma_multi");
  return SEG3D_OK;
}

extern "C" long long seg3d_pack_job_blocks_bf16(int A, int B, int T) {
  if (T == SEG3D_WINO_T || T == SEG3D_WINO2D_T) return -1;   // Winograd-transformed images exist in fp32 only (pack_mfma_chunk)
  return (long long)((B + 31) / 32) * ((A + 15) / 16);
}

extern "C" int seg3d_pack_weights_mfma_bf16_multi(const Seg3dPackJob* jobs_device, int njobs, long long total_blocks,
                                                  void* stream) {
  SEG3D_REQUIRE(jobs_device && njobs > 0 && total_blocks > 0 && total_blocks < (1ll << 31),
                "seg3d_pack_weights_mfma_bf16_multi: bad arguments");
  hipLaunchKernelGGL(pack_mfma_bf16_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                     jobs_device, njobs);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_mfma_bf16_multi");
  return SEG3D_OK;
}

extern "C" long long seg3d_packed_mfma_floats(int A, int B, int T) {
  return (long long)((B + 31) / 32) * ((A + 7) / 8) * T * 256;
}
