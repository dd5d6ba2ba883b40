// gn.hip -- GroupNorm(1, C) (+ ReLU, + residual add) forward and backward on NDHWC fp32 tensors.
//
// Reference semantics: nn.GroupNorm(1, C), eps 1e-5, affine, i.e. per-sample LayerNorm over C*D*H*W elements
// (network/module/conv_gn_relu3.py:11, vnet_inblock.py:10, vnet_downblock.py:12, vnet_upblock.py:12,
// vnet_outblock.py:14,17), followed by ReLU (conv_gn_relu3.py:14) and, at the end of a residual block,
// ReLU(input + GN(conv)) (residual_block3.py:24,46).
//
// All kernels are HBM-bound streaming passes (16 B per lane).  Spatial reductions: per-thread fp32 partials over
// <= a few thousand elements, wave64 shuffle reduction, one slot per workgroup, then an fp64 finalize in a fixed
// order -- deterministic and accurate enough for the 1e-4 parity bar on 14.2 M-element samples.
//
// forward :  y (raw conv output) -> [stats partials] -> mean/rstd -> out = act(gamma*xhat + beta (+ res))
// backward:  g = dout * [out > 0];  per (n, c): A = sum g, B = sum g*xhat, X = sum xhat
//            dgamma = sum_n B, dbeta = sum_n A, s1_n = sum_c gamma A / M, s2_n = sum_c gamma B / M
//            dy = rstd_n * (gamma_c g - s1_n - xhat s2_n),  dres = g
//            dbias_c (of the producing conv) = sum_{n,s} dy = sum_n rstd_n (gamma_c A_nc - S s1_n - s2_n X_nc)
#include "seg3d_common.h"
#include "seg3d_hip.h"

#define GN_STATS_CHUNK 16384  // elements per workgroup in the statistics pass

// ---- statistics ----------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_stats_partial_kernel(const float* __restrict__ y, float* __restrict__ part,
                                                                 i64 M, int nblk) {
  __shared__ float red[8];
  const int n = blockIdx.y;
  const float* base = y + (i64)n * M;
  const i64 e0 = (i64)blockIdx.x * GN_STATS_CHUNK;
  i64 e1 = e0 + GN_STATS_CHUNK;
  if (e1 > M) e1 = M;
  float s[2] = {0.f, 0.f};
  // the chunk start is a multiple of 4 floats; M*4 bytes per sample keeps 16-byte alignment when M % 4 == 0
  if ((M & 3) == 0) {
    for (i64 e = e0 + 4 * threadIdx.x; e < e1; e += 1024) {
      const float4 v = *reinterpret_cast<const float4*>(base + e);
      s[0] += (v.x + v.y) + (v.z + v.w);
      s[1] += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
    }
  } else {
    for (i64 e = e0 + threadIdx.x; e < e1; e += 256) {
      const float v = base[e];
      s[0] += v;
      s[1] += v * v;
    }
  }
  block_sum_256<2>(s, red);
  if (threadIdx.x == 0) {
    part[((i64)n * nblk + blockIdx.x) * 2 + 0] = s[0];
    part[((i64)n * nblk + blockIdx.x) * 2 + 1] = s[1];
  }
}

// mean_rstd[n] = (mean, rstd) from `count` partial (sum, sumsq) pairs per sample; fp64, fixed order.
__global__ __launch_bounds__(256) void gn_stats_finalize_kernel(const float* __restrict__ part,
                                                                  float* __restrict__ mean_rstd, int count, double M,
                                                                  double eps) {
  // 256 threads per sample, 4 independent float2 loads in flight per thread (the kernel is pure latency: up to ~7000
  // partial pairs per sample behind one dependent chain per thread), fixed combination order -> reproducible
  __shared__ double red[8];
  const int n = blockIdx.x;
  const float2* p = reinterpret_cast<const float2*>(part + (i64)n * count * 2);
  double s = 0.0, ss = 0.0;
  int k = threadIdx.x;
  for (; k + 768 < count; k += 1024) {
    const float2 a = p[k], b = p[k + 256], c = p[k + 512], d = p[k + 768];
    s += ((double)a.x + (double)b.x) + ((double)c.x + (double)d.x);
    ss += ((double)a.y + (double)b.y) + ((double)c.y + (double)d.y);
  }
  for (; k < count; k += 256) {
    const float2 a = p[k];
    s += (double)a.x;
    ss += (double)a.y;
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) {
    red[wave] = s;
    red[4 + wave] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    s = (red[0] + red[1]) + (red[2] + red[3]);
    ss = (red[4] + red[5]) + (red[6] + red[7]);
    const double mean = s / M;
    double var = ss / M - mean * mean;
    if (var < 0.0) var = 0.0;
    mean_rstd[2 * n + 0] = (float)mean;
    mean_rstd[2 * n + 1] = (float)(1.0 / sqrt(var + eps));
  }
}

extern "C" long long seg3d_gn_stats_count(long long M) { return (M + GN_STATS_CHUNK - 1) / GN_STATS_CHUNK; }

// y [N][M] raw values -> part [N][seg3d_gn_stats_count(M)][2]
extern "C" int seg3d_gn_stats_partial(const float* y, float* part, int N, long long M, void* stream) {
  SEG3D_REQUIRE(y && part && N > 0 && M > 0, "seg3d_gn_stats_partial: bad arguments");
  const int nblk = (int)seg3d_gn_stats_count(M);
  hipLaunchKernelGGL(gn_stats_partial_kernel, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, y, part, (i64)M, nblk);
  SEG3D_LAUNCH_CHECK("seg3d_gn_stats_partial");
  return SEG3D_OK;
}

extern "C" int seg3d_gn_stats_finalize(const float* part, float* mean_rstd, int N, int count, long long M, float eps,
                                       void* stream) {
  SEG3D_REQUIRE(part && mean_rstd && N > 0 && count > 0 && M > 0, "seg3d_gn_stats_finalize: bad arguments");
  hipLaunchKernelGGL(gn_stats_finalize_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, part, mean_rstd, count,
                     (double)M, (double)eps);
  SEG3D_LAUNCH_CHECK("seg3d_gn_stats_finalize");
  return SEG3D_OK;
}

// Loads / stores of the streaming GroupNorm passes carry non-temporal hints (each tensor is touched once per pass and is
// far larger than the L2: fp32 step 23.21 -> 23.06 ms).  SEG3D_GN_NT=0 (measurement builds): plain accesses.
#ifndef SEG3D_GN_NT
#define SEG3D_GN_NT 1
#endif
template <bool BF> struct GnQuad : Seg3dQuad<BF> {};
#if SEG3D_GN_NT
template <> struct GnQuad<false> {
  typedef seg3d_f32x4 raw;
  static __device__ __forceinline__ raw load(const void* p, i64 elem) {
    return __builtin_nontemporal_load(reinterpret_cast<const seg3d_f32x4*>(reinterpret_cast<const float*>(p) + elem));
  }
  static __device__ __forceinline__ seg3d_f32x4 cvt(raw r) { return r; }
  static __device__ __forceinline__ void store(void* p, i64 elem, seg3d_f32x4 v) {
    __builtin_nontemporal_store(v, reinterpret_cast<seg3d_f32x4*>(reinterpret_cast<float*>(p) + elem));
  }
};
template <> struct GnQuad<true> {
  typedef uint2 raw;
  static __device__ __forceinline__ raw load(const void* p, i64 elem) {
    typedef unsigned long long u64;
    const u64 v = __builtin_nontemporal_load(reinterpret_cast<const u64*>(reinterpret_cast<const seg3d_bf16*>(p) + elem));
    return make_uint2((unsigned)v, (unsigned)(v >> 32));
  }
  static __device__ __forceinline__ seg3d_f32x4 cvt(raw r) { return Seg3dQuad<true>::cvt(r); }
  static __device__ __forceinline__ void store(void* p, i64 elem, seg3d_f32x4 v) {
    typedef unsigned long long u64;
    const u64 o = (u64)seg3d_pack2bf(v[0], v[1]) | ((u64)seg3d_pack2bf(v[2], v[3]) << 32);
    __builtin_nontemporal_store(o, reinterpret_cast<u64*>(reinterpret_cast<seg3d_bf16*>(p) + elem));
  }
};
#endif

// (voxel, channel quad, sample) of a grid-stride loop over [total_vox][CQ] quads WITHOUT per-element divisions: idx / CQ and
// v / S are 64-bit divisions (~70 vector instructions each, two per 16 bytes moved); the walker divides once per thread and
// then advances by the constant stride with adds and compares
#ifndef GN_U
#define GN_U 2   // quads per thread and trip in the apply passes (1 / 2 / 4 measured: 16.00 / 15.95 / 15.99 ms per step, i.e. equal within the noise -- the passes sit at what the memory system gives a 3-read + 2-write stream)
#endif
struct GnWalker {
  i64 v;       // voxel (row of the [total_vox][C] tensor)
  i64 vs;      // voxel within its sample
  int q, n;    // channel quad, sample
  i64 dv;      // stride / CQ
  int dq;      // stride % CQ
  __device__ __forceinline__ GnWalker(i64 idx, i64 stride, int CQ, i64 S) {
    v = idx / CQ;
    q = (int)(idx - v * CQ);
    n = (int)(v / S);
    vs = v - (i64)n * S;
    dv = stride / CQ;
    dq = (int)(stride - dv * CQ);
  }
  __device__ __forceinline__ void step(int CQ, i64 S) {
    q += dq;
    i64 adv = dv;
    if (q >= CQ) {
      q -= CQ;
      ++adv;
    }
    v += adv;
    vs += adv;
    while (vs >= S) {
      vs -= S;
      ++n;
    }
  }
};

// ---- apply: out = act(gamma*(y-mean)*rstd + beta (+ res)) -------------------------------------------------------
// RES_BF / OUT_BF (bf16 mode): the residual / the output are bf16 activations; y, statistics and the arithmetic fp32
template <bool VEC, bool RES_BF, bool OUT_BF, bool Y_BF = false>
__device__ __forceinline__ void gn_apply_body(const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                              const float* __restrict__ gamma, const float* __restrict__ beta,
                                              const void* __restrict__ res_v, void* __restrict__ out_v, i64 S, int C,
                                              i64 total_vox, int relu, int ldo) {
  const float* res = reinterpret_cast<const float*>(res_v);   // only dereferenced as float when !RES_BF
  float* out = reinterpret_cast<float*>(out_v);
  if (VEC) {
    const int CQ = C >> 2;
    const i64 total = total_vox * CQ;
    const i64 stride = (i64)gridDim.x * 256;
    GnWalker wk((i64)blockIdx.x * 256 + threadIdx.x, stride, CQ, S);
    // GN_U quads per trip, every load issued before the first store: a wave keeps GN_U x (1 or 2) 16-byte loads in flight
    // instead of one or two -- what counts when these passes share a CU with a weight-gradient kernel of the side stream and
    // get a fraction of the wave slots, and it keeps a trip's loads clear of the previous trip's stores
    // (on gfx9 the wait for a load also waits for the stores issued before it)
    auto one = [&](i64 idx, i64 v, int q, int n, typename GnQuad<Y_BF>::raw yraw, typename GnQuad<RES_BF>::raw rraw) {
      const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
      const seg3d_f32x4 yq = GnQuad<Y_BF>::cvt(yraw);   // y: bf16 storage when Y_BF
      const float4 yv = make_float4(yq[0], yq[1], yq[2], yq[3]);
      const float4 g = *reinterpret_cast<const float4*>(gamma + 4 * q);
      const float4 b = *reinterpret_cast<const float4*>(beta + 4 * q);
      float4 o;
      o.x = (yv.x - mean) * rstd * g.x + b.x;
      o.y = (yv.y - mean) * rstd * g.y + b.y;
      o.z = (yv.z - mean) * rstd * g.z + b.z;
      o.w = (yv.w - mean) * rstd * g.w + b.w;
      if (res) {
        const seg3d_f32x4 rv = GnQuad<RES_BF>::cvt(rraw);
        o.x += rv[0]; o.y += rv[1]; o.z += rv[2]; o.w += rv[3];
      }
      if (relu) {
        o.x = fmaxf(o.x, 0.f); o.y = fmaxf(o.y, 0.f); o.z = fmaxf(o.z, 0.f); o.w = fmaxf(o.w, 0.f);
      }
      const seg3d_f32x4 ov = {o.x, o.y, o.z, o.w};
      GnQuad<OUT_BF>::store(out_v, v * ldo + 4 * q, ov);
    };
    i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; idx + (GN_U - 1) * stride < total; idx += GN_U * stride) {
      i64 vv[GN_U];
      int qq[GN_U], nn[GN_U];
      typename GnQuad<Y_BF>::raw yraw[GN_U];
      typename GnQuad<RES_BF>::raw rraw[GN_U];
#pragma unroll
      for (int u = 0; u < GN_U; ++u) {
        vv[u] = wk.v, qq[u] = wk.q, nn[u] = wk.n;
        wk.step(CQ, S);
        yraw[u] = GnQuad<Y_BF>::load(y, (idx + u * stride) * 4);
        if (res) rraw[u] = GnQuad<RES_BF>::load(res_v, (idx + u * stride) * 4);
      }
#pragma unroll
      for (int u = 0; u < GN_U; ++u) one(idx + u * stride, vv[u], qq[u], nn[u], yraw[u], rraw[u]);
    }
    for (; idx < total; idx += stride, wk.step(CQ, S)) {
      typename GnQuad<RES_BF>::raw rraw = {};
      if (res) rraw = GnQuad<RES_BF>::load(res_v, idx * 4);
      one(idx, wk.v, wk.q, wk.n, GnQuad<Y_BF>::load(y, idx * 4), rraw);
    }
  } else {
    const i64 total = total_vox * C;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
      const i64 v = idx / C;
      const int c = (int)(idx - v * C);
      const int n = (int)(v / S);
      const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
      float o = (y[idx] - mean) * rstd * gamma[c] + beta[c];
      if (res) o += res[idx];
      if (relu) o = fmaxf(o, 0.f);
      out[v * ldo + c] = o;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gn_apply_kernel(const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                         const float* __restrict__ gamma, const float* __restrict__ beta,
                                                         const float* __restrict__ res, float* __restrict__ out, i64 S,
                                                         int C, i64 total_vox, int relu, int ldo) {
  gn_apply_body<VEC, false, false>(y, mean_rstd, gamma, beta, res, out, S, C, total_vox, relu, ldo);
}

template <bool RES_BF, bool OUT_BF, bool Y_BF>
__global__ __launch_bounds__(256) void gn_apply_bf16_kernel(const float* __restrict__ y,
                                                              const float* __restrict__ mean_rstd,
                                                              const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const void* __restrict__ res,
                                                              void* __restrict__ out, i64 S, int C, i64 total_vox,
                                                              int relu, int ldo) {
  gn_apply_body<true, RES_BF, OUT_BF, Y_BF>(y, mean_rstd, gamma, beta, res, out, S, C, total_vox, relu, ldo);
}

// bf16 mode: `res` (optional) is bf16 when res_bf16, `out` is bf16 when out_bf16 (ld_out then counts bf16 elements);
// y fp32.  C % 4 == 0.
extern "C" int seg3d_gn_apply_mixed(const void* yv, const float* mean_rstd, const float* gamma, const float* beta,
                                    const void* res, void* out, int N, long long S, int C, int relu, int ld_out,
                                    int res_bf16, int out_bf16, int y_bf16, void* stream) {
  const float* y = reinterpret_cast<const float*>(yv);   // bf16 storage when y_bf16
  SEG3D_REQUIRE(y && mean_rstd && gamma && beta && out && N > 0 && S > 0 && C > 0, "seg3d_gn_apply_mixed: bad arguments");
  SEG3D_REQUIRE((C & 3) == 0, "seg3d_gn_apply_mixed: C must be a multiple of 4 (got %d)", C);
  SEG3D_REQUIRE(ld_out == 0 || (ld_out >= C && (ld_out & 3) == 0), "seg3d_gn_apply_mixed: ld_out must be 0 or a multiple of 4 >= C");
  const int ldo = ld_out ? ld_out : C;
  const i64 total_vox = (i64)N * S;
  hipStream_t s = (hipStream_t)stream;
  const dim3 grid(seg3d_ew_grid(total_vox * (C / 4), 256));
  const bool rb = res && res_bf16, ob = out_bf16 != 0;
#define GN_APPLY_MIXED(RB, OB)                                                                                     \
  do {                                                                                                             \
    if (y_bf16)                                                                                                    \
      hipLaunchKernelGGL((gn_apply_bf16_kernel<RB, OB, true>), grid, dim3(256), 0, s, y, mean_rstd, gamma, beta, res, out, \
                         (i64)S, C, total_vox, relu, ldo);                                                         \
    else                                                                                                           \
      hipLaunchKernelGGL((gn_apply_bf16_kernel<RB, OB, false>), grid, dim3(256), 0, s, y, mean_rstd, gamma, beta, res,   \
                         out, (i64)S, C, total_vox, relu, ldo);                                                    \
  } while (0)
  if (rb && ob) GN_APPLY_MIXED(true, true);
  else if (rb) GN_APPLY_MIXED(true, false);
  else if (ob) GN_APPLY_MIXED(false, true);
  else GN_APPLY_MIXED(false, false);
#undef GN_APPLY_MIXED
  SEG3D_LAUNCH_CHECK("seg3d_gn_apply_mixed");
  return SEG3D_OK;
}

extern "C" int seg3d_gn_apply(const float* y, const float* mean_rstd, const float* gamma, const float* beta,
                              const float* res, float* out, int N, long long S, int C, int relu, int ld_out,
                              void* stream) {
  SEG3D_REQUIRE(y && mean_rstd && gamma && beta && out && N > 0 && S > 0 && C > 0, "seg3d_gn_apply: bad arguments");
  SEG3D_REQUIRE(ld_out == 0 || (ld_out >= C && (ld_out & 3) == 0), "seg3d_gn_apply: ld_out must be 0 or a multiple of 4 >= C");
  const int ldo = ld_out ? ld_out : C;
  const i64 total_vox = (i64)N * S;
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) == 0) {
    hipLaunchKernelGGL((gn_apply_kernel<true>), dim3(seg3d_ew_grid(total_vox * (C / 4), 256)), dim3(256), 0, s, y, mean_rstd,
                       gamma, beta, res, out, (i64)S, C, total_vox, relu, ldo);
  } else {
    hipLaunchKernelGGL((gn_apply_kernel<false>), dim3(seg3d_ew_grid(total_vox * C, 256)), dim3(256), 0, s, y, mean_rstd,
                       gamma, beta, res, out, (i64)S, C, total_vox, relu, ldo);
  }
  SEG3D_LAUNCH_CHECK("seg3d_gn_apply");
  return SEG3D_OK;
}

// ---- backward reduce: per (n, c) partial A = sum g, B = sum g*xhat, X = sum xhat ---------------------------------
// voxels per workgroup of the backward reduce: <= 2048, but small levels are cut finer so that the pass still
// spreads over >= ~128 workgroups per sample (a 12^3 x 256-channel tensor would otherwise run on 4 workgroups)
static inline int gn_bwd_vpb(i64 S) {
  i64 v = (S + 127) / 128;
  if (v < 32) v = 32;
  if (v > 2048) v = 2048;
  return (int)v;
}

// fast path: C % 4 == 0 and (C/4) divides 256: thread = (channel quad, voxel lane)
// ACT_BF (bf16 mode): dout and out (activation-side tensors) are bf16
template <bool ACT_BF, bool Y_BF = false>
__device__ __forceinline__ void gn_bwd_reduce_vec_body(const void* __restrict__ dout, const void* __restrict__ out,
                                                       const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                       const float* __restrict__ gamma, const float* __restrict__ beta,
                                                       float* __restrict__ part, i64 S, int C, int nblk, int relu,
                                                       int vpb, int ldd) {
  __shared__ float red[256 * 12];
  const int n = blockIdx.y;
  const int CQ = C >> 2, VL = 256 / CQ;
  const int q = threadIdx.x % CQ, vl = threadIdx.x / CQ;
  const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
  const i64 s0 = (i64)blockIdx.x * vpb;
  i64 s1 = s0 + vpb;
  if (s1 > S) s1 = S;
  float a[4] = {0, 0, 0, 0}, bb[4] = {0, 0, 0, 0}, xx[4] = {0, 0, 0, 0};
  // 4 voxels per trip: 12 independent 16-byte loads in flight per thread (the kernel is pure streaming)
  for (i64 sv = s0 + vl; sv < s1; sv += 4 * VL) {
    float4 g[4], yv[4], o[4];
    typename GnQuad<ACT_BF>::raw graw[4], oraw[4];
    typename GnQuad<Y_BF>::raw yraw[4];
    bool ok[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const i64 svu = sv + u * VL;
      ok[u] = svu < s1;
      const i64 off = ((i64)n * S + (ok[u] ? svu : s0)) * C + 4 * q;
      graw[u] = GnQuad<ACT_BF>::load(dout, ((i64)n * S + (ok[u] ? svu : s0)) * ldd + 4 * q);
      yraw[u] = GnQuad<Y_BF>::load(y, off);
      if (relu && out) oraw[u] = GnQuad<ACT_BF>::load(out, off);
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const seg3d_f32x4 gv = GnQuad<ACT_BF>::cvt(graw[u]);
      g[u] = make_float4(gv[0], gv[1], gv[2], gv[3]);
      const seg3d_f32x4 yq = GnQuad<Y_BF>::cvt(yraw[u]);
      yv[u] = make_float4(yq[0], yq[1], yq[2], yq[3]);
      if (relu && out) {
        const seg3d_f32x4 ov = GnQuad<ACT_BF>::cvt(oraw[u]);
        o[u] = make_float4(ov[0], ov[1], ov[2], ov[3]);
      }
    }
    if (relu && !out) {
      // no residual: the forward output is a pure function of y -> recompute it (same expression as gn_apply_kernel)
      // instead of reading a third tensor
      const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * q);
      const float4 bt = *reinterpret_cast<const float4*>(beta + 4 * q);
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        o[u].x = (yv[u].x - mean) * rstd * gm.x + bt.x; o[u].y = (yv[u].y - mean) * rstd * gm.y + bt.y;
        o[u].z = (yv[u].z - mean) * rstd * gm.z + bt.z; o[u].w = (yv[u].w - mean) * rstd * gm.w + bt.w;
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (!ok[u]) continue;
      float4 gg = g[u];
      if (relu) {
        gg.x = o[u].x > 0.f ? gg.x : 0.f; gg.y = o[u].y > 0.f ? gg.y : 0.f;
        gg.z = o[u].z > 0.f ? gg.z : 0.f; gg.w = o[u].w > 0.f ? gg.w : 0.f;
      }
      const float x0 = (yv[u].x - mean) * rstd, x1 = (yv[u].y - mean) * rstd, x2 = (yv[u].z - mean) * rstd,
                  x3 = (yv[u].w - mean) * rstd;
      a[0] += gg.x; a[1] += gg.y; a[2] += gg.z; a[3] += gg.w;
      bb[0] += gg.x * x0; bb[1] += gg.y * x1; bb[2] += gg.z * x2; bb[3] += gg.w * x3;
      xx[0] += x0; xx[1] += x1; xx[2] += x2; xx[3] += x3;
    }
  }
  float* r = red + threadIdx.x * 12;
#pragma unroll
  for (int k = 0; k < 4; ++k) { r[k] = a[k]; r[4 + k] = bb[k]; r[8 + k] = xx[k]; }
  __syncthreads();
  if (vl == 0) {
    float acc[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) acc[k] = r[k];
    for (int j = 1; j < VL; ++j) {
      const float* o = red + (j * CQ + q) * 12;
#pragma unroll
      for (int k = 0; k < 12; ++k) acc[k] += o[k];
    }
    // part[n][blk][c][3]
    float* dst = part + (((i64)n * nblk + blockIdx.x) * C + 4 * q) * 3;
#pragma unroll
    for (int k = 0; k < 4; ++k) { dst[3 * k + 0] = acc[k]; dst[3 * k + 1] = acc[4 + k]; dst[3 * k + 2] = acc[8 + k]; }
  }
}

__global__ __launch_bounds__(256) void gn_bwd_reduce_vec_kernel(const float* __restrict__ dout,
                                                                  const float* __restrict__ out,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ mean_rstd,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta,
                                                                  float* __restrict__ part, i64 S, int C, int nblk,
                                                                  int relu, int vpb, int ldd) {
  gn_bwd_reduce_vec_body<false>(dout, out, y, mean_rstd, gamma, beta, part, S, C, nblk, relu, vpb, ldd);
}

template <bool Y_BF>
__global__ __launch_bounds__(256) void gn_bwd_reduce_vec_bf16_kernel(const void* __restrict__ dout,
                                                                       const void* __restrict__ out,
                                                                       const float* __restrict__ y,
                                                                       const float* __restrict__ mean_rstd,
                                                                       const float* __restrict__ gamma,
                                                                       const float* __restrict__ beta,
                                                                       float* __restrict__ part, i64 S, int C, int nblk,
                                                                       int relu, int vpb, int ldd) {
  gn_bwd_reduce_vec_body<true, Y_BF>(dout, out, y, mean_rstd, gamma, beta, part, S, C, nblk, relu, vpb, ldd);
}

// small-C path (C <= 16, e.g. the num_classes-channel head): thread = voxel, channels in registers
#define GN_SMALLC 16
__global__ __launch_bounds__(256) void gn_bwd_reduce_small_kernel(const float* __restrict__ dout,
                                                                    const float* __restrict__ out,
                                                                    const float* __restrict__ y,
                                                                    const float* __restrict__ mean_rstd,
                                                                    const float* __restrict__ gamma,
                                                                    const float* __restrict__ beta,
                                                                    float* __restrict__ part, i64 S, int C, int nblk,
                                                                    int relu, int vpb, int ldd) {
  __shared__ float red[4 * 3];
  const int n = blockIdx.y;
  const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
  const i64 s0 = (i64)blockIdx.x * vpb;
  i64 s1 = s0 + vpb;
  if (s1 > S) s1 = S;
  float a[GN_SMALLC], bb[GN_SMALLC], xx[GN_SMALLC];
#pragma unroll
  for (int c = 0; c < GN_SMALLC; ++c) { a[c] = 0.f; bb[c] = 0.f; xx[c] = 0.f; }
  for (i64 sv = s0 + threadIdx.x; sv < s1; sv += 256) {
    const i64 off = ((i64)n * S + sv) * C;
#pragma unroll
    for (int c = 0; c < GN_SMALLC; ++c) {
      if (c < C) {
        float g = dout[((i64)n * S + sv) * ldd + c];
        const float ov = out ? out[off + c] : (y[off + c] - mean) * rstd * gamma[c] + beta[c];
        if (relu && !(ov > 0.f)) g = 0.f;
        const float xh = (y[off + c] - mean) * rstd;
        a[c] += g;
        bb[c] += g * xh;
        xx[c] += xh;
      }
    }
  }
#pragma unroll
  for (int c = 0; c < GN_SMALLC; ++c) {
    if (c < C) {  // uniform
      float v[3] = {a[c], bb[c], xx[c]};
      block_sum_256<3>(v, red);
      if (threadIdx.x == 0) {
        float* dst = part + (((i64)n * nblk + blockIdx.x) * C + c) * 3;
        dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2];
      }
    }
  }
}

extern "C" long long seg3d_gn_bwd_blocks(long long S) { const int v = gn_bwd_vpb(S); return (S + v - 1) / v; }

static bool gn_vec_ok(int C) { return (C & 3) == 0 && (C >> 2) <= 256 && (256 % (C >> 2)) == 0; }

// part: [N][seg3d_gn_bwd_blocks(S)][C][3]
// `out` = forward output of the unit (needed for the ReLU mask only when a residual was added); with out == NULL the
// mask is recomputed from y, gamma, beta.
extern "C" int seg3d_gn_bwd_reduce(const float* dout, const float* out, const float* y, const float* mean_rstd,
                                   const float* gamma, const float* beta, float* part, int N, long long S, int C,
                                   int relu, int ld_dout, void* stream) {
  SEG3D_REQUIRE(dout && y && mean_rstd && part && N > 0 && S > 0 && C > 0, "seg3d_gn_bwd_reduce: bad arguments");
  SEG3D_REQUIRE(ld_dout == 0 || (ld_dout >= C && (ld_dout & 3) == 0),
                "seg3d_gn_bwd_reduce: ld_dout must be 0 or a multiple of 4 >= C");
  const int ldd = ld_dout ? ld_dout : C;
  SEG3D_REQUIRE(!relu || out || (gamma && beta), "seg3d_gn_bwd_reduce: relu mask needs the forward output or gamma/beta");
  const int nblk = (int)seg3d_gn_bwd_blocks(S);
  hipStream_t s = (hipStream_t)stream;
  if (gn_vec_ok(C)) {
    hipLaunchKernelGGL(gn_bwd_reduce_vec_kernel, dim3(nblk, N), dim3(256), 0, s, dout, out, y, mean_rstd, gamma, beta, part,
                       (i64)S, C, nblk, relu, gn_bwd_vpb(S), ldd);
  } else if (C <= GN_SMALLC) {
    hipLaunchKernelGGL(gn_bwd_reduce_small_kernel, dim3(nblk, N), dim3(256), 0, s, dout, out, y, mean_rstd, gamma, beta,
                       part, (i64)S, C, nblk, relu, gn_bwd_vpb(S), ldd);
  } else {
    SEG3D_UNSUPPORTED("seg3d_gn_bwd_reduce: unsupported channel count %d (need C<=16 or C%%4==0 with C/4 | 256)", C);
  }
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_reduce");
  return SEG3D_OK;
}

// bf16 mode: dout and out (optional) are bf16 activations-side tensors; y fp32.  C % 4 == 0 with (C/4) | 256.
extern "C" int seg3d_gn_bwd_reduce_bf16(const void* dout, const void* out, const void* yv, const float* mean_rstd,
                                        const float* gamma, const float* beta, float* part, int N, long long S, int C,
                                        int relu, int ld_dout, int y_bf16, void* stream) {
  const float* y = reinterpret_cast<const float*>(yv);
  SEG3D_REQUIRE(dout && y && mean_rstd && part && N > 0 && S > 0 && C > 0, "seg3d_gn_bwd_reduce_bf16: bad arguments");
  SEG3D_REQUIRE(ld_dout == 0 || (ld_dout >= C && (ld_dout & 3) == 0),
                "seg3d_gn_bwd_reduce_bf16: ld_dout must be 0 or a multiple of 4 >= C");
  SEG3D_REQUIRE(gn_vec_ok(C), "seg3d_gn_bwd_reduce_bf16: needs C %% 4 == 0 with C/4 dividing 256 (got %d)", C);
  SEG3D_REQUIRE(!relu || out || (gamma && beta), "seg3d_gn_bwd_reduce_bf16: relu mask needs the forward output or gamma/beta");
  const int ldd = ld_dout ? ld_dout : C;
  const int nblk = (int)seg3d_gn_bwd_blocks(S);
  if (y_bf16)
    hipLaunchKernelGGL(gn_bwd_reduce_vec_bf16_kernel<true>, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, dout, out, y,
                       mean_rstd, gamma, beta, part, (i64)S, C, nblk, relu, gn_bwd_vpb(S), ldd);
  else
    hipLaunchKernelGGL(gn_bwd_reduce_vec_bf16_kernel<false>, dim3(nblk, N), dim3(256), 0, (hipStream_t)stream, dout, out, y,
                       mean_rstd, gamma, beta, part, (i64)S, C, nblk, relu, gn_bwd_vpb(S), ldd);
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_reduce_bf16");
  return SEG3D_OK;
}

// finalize: one workgroup per sample sums the partial slabs (fp64), writes abx[n][c][3] and s12[n][2]
__device__ __forceinline__ void gn_bwd_finalize_sample_body(const float* __restrict__ part,
                                                            const float* __restrict__ gamma,
                                                            float* __restrict__ abx, float* __restrict__ s12,
                                                            int C, int nblk, double M) {
  // thread = (channel c, slab group p): PARTS groups split the nblk partial slabs; combined in fixed order through LDS
  __shared__ double red[256 * 3];
  __shared__ double red2[2 * 4];
  const int n = blockIdx.x;
  const int CW = C < 256 ? C : 256;
  const int PARTS = 256 / CW;
  double s1 = 0.0, s2 = 0.0;
  for (int c0 = 0; c0 < C; c0 += CW) {
    const int cl = threadIdx.x % CW, pg = threadIdx.x / CW;
    const int c = c0 + cl;
    double A = 0.0, B = 0.0, X = 0.0;
    if (pg < PARTS && c < C) {
      const float* p = part + ((i64)n * nblk * C + c) * 3;
      int k = pg;
      for (; k + 3 * PARTS < nblk; k += 4 * PARTS) {   // 12 independent loads in flight (latency-bound kernel)
        const float* q0 = p + (i64)k * C * 3;
        const float* q1 = q0 + (i64)PARTS * C * 3;
        const float* q2 = q1 + (i64)PARTS * C * 3;
        const float* q3 = q2 + (i64)PARTS * C * 3;
        const float a0 = q0[0], b0 = q0[1], x0 = q0[2], a1 = q1[0], b1 = q1[1], x1 = q1[2];
        const float a2 = q2[0], b2 = q2[1], x2 = q2[2], a3 = q3[0], b3 = q3[1], x3 = q3[2];
        A += ((double)a0 + (double)a1) + ((double)a2 + (double)a3);
        B += ((double)b0 + (double)b1) + ((double)b2 + (double)b3);
        X += ((double)x0 + (double)x1) + ((double)x2 + (double)x3);
      }
      for (; k < nblk; k += PARTS) {
        A += (double)p[(i64)k * C * 3 + 0];
        B += (double)p[(i64)k * C * 3 + 1];
        X += (double)p[(i64)k * C * 3 + 2];
      }
    }
    __syncthreads();
    red[threadIdx.x * 3 + 0] = A; red[threadIdx.x * 3 + 1] = B; red[threadIdx.x * 3 + 2] = X;
    __syncthreads();
    if (pg == 0 && c < C) {
      for (int k = 1; k < PARTS; ++k) {
        A += red[(k * CW + cl) * 3 + 0]; B += red[(k * CW + cl) * 3 + 1]; X += red[(k * CW + cl) * 3 + 2];
      }
      float* d = abx + ((i64)n * C + c) * 3;
      d[0] = (float)A; d[1] = (float)B; d[2] = (float)X;
      s1 += (double)gamma[c] * A;
      s2 += (double)gamma[c] * B;
    }
  }
  s1 = wave_sum_d(s1);
  s2 = wave_sum_d(s2);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) { red2[wave] = s1; red2[4 + wave] = s2; }
  __syncthreads();
  if (threadIdx.x == 0) {
    s12[2 * n + 0] = (float)((red2[0] + red2[1] + red2[2] + red2[3]) / M);
    s12[2 * n + 1] = (float)((red2[4] + red2[5] + red2[6] + red2[7]) / M);
  }
}

__global__ __launch_bounds__(256) void gn_bwd_finalize_sample_kernel(const float* __restrict__ part,
                                                                       const float* __restrict__ gamma,
                                                                       float* __restrict__ abx, float* __restrict__ s12,
                                                                       int C, int nblk, double M) {
  gn_bwd_finalize_sample_body(part, gamma, abx, s12, C, nblk, M);
}

// Both finalize stages in ONE launch (the two tiny kernels cost 9 + 6 us of a 7 ms bf16 step 29 times): every sample's
// workgroup takes a ticket when its abx / s12 are written; the workgroup that draws the last ticket runs the parameter
// stage over all samples (fixed order: same result as the two-kernel path) and puts the ticket counter back to zero,
// so the counter (one device int, zero before the first call) is reusable by the next call and by graph replays.
__global__ __launch_bounds__(256) void gn_bwd_finalize_fused_kernel(const float* __restrict__ part,
                                                                      const float* __restrict__ gamma,
                                                                      const float* __restrict__ mean_rstd,
                                                                      float* __restrict__ abx, float* __restrict__ s12,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ dbias, int* __restrict__ ticket, int N,
                                                                      int C, int nblk, double S, int acc_mask) {
  gn_bwd_finalize_sample_body(part, gamma, abx, s12, C, nblk, S * (double)C);
  __shared__ int is_last;
  __threadfence();          // this workgroup's abx / s12 are visible device-wide before its ticket is
  __syncthreads();
  if (threadIdx.x == 0) is_last = (atomicAdd(ticket, 1) == N - 1);
  __syncthreads();
  if (!is_last) return;
  __threadfence();
  if (threadIdx.x == 0) *ticket = 0;
  const volatile float* vabx = abx;    // written by other workgroups of this launch: read past the L1
  const volatile float* vs12 = s12;
  for (int c = threadIdx.x; c < C; c += 256) {
    double dg = 0.0, db = 0.0, dc = 0.0;
    for (int n = 0; n < N; ++n) {
      const volatile float* d = vabx + ((i64)n * C + c) * 3;
      const double A = d[0], B = d[1], X = d[2];
      dg += B;
      db += A;
      dc += (double)mean_rstd[2 * n + 1] * ((double)gamma[c] * A - S * (double)vs12[2 * n] - (double)vs12[2 * n + 1] * X);
    }
    dgamma[c] = (acc_mask & 1) ? dgamma[c] + (float)dg : (float)dg;
    dbeta[c] = (acc_mask & 2) ? dbeta[c] + (float)db : (float)db;
    if (dbias) dbias[c] = (acc_mask & 4) ? dbias[c] + (float)dc : (float)dc;
  }
}

// dgamma[c] = sum_n B, dbeta[c] = sum_n A, dbias[c] = sum_n rstd_n (gamma_c A - S s1_n - s2_n X)
__global__ __launch_bounds__(256) void gn_bwd_finalize_param_kernel(const float* __restrict__ abx,
                                                                      const float* __restrict__ s12,
                                                                      const float* __restrict__ mean_rstd,
                                                                      const float* __restrict__ gamma,
                                                                      float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                                      float* __restrict__ dbias, int N, int C, double S,
                                                                      int acc_mask) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= C) return;
  double dg = 0.0, db = 0.0, dc = 0.0;
  for (int n = 0; n < N; ++n) {
    const float* d = abx + ((i64)n * C + c) * 3;
    const double A = d[0], B = d[1], X = d[2];
    dg += B;
    db += A;
    dc += (double)mean_rstd[2 * n + 1] * ((double)gamma[c] * A - S * (double)s12[2 * n] - (double)s12[2 * n + 1] * X);
  }
  dgamma[c] = (acc_mask & 1) ? dgamma[c] + (float)dg : (float)dg;
  dbeta[c] = (acc_mask & 2) ? dbeta[c] + (float)db : (float)db;
  if (dbias) dbias[c] = (acc_mask & 4) ? dbias[c] + (float)dc : (float)dc;
}

// abx: [N][C][3] scratch, s12: [N][2]
extern "C" int seg3d_gn_bwd_finalize(const float* part, const float* gamma, const float* mean_rstd, float* abx, float* s12,
                                     float* dgamma, float* dbeta, float* dbias, int N, long long S, int C, int acc_mask,
                                     void* stream) {
  SEG3D_REQUIRE(part && gamma && mean_rstd && abx && s12 && dgamma && dbeta && N > 0 && S > 0 && C > 0,
                "seg3d_gn_bwd_finalize: bad arguments");
  const int nblk = (int)seg3d_gn_bwd_blocks(S);
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(gn_bwd_finalize_sample_kernel, dim3(N), dim3(256), 0, s, part, gamma, abx, s12, C, nblk,
                     (double)S * (double)C);
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_finalize(sample)");
  hipLaunchKernelGGL(gn_bwd_finalize_param_kernel, dim3(seg3d_cdiv(C, 256)), dim3(256), 0, s, abx, s12, mean_rstd, gamma,
                     dgamma, dbeta, dbias, N, C, (double)S, acc_mask);
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_finalize(param)");
  return SEG3D_OK;
}

// the same with one launch; ticket: one device int that is zero before the first call (left zero by every call)
extern "C" int seg3d_gn_bwd_finalize_fused(const float* part, const float* gamma, const float* mean_rstd, float* abx,
                                           float* s12, float* dgamma, float* dbeta, float* dbias, int* ticket, int N,
                                           long long S, int C, int acc_mask, void* stream) {
  SEG3D_REQUIRE(part && gamma && mean_rstd && abx && s12 && dgamma && dbeta && ticket && N > 0 && S > 0 && C > 0,
                "seg3d_gn_bwd_finalize_fused: bad arguments");
  const int nblk = (int)seg3d_gn_bwd_blocks(S);
  hipLaunchKernelGGL(gn_bwd_finalize_fused_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, part, gamma, mean_rstd, abx, s12,
                     dgamma, dbeta, dbias, ticket, N, C, nblk, (double)S, acc_mask);
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_finalize_fused");
  return SEG3D_OK;
}

// ---- backward apply: dy = rstd (gamma g - s1 - xhat s2),  dres = g -----------------------------------------------
// ACT_BF: dout / out are bf16; DY_BF: dy (the gradient handed to the conv's dgrad / wgrad kernels) is written as bf16.
// dres (gradient of the identity path, folded into a dgrad epilogue) stays fp32.
template <bool VEC, bool ACT_BF, bool DY_BF, bool Y_BF = false>
__device__ __forceinline__ void gn_bwd_apply_body(const void* __restrict__ dout_v, const void* __restrict__ out_v,
                                                  const float* __restrict__ y, const float* __restrict__ mean_rstd,
                                                  const float* __restrict__ s12, const float* __restrict__ gamma,
                                                  const float* __restrict__ beta, void* __restrict__ dy_v,
                                                  float* __restrict__ dres, i64 S, int C, i64 total_vox, int relu,
                                                  int ldd) {
  const float* dout = reinterpret_cast<const float*>(dout_v);   // dereferenced as float only when !ACT_BF
  const float* out = reinterpret_cast<const float*>(out_v);
  float* dy = reinterpret_cast<float*>(dy_v);
  if (VEC) {
    const int CQ = C >> 2;
    const i64 total = total_vox * CQ;
    const i64 stride = (i64)gridDim.x * 256;
    GnWalker wk((i64)blockIdx.x * 256 + threadIdx.x, stride, CQ, S);
    const bool need_out = relu && out;
    // GN_U quads per trip, all loads ahead of the first store (see gn_apply_body)
    auto one = [&](i64 idx, int q, int n, typename GnQuad<ACT_BF>::raw graw, typename GnQuad<Y_BF>::raw yraw,
                   typename GnQuad<ACT_BF>::raw oraw) {
      const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
      const float s1 = s12[2 * n], s2 = s12[2 * n + 1];
      const seg3d_f32x4 gq = GnQuad<ACT_BF>::cvt(graw);
      float4 g = make_float4(gq[0], gq[1], gq[2], gq[3]);
      const seg3d_f32x4 yq = GnQuad<Y_BF>::cvt(yraw);
      const float4 yv = make_float4(yq[0], yq[1], yq[2], yq[3]);
      const float4 gm = *reinterpret_cast<const float4*>(gamma + 4 * q);
      if (relu) {
        float4 o;
        if (out) {
          const seg3d_f32x4 oq = GnQuad<ACT_BF>::cvt(oraw);
          o = make_float4(oq[0], oq[1], oq[2], oq[3]);
        } else {
          const float4 bt = *reinterpret_cast<const float4*>(beta + 4 * q);
          o.x = (yv.x - mean) * rstd * gm.x + bt.x; o.y = (yv.y - mean) * rstd * gm.y + bt.y;
          o.z = (yv.z - mean) * rstd * gm.z + bt.z; o.w = (yv.w - mean) * rstd * gm.w + bt.w;
        }
        g.x = o.x > 0.f ? g.x : 0.f; g.y = o.y > 0.f ? g.y : 0.f; g.z = o.z > 0.f ? g.z : 0.f; g.w = o.w > 0.f ? g.w : 0.f;
      }
      float4 d;
      d.x = rstd * (gm.x * g.x - s1 - (yv.x - mean) * rstd * s2);
      d.y = rstd * (gm.y * g.y - s1 - (yv.y - mean) * rstd * s2);
      d.z = rstd * (gm.z * g.z - s1 - (yv.z - mean) * rstd * s2);
      d.w = rstd * (gm.w * g.w - s1 - (yv.w - mean) * rstd * s2);
      const seg3d_f32x4 dq = {d.x, d.y, d.z, d.w};
      GnQuad<DY_BF>::store(dy_v, idx * 4, dq);
      if (dres) *reinterpret_cast<float4*>(dres + idx * 4) = g;
    };
    i64 idx = (i64)blockIdx.x * 256 + threadIdx.x;
    for (; idx + (GN_U - 1) * stride < total; idx += GN_U * stride) {
      int qq[GN_U], nn[GN_U];
      typename GnQuad<ACT_BF>::raw graw[GN_U], oraw[GN_U];
      typename GnQuad<Y_BF>::raw yraw[GN_U];
#pragma unroll
      for (int u = 0; u < GN_U; ++u) {
        qq[u] = wk.q, nn[u] = wk.n;
        graw[u] = GnQuad<ACT_BF>::load(dout_v, wk.v * ldd + 4 * wk.q);
        wk.step(CQ, S);
        yraw[u] = GnQuad<Y_BF>::load(y, (idx + u * stride) * 4);
        if (need_out) oraw[u] = GnQuad<ACT_BF>::load(out_v, (idx + u * stride) * 4);
      }
#pragma unroll
      for (int u = 0; u < GN_U; ++u) one(idx + u * stride, qq[u], nn[u], graw[u], yraw[u], oraw[u]);
    }
    for (; idx < total; idx += stride, wk.step(CQ, S)) {
      typename GnQuad<ACT_BF>::raw oraw = {};
      if (need_out) oraw = GnQuad<ACT_BF>::load(out_v, idx * 4);
      one(idx, wk.q, wk.n, GnQuad<ACT_BF>::load(dout_v, wk.v * ldd + 4 * wk.q), GnQuad<Y_BF>::load(y, idx * 4), oraw);
    }
  } else {
    const i64 total = total_vox * C;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
      const i64 v = idx / C;
      const int c = (int)(idx - v * C);
      const int n = (int)(v / S);
      const float mean = mean_rstd[2 * n], rstd = mean_rstd[2 * n + 1];
      float g = dout[v * ldd + c];
      const float ov = out ? out[idx] : (y[idx] - mean) * rstd * gamma[c] + beta[c];
      if (relu && !(ov > 0.f)) g = 0.f;
      dy[idx] = rstd * (gamma[c] * g - s12[2 * n] - (y[idx] - mean) * rstd * s12[2 * n + 1]);
      if (dres) dres[idx] = g;
    }
  }
}

template <bool VEC>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                             const float* __restrict__ y,
                                                             const float* __restrict__ mean_rstd,
                                                             const float* __restrict__ s12,
                                                             const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, float* __restrict__ dy,
                                                             float* __restrict__ dres, i64 S, int C, i64 total_vox,
                                                             int relu, int ldd) {
  gn_bwd_apply_body<VEC, false, false>(dout, out, y, mean_rstd, s12, gamma, beta, dy, dres, S, C, total_vox, relu, ldd);
}

template <bool DY_BF, bool Y_BF>
__global__ __launch_bounds__(256) void gn_bwd_apply_bf16_kernel(const void* __restrict__ dout, const void* __restrict__ out,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ mean_rstd,
                                                                  const float* __restrict__ s12,
                                                                  const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, void* __restrict__ dy,
                                                                  float* __restrict__ dres, i64 S, int C, i64 total_vox,
                                                                  int relu, int ldd) {
  gn_bwd_apply_body<true, true, DY_BF, Y_BF>(dout, out, y, mean_rstd, s12, gamma, beta, dy, dres, S, C, total_vox, relu,
                                             ldd);
}

// bf16 mode: dout / out bf16; dy bf16 when dy_bf16 (else fp32); dres fp32.  C % 4 == 0.
extern "C" int seg3d_gn_bwd_apply_bf16(const void* dout, const void* out, const void* yv, const float* mean_rstd,
                                       const float* s12, const float* gamma, const float* beta, void* dy, float* dres,
                                       int N, long long S, int C, int relu, int ld_dout, int dy_bf16, int y_bf16,
                                       void* stream) {
  const float* y = reinterpret_cast<const float*>(yv);
  SEG3D_REQUIRE(dout && y && mean_rstd && s12 && gamma && dy && N > 0 && S > 0 && C > 0,
                "seg3d_gn_bwd_apply_bf16: bad arguments");
  SEG3D_REQUIRE((C & 3) == 0, "seg3d_gn_bwd_apply_bf16: C must be a multiple of 4 (got %d)", C);
  SEG3D_REQUIRE(ld_dout == 0 || (ld_dout >= C && (ld_dout & 3) == 0),
                "seg3d_gn_bwd_apply_bf16: ld_dout must be 0 or a multiple of 4 >= C");
  SEG3D_REQUIRE(!relu || out || beta, "seg3d_gn_bwd_apply_bf16: relu mask needs the forward output or beta");
  const int ldd = ld_dout ? ld_dout : C;
  const i64 total_vox = (i64)N * S;
  const dim3 grid(seg3d_ew_grid(total_vox * (C / 4), 256));
  hipStream_t s = (hipStream_t)stream;
#define GN_BWD_APPLY16(DB, YB)                                                                                   \
  hipLaunchKernelGGL((gn_bwd_apply_bf16_kernel<DB, YB>), grid, dim3(256), 0, s, dout, out, y, mean_rstd, s12, gamma, beta, \
                     dy, dres, (i64)S, C, total_vox, relu, ldd)
  if (dy_bf16 && y_bf16) GN_BWD_APPLY16(true, true);
  else if (dy_bf16) GN_BWD_APPLY16(true, false);
  else if (y_bf16) GN_BWD_APPLY16(false, true);
  else GN_BWD_APPLY16(false, false);
#undef GN_BWD_APPLY16
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_apply_bf16");
  return SEG3D_OK;
}

extern "C" int seg3d_gn_bwd_apply(const float* dout, const float* out, const float* y, const float* mean_rstd,
                                  const float* s12, const float* gamma, const float* beta, float* dy, float* dres, int N,
                                  long long S, int C, int relu, int ld_dout, void* stream) {
  SEG3D_REQUIRE(dout && y && mean_rstd && s12 && gamma && dy && N > 0 && S > 0 && C > 0, "seg3d_gn_bwd_apply: bad arguments");
  SEG3D_REQUIRE(ld_dout == 0 || (ld_dout >= C && (ld_dout & 3) == 0),
                "seg3d_gn_bwd_apply: ld_dout must be 0 or a multiple of 4 >= C");
  const int ldd = ld_dout ? ld_dout : C;
  SEG3D_REQUIRE(!relu || out || beta, "seg3d_gn_bwd_apply: relu mask needs the forward output or beta");
  const i64 total_vox = (i64)N * S;
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) == 0) {
    hipLaunchKernelGGL((gn_bwd_apply_kernel<true>), dim3(seg3d_ew_grid(total_vox * (C / 4), 256)), dim3(256), 0, s, dout, out,
                       y, mean_rstd, s12, gamma, beta, dy, dres, (i64)S, C, total_vox, relu, ldd);
  } else {
    hipLaunchKernelGGL((gn_bwd_apply_kernel<false>), dim3(seg3d_ew_grid(total_vox * C, 256)), dim3(256), 0, s, dout, out, y,
                       mean_rstd, s12, gamma, beta, dy, dres, (i64)S, C, total_vox, relu, ldd);
  }
  SEG3D_LAUNCH_CHECK("seg3d_gn_bwd_apply");
  return SEG3D_OK;
}
