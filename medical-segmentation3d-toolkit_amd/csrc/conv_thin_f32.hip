// conv_thin_f32.hip -- the fp32 thin 3x3x3 layers of the headline path on the fp32 matrix cores with whole voxel rows
// read ONCE (round 2; the VALU / 32x32x2 kernels of conv_thin.hip remain the general-shape fallbacks).
//
//   head forward  OutputBlock.conv1  Conv3d(16|32 -> num_classes <= 5, k3 p1)      network/module/vnet_outblock.py:13
//   stem forward  InputBlock.conv    Conv3d(in_channels <= 8 -> 16, k3 p1)          network/module/vnet_inblock.py:9
//                 (and the head's data-gradient, the same op with 2..5 input channels)  -- persistent thin-input kernel below
//
// Head forward.  12 GFLOP over a 453 MB input (AI 25 FLOP/B): HBM-bound in principle, but a GEMM with only num_classes
// output columns wastes 30/32 of an MFMA tile and the LDS-tiled VALU kernel (conv_thin.hip) re-reads every 128-byte
// voxel row in four 32-byte passes (2.2 GB of traffic for 0.48 GB of algorithmic bytes, 323 us).  Formulation (the one
// the bf16 head kernel uses, here in exact fp32 on v_mfma_f32_16x16x4_f32):
//     P[v][(kz, ky, co)] = sum_{kx, ci} x[v + (kx - 1)][ci] * w[kz][ky][kx][ci][co]      K = 3 Cin, rows = (kz, ky, co)
//     y[z][y][x][co]     = sum_{kz, ky} P[(z + kz - 1, y + ky - 1, x)][(kz, ky, co)]
// The (kz, ky) taps fill the OUTPUT rows of the MFMA: with two classes 8 of the 9 taps are exactly the 16 rows of one
// 16x16x4 MFMA (no wasted rows), and the ninth tap (3 Cin x 2 products per voxel) runs on the VALU in the shadow of the
// matrix pipe; 3 / 5 classes use 2 / 3 row groups for all nine taps, 4 classes 2 groups + the VALU tap.
// A wave owns a column of 8 (y) x 16 (x) outputs and marches through tz z-planes.  One MFMA column is one voxel of a
// 16-voxel x row: lane = (x, kq), kq = lane >> 4 supplying K entries (channels kq * Cin/4 + j in MFMA j).  NDHWC makes a
// lane's channels one or two 16-byte loads; the x - 1 / x + 1 operands come from the neighbouring lanes (DPP row shifts,
// only the two end lanes load their outside neighbour): activations never pass through LDS and every voxel row is read
// once per (y, z) halo row.  P goes through a wave-private LDS plane; after each plane the lane adds the 3 x 3 shifted
// entries into three rolling output planes (kz = 0, 1, 2) and stores the finished one.  No workgroup barrier anywhere.
// Cost: (tz + 2) / tz * 10 / 8 halo rows per output row, 24 MFMAs of 32 cycles per 16 voxels (Cin 32, 2 classes).
#include "seg3d_common.h"
#include "seg3d_hip.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#define TF_TY 8
#define TF_TX 16
#define TF_HY (TF_TY + 2)

template <int CIN, int COUT> struct ThinOutF32 {
  static constexpr int CPL = CIN / 4;                                // channels per lane: kq * CPL + j, j < CPL
  static constexpr int NTM = (COUT == 2 || COUT == 4) ? 8 : 9;       // (kz, ky) taps on the matrix cores
  static constexpr int NR = (NTM * COUT + 15) / 16;                  // 16-row groups: 1, 2, 2, 3 for 2, 3, 4, 5 classes
  static constexpr int NV = NTM == 8 ? 4 * COUT : 0;                 // the VALU tap's four K-quarter partial sums
  static constexpr int STR = NR * 16 + NV + 4;                       // floats per voxel of a P plane: 28, 36, 52, 52 --
                                                                     // 8 consecutive voxels' 16-byte stores hit 32 banks
  static constexpr int ROW = TF_TX * STR + 4;
  static constexpr int PLANE = TF_HY * ROW;
  static constexpr int WA = NR * 3 * CPL * 64;                       // floats of the MFMA weight section
  static constexpr int WV = NTM == 8 ? 3 * CPL * COUT * 4 : 0;       // floats of the VALU tap section
};

static inline int thin_out_f32_cout_t(int Cout) { return Cout <= 2 ? 2 : Cout; }

// wpk = [rg][kx][j][lane] MFMA row operands, then [kx][j][co][kq] for the VALU tap (tap (kz, ky) = (2, 2))
__global__ __launch_bounds__(256) void pack_thin_out_f32mfma_kernel(const float* __restrict__ w, float* __restrict__ wpk, int B,
                                                                      int CPL, int COUT, int NTM, int WA, int total, i64 sa,
                                                                      i64 sb, int flip) {
  for (int idx = blockIdx.x * 256 + threadIdx.x; idx < total; idx += gridDim.x * 256) {
    int t, ci, co;
    bool ok;
    if (idx < WA) {
      const int lane = idx & 63, m = idx >> 6;
      const int j = m % CPL, kx = (m / CPL) % 3, rg = m / (3 * CPL);
      const int R = rg * 16 + (lane & 15);
      const int t9 = R / COUT;
      co = R - t9 * COUT;
      ci = (lane >> 4) * CPL + j;
      t = t9 * 3 + kx;
      ok = t9 < NTM && co < B;
    } else {
      const int e = idx - WA;
      const int kq = e & 3, r = e >> 2;
      co = r % COUT;
      const int m = r / COUT;
      const int j = m % CPL, kx = m / CPL;
      ci = kq * CPL + j;
      t = 24 + kx;
      ok = co < B;
    }
    wpk[idx] = ok ? w[ci * sa + co * sb + (flip ? 26 - t : t)] : 0.f;
  }
}

extern "C" int seg3d_conv3d_k3_thin_out_f32mfma_supported(int Cin, int Cout) {
  return (Cin == 16 || Cin == 32) && Cout >= 1 && Cout <= 5;
}

static void thin_out_f32_sizes(int Cin, int Cout, int* cpl, int* cout_t, int* ntm, int* wa, int* total) {
  const int CO = thin_out_f32_cout_t(Cout);
  const int NTM = (CO == 2 || CO == 4) ? 8 : 9;
  const int NR = (NTM * CO + 15) / 16;
  *cpl = Cin / 4;
  *cout_t = CO;
  *ntm = NTM;
  *wa = NR * 3 * (Cin / 4) * 64;
  *total = *wa + (NTM == 8 ? 3 * (Cin / 4) * CO * 4 : 0);
}

extern "C" long long seg3d_thin_out_f32mfma_packed_floats(int Cin, int Cout) {
  if (!seg3d_conv3d_k3_thin_out_f32mfma_supported(Cin, Cout)) return 0;
  int cpl, co, ntm, wa, total;
  thin_out_f32_sizes(Cin, Cout, &cpl, &co, &ntm, &wa, &total);
  return total;
}

extern "C" int seg3d_pack_weights_thin_out_f32mfma(const float* w, float* wpk, int A, int B, long long sa, long long sb,
                                                   int flip, void* stream) {
  SEG3D_REQUIRE(w && wpk && seg3d_conv3d_k3_thin_out_f32mfma_supported(A, B),
                "seg3d_pack_weights_thin_out_f32mfma: need Cin in {16, 32} and Cout <= 5");
  int cpl, co, ntm, wa, total;
  thin_out_f32_sizes(A, B, &cpl, &co, &ntm, &wa, &total);
  hipLaunchKernelGGL(pack_thin_out_f32mfma_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, w, wpk, B,
                     cpl, co, ntm, wa, total, (i64)sa, (i64)sb, flip);
  SEG3D_LAUNCH_CHECK("seg3d_pack_weights_thin_out_f32mfma");
  return SEG3D_OK;
}

// The fp32 MFMA executes on the fp32 VALU datapath (tools/ubench/mfma_valu: one v_fma_f32 beside a 32-cycle
// v_mfma_f32_16x16x4_f32 costs ~5 cycles MORE, beside a bf16 MFMA nothing): every vector instruction of the row loop is
// paid in full, so the loop carries no address arithmetic (uniform row base in SGPRs + a constant per-lane byte offset),
// no per-row selects (rows outside the volume skip their MFMAs and store zeros; lanes outside it exist only in the MASKED
// instantiation, chosen per tile) -- what is left per 16-voxel row: 24 MFMAs, 16 DPP moves, the VALU tap's 48 FMAs.
template <int CIN, int COUT, int RING, bool MASKED>
__device__ __forceinline__ void thin_out_f32mfma_tile(const float* __restrict__ x, const float* __restrict__ wpk,
                                                       const float* __restrict__ bias, float* __restrict__ y,
                                                       float* __restrict__ stats, float* __restrict__ pw, int D, int H, int W,
                                                       int Cout, int n, int z0, int y0, int x0, int nz, i64 stat_slot) {
  typedef ThinOutF32<CIN, COUT> T;
  constexpr int CPL = T::CPL, Q = CPL / 4, NR = T::NR, NTM = T::NTM, STR = T::STR, ROW = T::ROW;
  const int lane = threadIdx.x & 63, xl = lane & 15, kq = lane >> 4;

  // weights: the MFMA row operands (lane = (row r = lane & 15, k = kq)) and this lane's K quarter of the VALU tap
  float wa[NR][3][CPL];
#pragma unroll
  for (int rg = 0; rg < NR; ++rg)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int j = 0; j < CPL; ++j) wa[rg][kx][j] = wpk[(((rg * 3 + kx) * CPL + j) << 6) + lane];
  // the VALU tap runs on v_pk_fma_f32 (beside an MFMA it costs what one v_fma_f32 costs): channel PAIRS are packed,
  // wv[kx][t][c] = (w[2 t][c], w[2 t + 1][c]), and the two halves of an accumulator are added once per row
  f32x2 wv[3][CPL / 2][COUT];
  if (NTM == 8) {
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
      for (int t = 0; t < CPL / 2; ++t)
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          wv[kx][t][c][0] = wpk[T::WA + (((kx * CPL + 2 * t) * COUT + c) << 2) + kq];
          wv[kx][t][c][1] = wpk[T::WA + (((kx * CPL + 2 * t + 1) * COUT + c) << 2) + kq];
        }
  }

  // halo row (p, hy): plane gz = z0 + p - 1, row gy = y0 + hy - 1; the lane's voxel is x0 + xl, its channels kq * CPL ..
  // The two end lanes of a 16-lane x row also load their outside neighbour (x0 - 1 / x0 + 16); lanes whose voxel is
  // outside the row (MASKED tiles only) load the nearest voxel inside it and are zeroed when consumed.
  const int edge_dx = xl == 0 ? -1 : (xl == 15 ? 1 : 0);
  const bool centre_in = x0 + xl < W, edge_in = edge_dx != 0 && x0 + xl + edge_dx >= 0 && x0 + xl + edge_dx < W;
  const int cx = centre_in ? xl : W - 1 - x0;
  const int ex = edge_in ? xl + edge_dx : cx;
  const int coff = (cx * CIN + kq * CPL) * 4, eoff = (ex * CIN + kq * CPL) * 4;   // byte offsets from the row's first voxel
  auto row_base = [&](int p, int hy, bool& rowok) {     // wave-uniform
    const int gz = z0 + p - 1, gy = y0 + hy - 1;
    rowok = gz >= 0 && gz < D && gy >= 0 && gy < H && p <= nz + 1;
    // rows outside the volume (or past the tile) load the nearest row inside it: every lane offset stays in bounds
    const int gzc = gz < 0 ? 0 : (gz >= D ? D - 1 : gz), gyc = gy < 0 ? 0 : (gy >= H ? H - 1 : gy);
    const i64 vox = ((i64)(n * D + gzc) * H + gyc) * W + x0;
    return reinterpret_cast<const char*>(x + vox * CIN);
  };
  // Loads are unconditional: with no branch around them the waits in front of a row are counted vmcnt that leave the later
  // rows of the ring in flight.
  auto load4 = [&](const char* base, int off, int q) { return *reinterpret_cast<const f32x4*>(base + off + 16 * q); };

  // outputs of a lane: (ty, x = 2 xp, 2 xp + 1) of the planes in flight; out[k] is output plane p - k (taps kz = k)
  const int ty = lane >> 3, xp = lane & 7;
  float out[3][2][COUT];
#pragma unroll
  for (int k = 0; k < 3; ++k)
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < COUT; ++c) out[k][o][c] = 0.f;
  float bv[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) bv[c] = (bias && c < Cout) ? bias[c] : 0.f;
  float s[2] = {0.f, 0.f};
  const int oy = y0 + ty, ox = x0 + 2 * xp;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  f32x4 cc[RING][Q], ee[RING][Q];
#pragma unroll
  for (int sl = 0; sl < RING; ++sl) {
    bool ok0;
    const char* b0 = row_base(0, sl, ok0);
#pragma unroll
    for (int q = 0; q < Q; ++q) {
      cc[sl][q] = load4(b0, coff, q);
      ee[sl][q] = load4(b0, eoff, q);
    }
  }
#pragma unroll 1
  for (int p = 0; p <= nz + 1; ++p) {
#pragma unroll
    for (int hy = 0; hy < TF_HY; ++hy) {
      const int sl = hy % RING;
      bool cok, nok;
      row_base(p, hy, cok);
      const char* nb = row_base(p + (hy + RING) / TF_HY, (hy + RING) % TF_HY, nok);
      float* dst = pw + hy * ROW + xl * STR;
      f32x4 cen[Q], edg[Q];
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        cen[q] = cc[sl][q];
        edg[q] = ee[sl][q];
      }
      if (cok) {   // wave-uniform
        // operands: in[kx][t] = channels kq * CPL + 2 t, + 1 of voxel x + kx - 1
        f32x2 in[3][CPL / 2];
#pragma unroll
        for (int q = 0; q < Q; ++q) {
          if (MASKED) {
            cen[q] = centre_in ? cen[q] : zero4;
            edg[q] = edge_in ? edg[q] : zero4;
          }
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            const int cv = __float_as_int(cen[q][d]), ev = __float_as_int(edg[q][d]);
            in[0][2 * q + (d >> 1)][d & 1] = __int_as_float(__builtin_amdgcn_update_dpp(ev, cv, 0x111, 0xf, 0xf, false));   // row_shr:1: lane i <- i - 1
            in[1][2 * q + (d >> 1)][d & 1] = cen[q][d];
            in[2][2 * q + (d >> 1)][d & 1] = __int_as_float(__builtin_amdgcn_update_dpp(ev, cv, 0x101, 0xf, 0xf, false));   // row_shl:1: lane i <- i + 1
          }
        }
        f32x4 acc[NR][2];
#pragma unroll
        for (int rg = 0; rg < NR; ++rg) acc[rg][0] = acc[rg][1] = zero4;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
          for (int j = 0; j < CPL; ++j)
#pragma unroll
            for (int rg = 0; rg < NR; ++rg)   // two accumulation chains per row group: the MFMA's dependent latency is 40 cycles
              acc[rg][(kx * CPL + j) & 1] =
                  __builtin_amdgcn_mfma_f32_16x16x4f32(wa[rg][kx][j], in[kx][j >> 1][j & 1], acc[rg][(kx * CPL + j) & 1], 0, 0, 0);
        // result register i of row group rg is row rg * 16 + 4 kq + i (the (kz, ky, co) column of P) of voxel xl
#pragma unroll
        for (int rg = 0; rg < NR; ++rg) {
          const f32x4 v = acc[rg][0] + acc[rg][1];
          *reinterpret_cast<f32x4*>(dst + rg * 16 + 4 * kq) = v;
        }
        if (NTM == 8) {   // tap (kz, ky) = (2, 2): this lane's K quarter; the four quarters are added by the reader
          f32x2 v9[COUT];
#pragma unroll
          for (int c = 0; c < COUT; ++c) v9[c] = f32x2{0.f, 0.f};
#pragma unroll
          for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int t = 0; t < CPL / 2; ++t)
#pragma unroll
              for (int c = 0; c < COUT; ++c) v9[c] = __builtin_elementwise_fma(in[kx][t], wv[kx][t][c], v9[c]);
          if (COUT == 2) {
            const f32x2 v2 = {v9[0][0] + v9[0][1], v9[COUT - 1][0] + v9[COUT - 1][1]};
            *reinterpret_cast<f32x2*>(dst + NR * 16 + kq * COUT) = v2;
          } else {
            const f32x4 v4 = {v9[0][0] + v9[0][1], v9[1 % COUT][0] + v9[1 % COUT][1], v9[2 % COUT][0] + v9[2 % COUT][1],
                              v9[3 % COUT][0] + v9[3 % COUT][1]};
            *reinterpret_cast<f32x4*>(dst + NR * 16 + kq * COUT) = v4;
          }
        }
      } else {     // a row outside the volume: P = 0
#pragma unroll
        for (int rg = 0; rg < NR; ++rg) *reinterpret_cast<f32x4*>(dst + rg * 16 + 4 * kq) = zero4;
        if (NTM == 8) {
          if (COUT == 2) *reinterpret_cast<f32x2*>(dst + NR * 16 + kq * COUT) = f32x2{0.f, 0.f};
          else *reinterpret_cast<f32x4*>(dst + NR * 16 + kq * COUT) = zero4;
        }
      }
      // row (p, hy) + RING into the registers this row has just finished with (issued after its MFMAs: a load that is
      // issued earlier needs a second register set and a copy per register per row)
#pragma unroll
      for (int q = 0; q < Q; ++q) {
        cc[sl][q] = load4(nb, coff, q);
        ee[sl][q] = load4(nb, eoff, q);
      }
    }
    // halo plane p is complete in LDS (wave-private: ordering inside the wave is all that is needed)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
    for (int kz = 0; kz < 3; ++kz) {
      if (p - kz >= 0 && p - kz < nz) {   // wave-uniform
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
          const float* src = pw + (ty + ky) * ROW + (2 * xp) * STR;
          if (kz * 3 + ky < NTM) {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              const float* pp = src + o * STR + (kz * 3 + ky) * COUT;
              if (COUT == 2) {          // 8- / 16-byte aligned by construction (STR, ROW multiples of 4)
                const f32x2 v = *reinterpret_cast<const f32x2*>(pp);
                out[kz][o][0] += v[0];
                out[kz][o][COUT - 1] += v[1];
              } else if (COUT == 4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(pp);
#pragma unroll
                for (int c = 0; c < COUT; ++c) out[kz][o][c] += v[c & 3];
              } else {
#pragma unroll
                for (int c = 0; c < COUT; ++c) out[kz][o][c] += pp[c];
              }
            }
          } else {
#pragma unroll
            for (int o = 0; o < 2; ++o) {
              const float* pv = src + o * STR + NR * 16;     // [kq][co]
              if (COUT == 2) {
                const f32x4 a = *reinterpret_cast<const f32x4*>(pv), bq = *reinterpret_cast<const f32x4*>(pv + 4);
                out[kz][o][0] += (a[0] + a[2]) + (bq[0] + bq[2]);
                out[kz][o][COUT - 1] += (a[1] + a[3]) + (bq[1] + bq[3]);
              } else {
#pragma unroll
                for (int c = 0; c < COUT; ++c) out[kz][o][c] += (pv[c] + pv[COUT + c]) + (pv[2 * COUT + c] + pv[3 * COUT + c]);
              }
            }
          }
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const int oz = z0 + p - 2;   // output plane p - 2 has all three kz taps now
    if (p >= 2 && oy < H) {
      float* yp = y + ((((i64)n * D + oz) * H + oy) * W + ox) * Cout;
      float val[2][COUT];
#pragma unroll
      for (int o = 0; o < 2; ++o)
#pragma unroll
        for (int c = 0; c < COUT; ++c) val[o][c] = out[2][o][c] + bv[c];
      if (COUT == 2 && Cout == 2 && ox + 1 < W && (W & 1) == 0) {   // 16 bytes, aligned (even row length, even x)
        const f32x4 v = {val[0][0], val[0][COUT - 1], val[1][0], val[1][COUT - 1]};
        *reinterpret_cast<f32x4*>(yp) = v;
#pragma unroll
        for (int o = 0; o < 2; ++o)
#pragma unroll
          for (int c = 0; c < COUT; ++c) {
            s[0] += val[o][c];
            s[1] += val[o][c] * val[o][c];
          }
      } else {
#pragma unroll
        for (int o = 0; o < 2; ++o)
          if (ox + o < W) {
#pragma unroll
            for (int c = 0; c < COUT; ++c)
              if (c < Cout) {
                yp[o * Cout + c] = val[o][c];
                s[0] += val[o][c];
                s[1] += val[o][c] * val[o][c];
              }
          }
      }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
#pragma unroll
      for (int c = 0; c < COUT; ++c) {
        out[2][o][c] = out[1][o][c];
        out[1][o][c] = out[0][o][c];
        out[0][o][c] = 0.f;
      }
  }

  if (stats) {   // one slot per tile = per wave
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
      s[0] += __shfl_xor(s[0], m, 64);
      s[1] += __shfl_xor(s[1], m, 64);
    }
    if (lane == 0) {
      stats[stat_slot * 2 + 0] = s[0];
      stats[stat_slot * 2 + 1] = s[1];
    }
  }
}

template <int CIN, int COUT, int RING>
__global__ __launch_bounds__(256, (COUT == 2 && RING <= 5) ? 2 : 1) void conv3d_k3_thin_out_f32mfma_kernel(
    const float* __restrict__ x, const float* __restrict__ wpk, const float* __restrict__ bias, float* __restrict__ y,
    float* __restrict__ stats, int N, int D, int H, int W, int Cout, int tz, int ntz, int nty, int ntx) {
  typedef ThinOutF32<CIN, COUT> T;
  __shared__ __attribute__((aligned(16))) float pl[4 * T::PLANE];
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int ntiles = N * ntz * nty * ntx;
  const int wg = seg3d_xcd_tile(blockIdx.x, (ntiles + 3) >> 2);   // neighbouring tiles (shared halo rows) on one XCD
  if (wg < 0) return;
  int b = wg * 4 + wave;                                          // the wave's tile (x fastest); everything below is uniform
  if (b >= ntiles) return;
  const int tix = b % ntx; b /= ntx;
  const int tiy = b % nty; b /= nty;
  const int tiz = b % ntz;
  const int n = b / ntz;
  const int z0 = tiz * tz, y0 = tiy * TF_TY, x0 = tix * TF_TX;
  const int nz = (D - z0) < tz ? (D - z0) : tz;                   // output planes of this tile
  float* pw = pl + wave * T::PLANE;
  const i64 slot = (i64)n * (ntz * nty * ntx) + (tiz * nty + tiy) * ntx + tix;
  if (x0 == 0 || x0 + TF_TX >= W)
    thin_out_f32mfma_tile<CIN, COUT, RING, true>(x, wpk, bias, y, stats, pw, D, H, W, Cout, n, z0, y0, x0, nz, slot);
  else
    thin_out_f32mfma_tile<CIN, COUT, RING, false>(x, wpk, bias, y, stats, pw, D, H, W, Cout, n, z0, y0, x0, nz, slot);
}

// z extent of a wave's tile.  A SIMD's time is the sum over its waves of (tz + 2) halo planes x ~1300 cycles of fp32 work
// per 16-voxel row (MFMA and VALU share one datapath) plus ~270 cycles of memory / LDS waits per row that only a second
// wave on the SIMD hides (measured at 4 x 96^3: tz 16 245 us, tz 32 257 us, tz 8 262 us).
static int thin_out_f32_tz(int N, int D, int H, int W) {
  const int nty = seg3d_cdiv(H, TF_TY), ntx = seg3d_cdiv(W, TF_TX);
  const int cand[] = {8, 12, 16, 24, 32, 48};
  int best = 8;
  double best_cost = 1e30;
  for (int tz : cand) {
    if (tz > 8 && tz > D) continue;
    const i64 tiles = (i64)N * seg3d_cdiv(D, tz) * nty * ntx;
    const double cost = (double)((tiles + 1023) / 1024) * (tz + 2.5) * (tiles <= 1024 ? 1.2 : 1.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = tz;
    }
  }
  return best;
}

extern "C" long long seg3d_conv3d_k3_thin_out_f32mfma_stats_count(int N, int D, int H, int W) {
  const int tz = thin_out_f32_tz(N, D, H, W);
  return (long long)seg3d_cdiv(D, tz) * seg3d_cdiv(H, TF_TY) * seg3d_cdiv(W, TF_TX);
}

// x fp32 [N][D][H][W][Cin] (Cin in {16, 32}), wpk = seg3d_pack_weights_thin_out_f32mfma, y fp32 [N][D][H][W][Cout],
// Cout <= 5; stats (optional): [N][seg3d_conv3d_k3_thin_out_f32mfma_stats_count(N, D, H, W)][2]
extern "C" int seg3d_conv3d_k3_thin_out_f32mfma_fwd(const float* x, const float* wpk, const float* bias, float* y,
                                                    float* stats, int N, int D, int H, int W, int Cin, int Cout,
                                                    void* stream) {
  SEG3D_REQUIRE(x && wpk && y, "seg3d_conv3d_k3_thin_out_f32mfma_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0, "seg3d_conv3d_k3_thin_out_f32mfma_fwd: bad dims");
  SEG3D_REQUIRE(seg3d_conv3d_k3_thin_out_f32mfma_supported(Cin, Cout),
                "seg3d_conv3d_k3_thin_out_f32mfma_fwd: need Cin in {16, 32} and Cout <= 5");
  SEG3D_REQUIRE((i64)N * D * H * W * Cin < (1ll << 31), "seg3d_conv3d_k3_thin_out_f32mfma_fwd: tensor exceeds 2^31 elements");
  const int tz = thin_out_f32_tz(N, D, H, W);
  const int ntz = seg3d_cdiv(D, tz), nty = seg3d_cdiv(H, TF_TY), ntx = seg3d_cdiv(W, TF_TX);
  SEG3D_REQUIRE((i64)N * ntz * nty * ntx < SEG3D_FDIV_MAX, "seg3d_conv3d_k3_thin_out_f32mfma_fwd: more than 2^22 tiles");
  dim3 grid((unsigned)seg3d_xcd_grid((N * ntz * nty * ntx + 3) / 4));   // one tile per wave, four waves per workgroup
  hipStream_t s = (hipStream_t)stream;
#define SEG3D_TOF(CI, CO, RG)                                                                                             \
  hipLaunchKernelGGL((conv3d_k3_thin_out_f32mfma_kernel<CI, CO, RG>), grid, dim3(256), 0, s, x, wpk, bias, y, stats, N, D, H, \
                     W, Cout, tz, ntz, nty, ntx)
  const int CO = thin_out_f32_cout_t(Cout);
  if (Cin == 32) {
    if (CO == 2) SEG3D_TOF(32, 2, 5);
    else if (CO == 3) SEG3D_TOF(32, 3, 2);
    else if (CO == 4) SEG3D_TOF(32, 4, 2);
    else SEG3D_TOF(32, 5, 2);
  } else {
    if (CO == 2) SEG3D_TOF(16, 2, 5);
    else if (CO == 3) SEG3D_TOF(16, 3, 5);
    else if (CO == 4) SEG3D_TOF(16, 4, 5);
    else SEG3D_TOF(16, 5, 5);
  }
#undef SEG3D_TOF
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_out_f32mfma_fwd");
  return SEG3D_OK;
}

// ---------------------------------------------------------------------------------------------------------------
// thin-input conv, persistent form (stem forward CT = in_channels, head data-gradient CT = classes; fp32)
// ---------------------------------------------------------------------------------------------------------------
// Same arithmetic as conv3d_k3_thin_in_kernel (conv_thin.hip): K = 27 CT folded into one MFMA reduction, weights as the ROW
// operand in registers (packed by seg3d_pack_weights_thin_in), a lane ends up with one voxel and quads of consecutive
// output channels.  That kernel ran one 256-voxel tile per workgroup and spent ~490 vector instructions per wave and tile
// (index decode of the staged halo entries, per-step operand offsets, weight loads, block-wide statistics) beside 28 / 54
// MFMAs -- and fp32 MFMA and VALU share the datapath, so it ran at 0.30 of HBM.  Here a workgroup walks tiles:
// weights, staging coordinates (packed), face flags, relative offsets and the K steps' LDS offsets are computed once; the
// next tile's halo loads are in flight behind the current tile's MFMAs; statistics are one slot per WAVE (no block reduce).
#define TP_TZ 4
#define TP_TY 8
#define TP_TX 8
#define TP_MT (TP_TZ * TP_TY * TP_TX)
#define TP_HY (TP_TY + 2)
#define TP_HX (TP_TX + 2)
#define TP_NV ((TP_TZ + 2) * TP_HY * TP_HX)  // 600
#define TP_ROW 36                             // floats per voxel row of the epilogue transpose (32 + 4: conflict-free b128 writes)

typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CT>
__device__ __forceinline__ int thinp_koff(int k) {  // LDS float offset of GEMM-k inside the halo tile
  const int t = k / CT, a = k % CT;
  const int kz = t / 9, ky = (t / 3) % 3, kx = t % 3;
  return t < 27 ? ((kz * TP_HY + ky) * TP_HX + kx) * CT + a : 0;
}

extern "C" long long seg3d_conv3d_k3_thin_in_persistent_stats_count(int D, int H, int W, int Cout_blocks) {
  return 4ll * seg3d_cdiv(D, TP_TZ) * seg3d_cdiv(H, TP_TY) * seg3d_cdiv(W, TP_TX) * Cout_blocks;
}

template <int CT, bool OUT_BF>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_in_persistent_kernel(const float* __restrict__ x,
                                                                                const float* __restrict__ wp,
                                                                                const float* __restrict__ bias,
                                                                                float* __restrict__ y, float* __restrict__ stats,
                                                                                int N, int D, int H, int W, int Cout, int ntz,
                                                                                int nty, int ntx, int ntiles) {
  constexpr int KP = (27 * CT + 1) / 2;
  constexpr int TE = (TP_NV * CT + 255) / 256;
  __shared__ __attribute__((aligned(16))) float xs[TP_NV * CT + 4];
  __shared__ __attribute__((aligned(16))) float tbuf[OUT_BF ? 4 : 4 * 32 * TP_ROW];   // epilogue transpose, per wave [voxel 32][36]
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, lh = lane >> 5;
  const int cob = blockIdx.y;
  const bool regular = (D % TP_TZ) == 0 && (H % TP_TY) == 0 && (W % TP_TX) == 0;

  // weights of this lane's row for every k pair, kept in registers
  float bw[KP];
  {
    const float* wsrc = wp + ((i64)cob * KP * 2 + lh) * 32 + li;
#pragma unroll
    for (int p = 0; p < KP; ++p) bw[p] = wsrc[p * 64];
  }
  // staged halo entries of this thread (tile-invariant): packed coordinates, element offset from the tile's first voxel,
  // halo faces (bit 0..5 = z lo, z hi, y lo, y hi, x lo, x hi; bit 6 = no such entry)
  int tpos[TE], trel[TE], tface[TE];
#pragma unroll
  for (int k = 0; k < TE; ++k) {
    const int e = tid + k * 256;
    const int v = e / CT, a = e % CT;
    const int hx = v % TP_HX;
    const int t = v / TP_HX;
    const int hy = t % TP_HY, hz = t / TP_HY;
    const bool have = e < TP_NV * CT;
    tpos[k] = have ? ((hz << 20) | (hy << 10) | hx) : -1;
    trel[k] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * CT + a;
    tface[k] = have ? ((hz == 0 ? 1 : 0) | (hz == TP_TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TP_TY + 1 ? 8 : 0) |
                       (hx == 0 ? 16 : 0) | (hx == TP_TX + 1 ? 32 : 0)) : 64;
  }
  // this lane's two voxels (row blocks wave, wave + 4): LDS base of their halo position, offset in the output tile
  int abase[2], vrel[2], vpos[2];
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    const int idx = (wave + 4 * m) * 32 + li;
    const int tx = idx % TP_TX;
    const int t = idx / TP_TX;
    const int ty = t % TP_TY, tz = t / TP_TY;
    abase[m] = ((tz * TP_HY + ty) * TP_HX + tx) * CT;
    vrel[m] = (tz * H + ty) * W + tx;
    vpos[m] = (tz << 20) | (ty << 10) | tx;
  }
  // K step p: lane half lh supplies k = 2 p + lh.  Even CT: the two halves are the two channels (a, a + 1) of one tap, i.e.
  // LDS offsets (compile-time) + lh; odd CT: a per-lane table.
  int koffs[(CT & 1) ? KP : 1];
  if (CT & 1) {
#pragma unroll
    for (int p = 0; p < KP; ++p) koffs[p] = lh ? thinp_koff<CT>(2 * p + 1) : thinp_koff<CT>(2 * p);
  }
  const float4 zero4 = make_float4(0.f, 0.f, 0.f, 0.f);
  (void)zero4;

  float tst[TE];
  unsigned okmask = 0;
  auto load_tile = [&](int tile, int& n_out, int& z0_out, int& y0_out, int& x0_out) {   // tile is wave-uniform
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * TP_TZ, y0 = tiy * TP_TY, x0 = tix * TP_TX;
    n_out = n; z0_out = z0; y0_out = y0; x0_out = x0;
    const i64 vox0 = (i64)((n * D + z0) * H + y0) * W + x0;
    okmask = 0;
    if (regular) {
      const int faces = 64 | (z0 == 0 ? 1 : 0) | (z0 + TP_TZ >= D ? 2 : 0) | (y0 == 0 ? 4 : 0) | (y0 + TP_TY >= H ? 8 : 0) |
                        (x0 == 0 ? 16 : 0) | (x0 + TP_TX >= W ? 32 : 0);
      const float* tbase = x + vox0 * CT;
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const bool ok = (tface[k] & faces) == 0;
        tst[k] = ok ? tbase[trel[k]] : x[0];
        okmask |= (ok ? 1u : 0u) << k;
      }
    } else {
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const int hz = (tpos[k] >> 20) & 255, hy = (tpos[k] >> 10) & 1023, hx = tpos[k] & 1023;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = tpos[k] >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        tst[k] = x[ok ? vox0 * CT + trel[k] : (i64)0];
        okmask |= (ok ? 1u : 0u) << k;
      }
    }
  };

  const int tile0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x), tstep = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  int nn = 0, nz0 = 0, ny0 = 0, nx0 = 0;
  if (tile0 < ntiles) load_tile(tile0, nn, nz0, ny0, nx0);
  const int tiles_per_sample = ntz * nty * ntx;
  const bool fast = regular && (Cout & 7) == 0;
  const int ng = (Cout - cob * 32) >= 32 ? 4 : (Cout - cob * 32) >> 3;   // channel quads per lane (fast path)
  f32x4 bv[4];   // this lane's bias values, quad g = channels cob * 32 + 8 g + 4 lh ..
#pragma unroll
  for (int g = 0; g < 4; ++g)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int co = cob * 32 + 8 * g + 4 * lh + j;
      bv[g][j] = (bias && co < Cout) ? bias[co] : 0.f;
    }
  for (int tile = tile0; tile < ntiles; tile += tstep) {
    const int n = nn, z0 = nz0, y0 = ny0, x0 = nx0;
    __syncthreads();   // every wave is done reading the previous tile
#pragma unroll
    for (int k = 0; k < TE; ++k) {
      const int e = tid + k * 256;
      if (e < TP_NV * CT) xs[e] = ((okmask >> k) & 1u) ? tst[k] : 0.f;
    }
    __syncthreads();
    if (tile + tstep < ntiles) load_tile(tile + tstep, nn, nz0, ny0, nx0);
    f32x16 acc[2];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[m][r] = 0.f;
#pragma unroll
    for (int p = 0; p < KP; ++p) {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        float bval;
        if (CT & 1) bval = xs[abase[m] + koffs[p]];
        else bval = xs[abase[m] + lh + thinp_koff<CT>(2 * p)];
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x2f32(bw[p], bval, acc[m], 0, 0, 0);
      }
    }
    // epilogue: register quad g of row block m = voxel (lane's column), channels cob * 32 + 8 g + 4 lh ..
    float s0 = 0.f, s1 = 0.f;
    const i64 vox0 = (i64)((n * D + z0) * H + y0) * W + x0;
    if (fast) {   // whole tiles, Cout % 8 == 0: straight-line, ng quads per lane (wave-uniform)
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        seg3d_bf16* yo16 = reinterpret_cast<seg3d_bf16*>(y) + (OUT_BF ? (vox0 + vrel[m]) * Cout + cob * 32 + 4 * lh : 0);
        float* tr = tbuf + wave * (32 * TP_ROW);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (g < ng) {
            const f32x4 v4 = {acc[m][4 * g] + bv[g][0], acc[m][4 * g + 1] + bv[g][1], acc[m][4 * g + 2] + bv[g][2],
                              acc[m][4 * g + 3] + bv[g][3]};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
              s0 += v4[j];
              s1 = fmaf(v4[j], v4[j], s1);
            }
            if (OUT_BF) {
              uint2 pk;
              pk.x = seg3d_pack2bf(v4[0], v4[1]);
              pk.y = seg3d_pack2bf(v4[2], v4[3]);
              *reinterpret_cast<uint2*>(yo16 + 8 * g) = pk;
            } else {
              *reinterpret_cast<f32x4*>(tr + li * TP_ROW + 8 * g + 4 * lh) = v4;
            }
          }
        }
        if constexpr (!OUT_BF) {
          // fp32 rows leave through an LDS transpose (round 4): a lane owned 16-byte pieces of ITS voxel's row, so a store
          // instruction wrote 64 pieces 128 bytes apart (tools/ubench/stride_load.hip: 3.6 TB/s); transposed, eight lanes write a
          // voxel's 128 bytes and the eight voxels of an x row of the tile are one contiguous kilobyte per instruction
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
          for (int it = 0; it < 4; ++it) {
            const int v = it * 8 + (lane >> 3), piece = lane & 7;
            const int idx = (wave + 4 * m) * 32 + v;
            const int tx = idx % TP_TX, t2 = idx / TP_TX;
            const int vr = ((t2 / TP_TY) * H + (t2 % TP_TY)) * W + tx;
            if (piece < 2 * ng)
              *reinterpret_cast<f32x4*>(y + (vox0 + vr) * Cout + cob * 32 + 4 * piece) = *reinterpret_cast<const f32x4*>(tr + v * TP_ROW + 4 * piece);
          }
          __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
      }
    } else {
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        const int tz = vpos[m] >> 20, ty = (vpos[m] >> 10) & 1023, tx = vpos[m] & 1023;
        const bool vok = z0 + tz < D && y0 + ty < H && x0 + tx < W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const int co0 = cob * 32 + 8 * g + 4 * lh;
          if (co0 >= Cout) continue;
          float val[4];
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool cok = co0 + j < Cout;
            val[j] = acc[m][4 * g + j] + bv[g][j];
            const float sv = (vok && cok) ? val[j] : 0.f;
            s0 += sv;
            s1 = fmaf(sv, sv, s1);
          }
          if (!vok) continue;
          const i64 o = (vox0 + vrel[m]) * Cout + co0;
#pragma unroll
          for (int j = 0; j < 4; ++j)
            if (co0 + j < Cout) {
              if (OUT_BF) reinterpret_cast<seg3d_bf16*>(y)[o + j] = seg3d_f2bf(val[j]);
              else y[o + j] = val[j];
            }
        }
      }
    }
    if (stats) {   // one slot per wave
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) {
        s0 += __shfl_xor(s0, mm, 64);
        s1 += __shfl_xor(s1, mm, 64);
      }
      if (lane == 0) {
        const int tl = tile - n * tiles_per_sample;
        float* dst = stats + ((((i64)n * tiles_per_sample + tl) * gridDim.y + cob) * 4 + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
  }
}

// The same walk for Cout <= 16 (the stem: 16 channels) on v_mfma_f32_16x16x4_f32: the 32-row MFMA above is half empty
// there, and since the fp32 MFMA shares the VALU datapath its 28 x 64 cycles per 64 voxels are real time (40 us of the
// stem's 95).  Rows = the 16 output channels, a wave's 64 voxels are four 16-column groups, K steps of 4; a lane ends up
// with 4 consecutive channels of one voxel per group, so the four lanes of a voxel store its whole 64-byte row at once.
template <int CT, bool OUT_BF>
__global__ __launch_bounds__(256, 2) void conv3d_k3_thin_in_persistent16_kernel(const float* __restrict__ x,
                                                                                  const float* __restrict__ wp,
                                                                                  const float* __restrict__ bias,
                                                                                  float* __restrict__ y, float* __restrict__ stats,
                                                                                  int N, int D, int H, int W, int Cout, int ntz,
                                                                                  int nty, int ntx, int ntiles) {
  constexpr int KP = (27 * CT + 1) / 2;     // k pairs of the packed weight image
  constexpr int K16 = (27 * CT + 3) / 4;    // K steps of this kernel
  constexpr int TE = (TP_NV * CT + 255) / 256;
  __shared__ __attribute__((aligned(16))) float xs[TP_NV * CT + 4];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int col = lane & 15, kq = lane >> 4;
  const bool regular = (D % TP_TZ) == 0 && (H % TP_TY) == 0 && (W % TP_TX) == 0;

  // weights: row `col` (output channel), k = 4 p + kq, taken from the pair image [k >> 1][k & 1][32]; LDS offset of that k
  float bw[K16];
  int koffs[K16];
#pragma unroll
  for (int p = 0; p < K16; ++p) {
    const int k = 4 * p + kq;
    bw[p] = k < 2 * KP ? wp[(k >> 1) * 64 + (k & 1) * 32 + col] : 0.f;
    koffs[p] = kq == 0 ? thinp_koff<CT>(4 * p) : kq == 1 ? thinp_koff<CT>(4 * p + 1) : kq == 2 ? thinp_koff<CT>(4 * p + 2)
                                                                                                 : thinp_koff<CT>(4 * p + 3);
  }
  int tpos[TE], trel[TE], tface[TE];
#pragma unroll
  for (int k = 0; k < TE; ++k) {
    const int e = tid + k * 256;
    const int v = e / CT, a = e % CT;
    const int hx = v % TP_HX;
    const int t = v / TP_HX;
    const int hy = t % TP_HY, hz = t / TP_HY;
    const bool have = e < TP_NV * CT;
    tpos[k] = have ? ((hz << 20) | (hy << 10) | hx) : -1;
    trel[k] = (((hz - 1) * H + (hy - 1)) * W + (hx - 1)) * CT + a;
    tface[k] = have ? ((hz == 0 ? 1 : 0) | (hz == TP_TZ + 1 ? 2 : 0) | (hy == 0 ? 4 : 0) | (hy == TP_TY + 1 ? 8 : 0) |
                       (hx == 0 ? 16 : 0) | (hx == TP_TX + 1 ? 32 : 0)) : 64;
  }
  // this lane's four voxels: 64 wave + 16 cg + col
  int abase[4], vrel[4], vpos[4];
#pragma unroll
  for (int cg = 0; cg < 4; ++cg) {
    const int idx = wave * 64 + cg * 16 + col;
    const int tx = idx % TP_TX;
    const int t = idx / TP_TX;
    const int ty = t % TP_TY, tz = t / TP_TY;
    abase[cg] = ((tz * TP_HY + ty) * TP_HX + tx) * CT;
    vrel[cg] = (tz * H + ty) * W + tx;
    vpos[cg] = (tz << 20) | (ty << 10) | tx;
  }

  float tst[TE];
  unsigned okmask = 0;
  auto load_tile = [&](int tile, int& n_out, int& z0_out, int& y0_out, int& x0_out) {   // tile is wave-uniform
    int b = tile;
    const int tix = b % ntx; b /= ntx;
    const int tiy = b % nty; b /= nty;
    const int tiz = b % ntz;
    const int n = b / ntz;
    const int z0 = tiz * TP_TZ, y0 = tiy * TP_TY, x0 = tix * TP_TX;
    n_out = n; z0_out = z0; y0_out = y0; x0_out = x0;
    const i64 vox0 = (i64)((n * D + z0) * H + y0) * W + x0;
    okmask = 0;
    if (regular) {
      const int faces = 64 | (z0 == 0 ? 1 : 0) | (z0 + TP_TZ >= D ? 2 : 0) | (y0 == 0 ? 4 : 0) | (y0 + TP_TY >= H ? 8 : 0) |
                        (x0 == 0 ? 16 : 0) | (x0 + TP_TX >= W ? 32 : 0);
      const float* tbase = x + vox0 * CT;
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const bool ok = (tface[k] & faces) == 0;
        tst[k] = ok ? tbase[trel[k]] : x[0];
        okmask |= (ok ? 1u : 0u) << k;
      }
    } else {
#pragma unroll
      for (int k = 0; k < TE; ++k) {
        const int hz = (tpos[k] >> 20) & 255, hy = (tpos[k] >> 10) & 1023, hx = tpos[k] & 1023;
        const int gz = z0 + hz - 1, gy = y0 + hy - 1, gx = x0 + hx - 1;
        const bool ok = tpos[k] >= 0 && gz >= 0 && gz < D && gy >= 0 && gy < H && gx >= 0 && gx < W;
        tst[k] = x[ok ? vox0 * CT + trel[k] : (i64)0];
        okmask |= (ok ? 1u : 0u) << k;
      }
    }
  };

  const int tile0 = __builtin_amdgcn_readfirstlane((int)blockIdx.x), tstep = __builtin_amdgcn_readfirstlane((int)gridDim.x);
  int nn = 0, nz0 = 0, ny0 = 0, nx0 = 0;
  if (tile0 < ntiles) load_tile(tile0, nn, nz0, ny0, nx0);
  const int tiles_per_sample = ntz * nty * ntx;
  const bool chan_ok = 4 * kq < Cout;     // Cout % 4 == 0 (host-checked): the lane's channel quad is inside or outside
  f32x4 bv = {0.f, 0.f, 0.f, 0.f};
  if (bias && chan_ok) bv = *reinterpret_cast<const f32x4*>(bias + 4 * kq);
  for (int tile = tile0; tile < ntiles; tile += tstep) {
    const int n = nn, z0 = nz0, y0 = ny0, x0 = nx0;
    __syncthreads();   // every wave is done reading the previous tile
#pragma unroll
    for (int k = 0; k < TE; ++k) {
      const int e = tid + k * 256;
      if (e < TP_NV * CT) xs[e] = ((okmask >> k) & 1u) ? tst[k] : 0.f;
    }
    __syncthreads();
    if (tile + tstep < ntiles) load_tile(tile + tstep, nn, nz0, ny0, nx0);
    f32x4 acc[4];
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) acc[cg] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int p = 0; p < K16; ++p)
#pragma unroll
      for (int cg = 0; cg < 4; ++cg)
        acc[cg] = __builtin_amdgcn_mfma_f32_16x16x4f32(bw[p], xs[abase[cg] + koffs[p]], acc[cg], 0, 0, 0);
    float s0 = 0.f, s1 = 0.f;
    const i64 vox0 = (i64)((n * D + z0) * H + y0) * W + x0;
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
      const int tz = vpos[cg] >> 20, ty = (vpos[cg] >> 10) & 1023, tx = vpos[cg] & 1023;
      const bool vok = chan_ok && (regular || (z0 + tz < D && y0 + ty < H && x0 + tx < W));
      const f32x4 v4 = acc[cg] + bv;
      if (vok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          s0 += v4[j];
          s1 = fmaf(v4[j], v4[j], s1);
        }
        const i64 o = (vox0 + vrel[cg]) * Cout + 4 * kq;
        if (OUT_BF) {
          uint2 pk;
          pk.x = seg3d_pack2bf(v4[0], v4[1]);
          pk.y = seg3d_pack2bf(v4[2], v4[3]);
          *reinterpret_cast<uint2*>(reinterpret_cast<seg3d_bf16*>(y) + o) = pk;
        } else {
          *reinterpret_cast<f32x4*>(y + o) = v4;
        }
      }
    }
    if (stats) {   // one slot per wave
#pragma unroll
      for (int mm = 32; mm >= 1; mm >>= 1) {
        s0 += __shfl_xor(s0, mm, 64);
        s1 += __shfl_xor(s1, mm, 64);
      }
      if (lane == 0) {
        const int tl = tile - n * tiles_per_sample;
        float* dst = stats + (((i64)n * tiles_per_sample + tl) * 4 + wave) * 2;
        dst[0] = s0;
        dst[1] = s1;
      }
    }
  }
}

template <int CT>
static void launch_thin_in_persistent(const float* x, const float* wp, const float* bias, float* y, float* stats, int N, int D,
                                      int H, int W, int Cout, hipStream_t s, int out_bf16) {
  const int ntz = seg3d_cdiv(D, TP_TZ), nty = seg3d_cdiv(H, TP_TY), ntx = seg3d_cdiv(W, TP_TX);
  const int ntiles = N * ntz * nty * ntx;
  const int cobs = (Cout + 31) / 32;
  int wgs = 1024 / cobs;               // ~4 workgroups per CU over the whole grid (LDS 2.4 KB x CT, <= 128 registers)
  if (wgs > ntiles) wgs = ntiles;
  if (wgs < 1) wgs = 1;
  dim3 grid((unsigned)wgs, (unsigned)cobs);
  if (Cout <= 16 && (Cout & 3) == 0) {   // 16-row MFMA: no empty rows, whole 64-byte voxel rows per store
    if (out_bf16)
      hipLaunchKernelGGL((conv3d_k3_thin_in_persistent16_kernel<CT, true>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H,
                         W, Cout, ntz, nty, ntx, ntiles);
    else
      hipLaunchKernelGGL((conv3d_k3_thin_in_persistent16_kernel<CT, false>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H,
                         W, Cout, ntz, nty, ntx, ntiles);
    return;
  }
  if (out_bf16)
    hipLaunchKernelGGL((conv3d_k3_thin_in_persistent_kernel<CT, true>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H, W,
                       Cout, ntz, nty, ntx, ntiles);
  else
    hipLaunchKernelGGL((conv3d_k3_thin_in_persistent_kernel<CT, false>), grid, dim3(256), 0, s, x, wp, bias, y, stats, N, D, H, W,
                       Cout, ntz, nty, ntx, ntiles);
}

// x [N][D][H][W][CT] (CT <= 8), wp = seg3d_pack_weights_thin_in, y [N][D][H][W][Cout] (fp32, or bf16 storage when out_bf16);
// stats (optional): [N][seg3d_conv3d_k3_thin_in_persistent_stats_count(D, H, W, ceil(Cout / 32))][2]
extern "C" int seg3d_conv3d_k3_thin_in_persistent_fwd(const float* x, const float* wp, const float* bias, void* y, float* stats,
                                                      int N, int D, int H, int W, int CT, int Cout, int out_bf16, void* stream) {
  SEG3D_REQUIRE(x && wp && y, "seg3d_conv3d_k3_thin_in_persistent_fwd: null pointer");
  SEG3D_REQUIRE(N > 0 && D > 0 && H > 0 && W > 0 && Cout > 0, "seg3d_conv3d_k3_thin_in_persistent_fwd: bad dims");
  SEG3D_REQUIRE(CT >= 1 && CT <= 8, "seg3d_conv3d_k3_thin_in_persistent_fwd: thin channel count %d not in [1, 8]", CT);
  SEG3D_REQUIRE((i64)N * D * H * W * Cout < (1ll << 31) && (i64)N * D * H * W * CT < (1ll << 31),
                "seg3d_conv3d_k3_thin_in_persistent_fwd: tensor exceeds 2^31 elements");
  hipStream_t s = (hipStream_t)stream;
  float* yf = reinterpret_cast<float*>(y);
  switch (CT) {
    case 1: launch_thin_in_persistent<1>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 2: launch_thin_in_persistent<2>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 3: launch_thin_in_persistent<3>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 4: launch_thin_in_persistent<4>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 5: launch_thin_in_persistent<5>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 6: launch_thin_in_persistent<6>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    case 7: launch_thin_in_persistent<7>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
    default: launch_thin_in_persistent<8>(x, wp, bias, yf, stats, N, D, H, W, Cout, s, out_bf16); break;
  }
  SEG3D_LAUNCH_CHECK("seg3d_conv3d_k3_thin_in_persistent_fwd");
  return SEG3D_OK;
}
