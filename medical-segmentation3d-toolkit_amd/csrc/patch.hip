// patch.hip -- on-device sliding-window batcher: crop + normalise patches out of a resident volume, accumulate
// the per-patch probability maps back with overlap counts, then average and arg-max.
//
// Reference path restated (file:line relative to /root/reference/segmentation3d):
//   * ROI slice + crop_normalizers[0] on the ROI                     core/seg_infer.py:221-224
//       AdaptiveNormalizer: (x - mean(ROI)) / max(std(ROI), 1e-6), clip to +-clip_sigma   utils/normalizer.py:55-62
//       FixedNormalizer:    (x - mean) / stddev, optional clip to [-1, 1]                 utils/normalizer.py:22-25,
//                                                                                          utils/image_tools.py:221-238
//   * acc[c][z0:z1, y0:y1, x0:x1] += prob_c ; count[...] += 1.0       utils/image_tools.py:435-469, seg_infer.py:313-322
//   * probs *= 1 / count ; mask = argmax_c -> int8                    core/seg_infer.py:325-327, 336-339
// Voxel (x, y, z) of a volume with size (X, Y, Z) lives at [z][y][x] (utils/image_tools.py:448,465).
// All kernels are HBM-bound byte movers over fp32 / int8 data.
#include "seg3d_common.h"
#include "seg3d_hip.h"

#define PATCH_STAT_CHUNK 8192

// partial[p][blk][2] = (sum, sum of squares) in fp64 over a chunk of patch p's voxels
__global__ __launch_bounds__(256) void patch_stats_partial_kernel(const float* __restrict__ vol,
                                                                    const int* __restrict__ starts,
                                                                    double* __restrict__ partial, int Y, int X, int bx,
                                                                    int by, int bz, int nblk) {
  __shared__ double red[8];
  const int p = blockIdx.y;
  const int sx = starts[3 * p], sy = starts[3 * p + 1], sz = starts[3 * p + 2];
  const i64 nv = (i64)bx * by * bz;
  const i64 e0 = (i64)blockIdx.x * PATCH_STAT_CHUNK;
  i64 e1 = e0 + PATCH_STAT_CHUNK;
  if (e1 > nv) e1 = nv;
  double s = 0.0, ss = 0.0;
  for (i64 e = e0 + threadIdx.x; e < e1; e += 256) {
    const int lx = (int)(e % bx);
    const i64 t = e / bx;
    const int ly = (int)(t % by), lz = (int)(t / by);
    const double v = (double)vol[((i64)(sz + lz) * Y + (sy + ly)) * X + (sx + lx)];
    s += v;
    ss += v * v;
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = s;
    red[4 + (threadIdx.x >> 6)] = ss;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    partial[((i64)p * nblk + blockIdx.x) * 2 + 0] = red[0] + red[1] + red[2] + red[3];
    partial[((i64)p * nblk + blockIdx.x) * 2 + 1] = red[4] + red[5] + red[6] + red[7];
  }
}

// mean_std[p] = (mean, max(population std, 1e-6)) as float32
__global__ __launch_bounds__(64) void patch_stats_finalize_kernel(const double* __restrict__ partial,
                                                                    float* __restrict__ mean_std, int nblk, double nv) {
  const int p = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int k = threadIdx.x; k < nblk; k += 64) {
    s += partial[((i64)p * nblk + k) * 2];
    ss += partial[((i64)p * nblk + k) * 2 + 1];
  }
  s = wave_sum_d(s);
  ss = wave_sum_d(ss);
  if (threadIdx.x == 0) {
    const double mean = s / nv;
    double var = ss / nv - mean * mean;
    if (var < 0.0) var = 0.0;
    float sd = (float)sqrt(var);
    if (sd < 1e-6f) sd = 1e-6f;
    mean_std[2 * p] = (float)mean;
    mean_std[2 * p + 1] = sd;
  }
}

// batch[p][0][lz][ly][lx] = clip((vol[...] - mean_p) / std_p)
__global__ __launch_bounds__(256) void patch_gather_normalize_kernel(const float* __restrict__ vol,
                                                                       const int* __restrict__ starts,
                                                                       const float* __restrict__ mean_std,
                                                                       float* __restrict__ batch, int Y, int X, int bx,
                                                                       int by, int bz, int P, float fixed_mean,
                                                                       float fixed_std, int clip, float clip_lo,
                                                                       float clip_hi) {
  const i64 nv = (i64)bx * by * bz;
  const i64 total = nv * P;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int p = (int)(idx / nv);
    const i64 e = idx - (i64)p * nv;
    const int lx = (int)(e % bx);
    const i64 t = e / bx;
    const int ly = (int)(t % by), lz = (int)(t / by);
    const int sx = starts[3 * p], sy = starts[3 * p + 1], sz = starts[3 * p + 2];
    const float mean = mean_std ? mean_std[2 * p] : fixed_mean;
    const float sd = mean_std ? mean_std[2 * p + 1] : fixed_std;
    float v = (vol[((i64)(sz + lz) * Y + (sy + ly)) * X + (sx + lx)] - mean) / sd;
    if (clip) {
      if (v < clip_lo) v = clip_lo;
      if (v > clip_hi) v = clip_hi;
    }
    batch[idx] = v;
  }
}

extern "C" long long seg3d_patch_stats_blocks(int bx, int by, int bz) {
  return ((long long)bx * by * bz + PATCH_STAT_CHUNK - 1) / PATCH_STAT_CHUNK;
}

// normalizer_type: 0 = fixed (mean, stddev, clip to [-1,1] when clip != 0), 1 = adaptive (clip to +-clip_sigma),
// -1 = none.  starts: device int32 [P][3] as (x, y, z).  workspace: P * seg3d_patch_stats_blocks * 2 doubles and
// mean_std: P * 2 floats (adaptive only).
extern "C" int seg3d_patch_gather_normalize(const float* volume, const int* starts, float* batch, double* workspace,
                                            float* mean_std, int Z, int Y, int X, int bx, int by, int bz, int P,
                                            int normalizer_type, float mean, float stddev, int clip, float clip_sigma,
                                            void* stream) {
  SEG3D_REQUIRE(volume && starts && batch && P > 0, "seg3d_patch_gather_normalize: bad arguments");
  SEG3D_REQUIRE(bx > 0 && by > 0 && bz > 0 && bx <= X && by <= Y && bz <= Z,
                "seg3d_patch_gather_normalize: box (%d,%d,%d) does not fit volume (%d,%d,%d)", bx, by, bz, X, Y, Z);
  hipStream_t s = (hipStream_t)stream;
  const float* ms = nullptr;
  float fm = 0.f, fs = 1.f, lo = -1.f, hi = 1.f;
  int do_clip = 0;
  if (normalizer_type == 1) {
    SEG3D_REQUIRE(workspace && mean_std, "seg3d_patch_gather_normalize: adaptive normaliser needs workspace");
    SEG3D_REQUIRE(clip_sigma > 0.f, "seg3d_patch_gather_normalize: clip_sigma must be positive");
    const int nblk = (int)seg3d_patch_stats_blocks(bx, by, bz);
    hipLaunchKernelGGL(patch_stats_partial_kernel, dim3(nblk, P), dim3(256), 0, s, volume, starts, workspace, Y, X, bx, by,
                       bz, nblk);
    SEG3D_LAUNCH_CHECK("seg3d_patch_gather_normalize(stats)");
    hipLaunchKernelGGL(patch_stats_finalize_kernel, dim3(P), dim3(64), 0, s, workspace, mean_std, nblk,
                       (double)bx * by * bz);
    SEG3D_LAUNCH_CHECK("seg3d_patch_gather_normalize(finalize)");
    ms = mean_std;
    do_clip = 1;
    lo = -clip_sigma;
    hi = clip_sigma;
  } else if (normalizer_type == 0) {
    SEG3D_REQUIRE(stddev > 0.f, "seg3d_patch_gather_normalize: stddev must be positive");
    fm = mean;
    fs = stddev;
    do_clip = clip;
  } else if (normalizer_type != -1) {
    SEG3D_UNSUPPORTED("seg3d_patch_gather_normalize: unsupported normalization type %d", normalizer_type);
  }
  const i64 total = (i64)bx * by * bz * P;
  hipLaunchKernelGGL(patch_gather_normalize_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, s, volume, starts, ms,
                     batch, Y, X, bx, by, bz, P, fm, fs, do_clip, lo, hi);
  SEG3D_LAUNCH_CHECK("seg3d_patch_gather_normalize");
  return SEG3D_OK;
}

// One thread per volume voxel of the batch's bounding box; patches are applied in list order so the float summation
// order per voxel equals the reference's sequential loop (no atomics, reproducible).  The bounding box and the number
// of valid patches come from a small DEVICE control block so that a captured hipGraph can be replayed for every
// batch with unchanged kernel arguments:  ctl = {lo_x, lo_y, lo_z, extent_x, extent_y, extent_z, n_valid}.
__global__ __launch_bounds__(256) void patch_scatter_accumulate_kernel(const float* __restrict__ probs,
                                                                         const int* __restrict__ starts,
                                                                         const int* __restrict__ ctl,
                                                                         float* __restrict__ acc, float* __restrict__ count,
                                                                         int Z, int Y, int X, int bx, int by, int bz,
                                                                         int C) {
  const int lox = ctl[0], loy = ctl[1], loz = ctl[2], ex = ctl[3], ey = ctl[4], ez = ctl[5], P = ctl[6];
  const i64 total = (i64)ex * ey * ez;
  const i64 vol = (i64)Z * Y * X;
  const i64 nv = (i64)bx * by * bz;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int x = lox + (int)(idx % ex);
    const i64 t = idx / ex;
    const int y = loy + (int)(t % ey), z = loz + (int)(t / ey);
    if (x >= X || y >= Y || z >= Z) continue;
    const i64 v = ((i64)z * Y + y) * X + x;
    for (int p = 0; p < P; ++p) {
      const int lx = x - starts[3 * p], ly = y - starts[3 * p + 1], lz = z - starts[3 * p + 2];
      if (lx >= 0 && lx < bx && ly >= 0 && ly < by && lz >= 0 && lz < bz) {
        const i64 e = ((i64)lz * by + ly) * bx + lx;
        for (int c = 0; c < C; ++c) acc[(i64)c * vol + v] += probs[((i64)p * C + c) * nv + e];
        count[v] += 1.0f;
      }
    }
  }
}

// probs [P][C][bz][by][bx] -> acc [C][Z][Y][X] +=, count [Z][Y][X] += 1.  starts_xyz: device int32 [P][3];
// ctl: device int32 [7] (see kernel); max_box_voxels: host upper bound of the bounding-box volume (sizes the grid).
extern "C" int seg3d_patch_scatter_accumulate(const float* probs, const int* starts, const int* ctl, float* acc,
                                              float* count, int Z, int Y, int X, int bx, int by, int bz, int C,
                                              long long max_box_voxels, void* stream) {
  SEG3D_REQUIRE(probs && starts && ctl && acc && count && C > 0, "seg3d_patch_scatter_accumulate: bad arguments");
  SEG3D_REQUIRE(bx > 0 && by > 0 && bz > 0 && bx <= X && by <= Y && bz <= Z && max_box_voxels > 0,
                "seg3d_patch_scatter_accumulate: bad box");
  hipLaunchKernelGGL(patch_scatter_accumulate_kernel, dim3(seg3d_ew_grid(max_box_voxels, 256)), dim3(256), 0,
                     (hipStream_t)stream, probs, starts, ctl, acc, count, Z, Y, X, bx, by, bz, C);
  SEG3D_LAUNCH_CHECK("seg3d_patch_scatter_accumulate");
  return SEG3D_OK;
}

// acc[c][v] *= 1/count[v] (in place);  mask[v] = argmax_c (first maximum), int8.
// A voxel no patch covered (count 0: bounding-box runs of the coarse -> fine cascade) gets probabilities 0 and class 0: the
// reference divides SimpleITK images (core/seg_infer.py:325-327), and ITK's Div functor yields NumericTraits::max() -- not
// inf -- for a zero denominator, so its product with the zero accumulator is 0, never NaN.
__global__ __launch_bounds__(256) void finalize_argmax_kernel(float* __restrict__ acc, const float* __restrict__ count,
                                                                signed char* __restrict__ mask, int C, i64 vol,
                                                                i64 cstride) {
  for (i64 v = (i64)blockIdx.x * 256 + threadIdx.x; v < vol; v += (i64)gridDim.x * 256) {
    const float cnt = count[v];
    const float r = cnt > 0.f ? 1.0f / cnt : 0.f;
    int best = 0;
    float bv = 0.f;
    for (int c = 0; c < C; ++c) {
      const float p = acc[(i64)c * cstride + v] * r;
      acc[(i64)c * cstride + v] = p;
      if (c == 0 || p > bv) {
        best = c;
        bv = p;
      }
    }
    if (mask) mask[v] = (signed char)best;
  }
}

// class_stride: elements between the class planes of acc (the whole volume; `voxels` may be a z-slab of it -- a rank of
// the sharded sliding window finalizes only the slab it owns); 0 = voxels
extern "C" int seg3d_finalize_argmax(float* acc, const float* count, signed char* mask, int C, long long voxels,
                                     long long class_stride, void* stream) {
  SEG3D_REQUIRE(acc && count && C > 0 && voxels > 0 && (class_stride == 0 || class_stride >= voxels),
                "seg3d_finalize_argmax: bad arguments");
  hipLaunchKernelGGL(finalize_argmax_kernel, dim3(seg3d_ew_grid(voxels, 256)), dim3(256), 0, (hipStream_t)stream, acc, count,
                     mask, C, (i64)voxels, (i64)(class_stride ? class_stride : voxels));
  SEG3D_LAUNCH_CHECK("seg3d_finalize_argmax");
  return SEG3D_OK;
}
