// postproc.hip -- on-device pre/post-processing around the patch path:
//   * resampling between the image grid and the model-spacing grid      utils/image_tools.py:329-377 (sitk.Resample with
//     an identity transform, LINEAR / NN, default pixel value), called at core/seg_infer.py:267 and :330-333
//   * 26-connected component labelling of one label of a multi-label mask, component sizes, largest / size-thresholded
//     selection                                                            utils/image_tools.py:380-432
//   * bounding box of selected labels                                      utils/image_tools.py:481-510
// ITK semantics restated (no SimpleITK in this environment -> parity unpinned):
//   ResampleImageFilter: output index -> physical point -> continuous input index c (one affine map, evaluated in
//   double); inside test  -0.5 <= c < size - 0.5  per axis (ImageFunction::IsInsideBuffer), else the default pixel;
//   LinearInterpolateImageFunction clamps the 8-neighbourhood at the borders (a coordinate in [-0.5, 0) or
//   (size - 1, size - 0.5) takes the edge value); NearestNeighbor rounds half up.
//   ConnectedComponentImageFilter(FullyConnected) + RelabelComponent: components ordered by size, ties by first
//   voxel in raster order -- the labels here are the component's smallest linear index, which gives the same order.
// All kernels are HBM-bound byte movers / integer work.
#include "seg3d_common.h"
#include "seg3d_hip.h"

struct Affine12 {
  double m[12];  // c = M[:, :3] * (x, y, z) + M[:, 3], rows = (cx, cy, cz)
};

__global__ __launch_bounds__(256) void resample_affine_kernel(const float* __restrict__ src, float* __restrict__ dst, int Xi,
                                                                int Yi, int Zi, int Xo, int Yo, int Zo, Affine12 A,
                                                                int linear, float pad) {
  const i64 total = (i64)Xo * Yo * Zo;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int x = (int)(idx % Xo);
    const i64 t = idx / Xo;
    const int y = (int)(t % Yo), z = (int)(t / Yo);
    const double cx = A.m[0] * x + A.m[1] * y + A.m[2] * z + A.m[3];
    const double cy = A.m[4] * x + A.m[5] * y + A.m[6] * z + A.m[7];
    const double cz = A.m[8] * x + A.m[9] * y + A.m[10] * z + A.m[11];
    float out = pad;
    if (cx >= -0.5 && cx < Xi - 0.5 && cy >= -0.5 && cy < Yi - 0.5 && cz >= -0.5 && cz < Zi - 0.5) {
      if (linear) {
        const double fx = fmin(fmax(cx, 0.0), (double)(Xi - 1)), fy = fmin(fmax(cy, 0.0), (double)(Yi - 1)),
                     fz = fmin(fmax(cz, 0.0), (double)(Zi - 1));
        const int x0 = (int)floor(fx), y0 = (int)floor(fy), z0 = (int)floor(fz);
        const int x1 = x0 + 1 < Xi ? x0 + 1 : x0, y1 = y0 + 1 < Yi ? y0 + 1 : y0, z1 = z0 + 1 < Zi ? z0 + 1 : z0;
        const double dx = fx - x0, dy = fy - y0, dz = fz - z0;
        const i64 r00 = ((i64)z0 * Yi + y0) * Xi, r01 = ((i64)z0 * Yi + y1) * Xi, r10 = ((i64)z1 * Yi + y0) * Xi,
                  r11 = ((i64)z1 * Yi + y1) * Xi;
        const double v000 = src[r00 + x0], v100 = src[r00 + x1], v010 = src[r01 + x0], v110 = src[r01 + x1];
        const double v001 = src[r10 + x0], v101 = src[r10 + x1], v011 = src[r11 + x0], v111 = src[r11 + x1];
        const double a00 = v000 + (v100 - v000) * dx, a01 = v010 + (v110 - v010) * dx;
        const double a10 = v001 + (v101 - v001) * dx, a11 = v011 + (v111 - v011) * dx;
        const double b0 = a00 + (a01 - a00) * dy, b1 = a10 + (a11 - a10) * dy;
        out = (float)(b0 + (b1 - b0) * dz);
      } else {
        int xn = (int)floor(cx + 0.5), yn = (int)floor(cy + 0.5), zn = (int)floor(cz + 0.5);
        xn = xn < 0 ? 0 : (xn >= Xi ? Xi - 1 : xn);
        yn = yn < 0 ? 0 : (yn >= Yi ? Yi - 1 : yn);
        zn = zn < 0 ? 0 : (zn >= Zi ? Zi - 1 : zn);
        out = src[((i64)zn * Yi + yn) * Xi + xn];
      }
    }
    dst[idx] = out;
  }
}

// dst[z][y][x] (Xo, Yo, Zo) = src sampled at c = M * (x, y, z, 1); affine_host: 12 doubles, row-major 3 x 4
extern "C" int seg3d_resample_affine(const float* src, float* dst, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo,
                                     const double* affine_host, int linear, float pad, void* stream) {
  SEG3D_REQUIRE(src && dst && affine_host, "seg3d_resample_affine: null pointer");
  SEG3D_REQUIRE(Xi > 0 && Yi > 0 && Zi > 0 && Xo > 0 && Yo > 0 && Zo > 0, "seg3d_resample_affine: bad dims");
  Affine12 A;
  for (int k = 0; k < 12; ++k) A.m[k] = affine_host[k];
  const i64 total = (i64)Xo * Yo * Zo;
  hipLaunchKernelGGL(resample_affine_kernel, dim3(seg3d_ew_grid(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, Xi,
                     Yi, Zi, Xo, Yo, Zo, A, linear, pad);
  SEG3D_LAUNCH_CHECK("seg3d_resample_affine");
  return SEG3D_OK;
}

// ---- bounding box -----------------------------------------------------------------------------------------------------
struct LabelSet {
  int n;       // 0: every voxel > 0
  int v[16];
};

__global__ __launch_bounds__(256) void mask_bbox_kernel(const signed char* __restrict__ mask, int X, int Y, int Z, LabelSet ls,
                                                          int* __restrict__ box /* xmin ymin zmin xmax ymax zmax */) {
  int lo[3] = {0x7fffffff, 0x7fffffff, 0x7fffffff}, hi[3] = {-1, -1, -1};
  const i64 total = (i64)X * Y * Z;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int m = mask[idx];
    bool sel = false;
    if (ls.n == 0) {
      sel = m > 0;
    } else {
#pragma unroll
      for (int k = 0; k < 16; ++k) sel = sel || (k < ls.n && m == ls.v[k]);
    }
    if (sel) {
      const int x = (int)(idx % X);
      const i64 t = idx / X;
      const int y = (int)(t % Y), z = (int)(t / Y);
      lo[0] = min(lo[0], x); lo[1] = min(lo[1], y); lo[2] = min(lo[2], z);
      hi[0] = max(hi[0], x); hi[1] = max(hi[1], y); hi[2] = max(hi[2], z);
    }
  }
#pragma unroll
  for (int d = 0; d < 3; ++d) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      lo[d] = min(lo[d], __shfl_down(lo[d], off, 64));
      hi[d] = max(hi[d], __shfl_down(hi[d], off, 64));
    }
  }
  if ((threadIdx.x & 63) == 0) {
#pragma unroll
    for (int d = 0; d < 3; ++d) {
      if (hi[d] >= 0) {
        atomicMin(box + d, lo[d]);
        atomicMax(box + 3 + d, hi[d]);
      }
    }
  }
}

// box_device[6] must be initialised to {INT_MAX x3, -1 x3}; afterwards (xmin, ymin, zmin, xmax, ymax, zmax) inclusive, or
// untouched when no voxel is selected.  nlabels == 0 selects every voxel > 0 (get_bounding_box(mask, None)).
extern "C" int seg3d_mask_bounding_box(const signed char* mask, int X, int Y, int Z, const int* labels_host, int nlabels,
                                       int* box_device, void* stream) {
  SEG3D_REQUIRE(mask && box_device && X > 0 && Y > 0 && Z > 0, "seg3d_mask_bounding_box: bad arguments");
  SEG3D_REQUIRE(nlabels >= 0 && nlabels <= 16 && (nlabels == 0 || labels_host), "seg3d_mask_bounding_box: 0..16 labels");
  LabelSet ls;
  ls.n = nlabels;
  for (int k = 0; k < 16; ++k) ls.v[k] = k < nlabels ? labels_host[k] : 0;
  i64 blocks = ((i64)X * Y * Z + 256 * 16 - 1) / (256 * 16);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mask_bbox_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, mask, X, Y, Z, ls,
                     box_device);
  SEG3D_LAUNCH_CHECK("seg3d_mask_bounding_box");
  return SEG3D_OK;
}

// ---- 26-connected components of (mask == label) ----------------------------------------------------------------------
// Union-find on a parent array (label equivalence): parent[v] = v for foreground, -1 for background; every foreground
// voxel is merged with its 13 raster-predecessor neighbours (atomicMin on roots); a final pass flattens the trees, so a
// voxel's label is the smallest linear index of its component -- independent of the execution order.
__device__ __forceinline__ int ccl_find(const int* parent, int v) {
  int p = parent[v];
  while (p != v) {
    v = p;
    p = parent[v];
  }
  return v;
}

__device__ __forceinline__ void ccl_union(int* parent, int a, int b) {
  for (;;) {
    a = ccl_find(parent, a);
    b = ccl_find(parent, b);
    if (a == b) return;
    if (a < b) {
      const int t = a;
      a = b;
      b = t;
    }
    const int old = atomicMin(parent + a, b);  // a > b: hang the larger root under the smaller
    if (old == a) return;
    a = old;                                   // someone re-rooted a meanwhile: continue from there
  }
}

__global__ __launch_bounds__(256) void ccl_init_kernel(const signed char* __restrict__ mask, int label, int* __restrict__ parent,
                                                         i64 total) {
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256)
    parent[idx] = mask[idx] == label ? (int)idx : -1;
}

__global__ __launch_bounds__(256) void ccl_merge_kernel(int* __restrict__ parent, int X, int Y, int Z) {
  const i64 total = (i64)X * Y * Z;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    if (parent[idx] < 0) continue;
    const int x = (int)(idx % X);
    const i64 t = idx / X;
    const int y = (int)(t % Y), z = (int)(t / Y);
    // the 13 neighbours that precede (x, y, z) in raster order
#pragma unroll
    for (int dz = -1; dz <= 0; ++dz)
#pragma unroll
      for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
          if (dz == 0 && (dy > 0 || (dy == 0 && dx >= 0))) continue;
          const int xx = x + dx, yy = y + dy, zz = z + dz;
          if (xx < 0 || xx >= X || yy < 0 || yy >= Y || zz < 0) continue;
          const i64 u = ((i64)zz * Y + yy) * X + xx;
          if (parent[u] >= 0) ccl_union(parent, (int)idx, (int)u);
        }
  }
}

__global__ __launch_bounds__(256) void ccl_flatten_kernel(int* __restrict__ parent, i64 total) {
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256)
    if (parent[idx] >= 0) parent[idx] = ccl_find(parent, (int)idx);
}

// sizes[root] += run length; a thread walks 32 consecutive voxels and flushes one atomic per run of equal roots
__global__ __launch_bounds__(256) void ccl_count_kernel(const int* __restrict__ parent, int* __restrict__ sizes, i64 total) {
  const i64 base = ((i64)blockIdx.x * 256 + threadIdx.x) * 32;
  int cur = -1, run = 0;
  for (int k = 0; k < 32; ++k) {
    const i64 idx = base + k;
    const int r = idx < total ? parent[idx] : -1;
    if (r != cur) {
      if (cur >= 0) atomicAdd(sizes + cur, run);
      cur = r;
      run = 0;
    }
    ++run;
  }
  if (cur >= 0) atomicAdd(sizes + cur, run);
}

// best = max over roots of (size << 32 | ~root): the largest component, ties to the smallest root (first in raster order)
__global__ __launch_bounds__(256) void ccl_best_kernel(const int* __restrict__ parent, const int* __restrict__ sizes,
                                                         unsigned long long* __restrict__ best, i64 total) {
  unsigned long long key = 0ull;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    if (parent[idx] == (int)idx) {
      const unsigned long long k = ((unsigned long long)(unsigned)sizes[idx] << 32) | (0xffffffffull - (unsigned)idx);
      key = k > key ? k : key;
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    const unsigned long long o = __shfl_down(key, off, 64);
    key = o > key ? o : key;
  }
  if ((threadIdx.x & 63) == 0 && key) atomicMax(best, key);
}

// out[v] = value where the voxel belongs to a kept component (mode 0: the largest; mode 1: size >= threshold);
// combine: 0 = overwrite (0 elsewhere), 1 = add `value` to the existing entry (multi-label composition)
__global__ __launch_bounds__(256) void ccl_select_kernel(const int* __restrict__ parent, const int* __restrict__ sizes,
                                                           const unsigned long long* __restrict__ best, int mode, int threshold,
                                                           signed char value, int combine, signed char* __restrict__ out,
                                                           i64 total) {
  const int best_root = (int)(0xffffffffull - (*best & 0xffffffffull));
  const bool any = *best != 0ull;
  for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
    const int r = parent[idx];
    bool keep = false;
    if (r >= 0) keep = mode == 0 ? (any && r == best_root) : sizes[r] >= threshold;
    const signed char v = keep ? value : (signed char)0;
    out[idx] = combine ? (signed char)(out[idx] + v) : v;
  }
}

extern "C" long long seg3d_ccl_workspace_ints(long long voxels) { return 2 * voxels + 2; }

// Components of (mask == label) with 26-connectivity; keeps the largest (mode 0) or those with >= threshold voxels
// (mode 1) and writes `value` there: out = (combine ? out : 0) + value * kept.  workspace: seg3d_ccl_workspace_ints ints.
extern "C" int seg3d_ccl26_select(const signed char* mask, int label, int X, int Y, int Z, int mode, int threshold, int value,
                                  int combine, signed char* out, int* workspace, void* stream) {
  SEG3D_REQUIRE(mask && out && workspace && X > 0 && Y > 0 && Z > 0, "seg3d_ccl26_select: bad arguments");
  SEG3D_REQUIRE((i64)X * Y * Z < (1ll << 31), "seg3d_ccl26_select: volume exceeds 2^31 voxels");
  SEG3D_REQUIRE(mode == 0 || mode == 1, "seg3d_ccl26_select: mode must be 0 (largest) or 1 (size threshold)");
  const i64 total = (i64)X * Y * Z;
  hipStream_t s = (hipStream_t)stream;
  int* parent = workspace;
  int* sizes = workspace + total;
  unsigned long long* best = reinterpret_cast<unsigned long long*>(workspace + 2 * total);  // 8-byte aligned: 2*total even
  const dim3 grid(seg3d_ew_grid(total, 256));
  if (hipMemsetAsync(sizes, 0, (size_t)(total + 2) * sizeof(int), s) != hipSuccess) {
    seg3d_set_error("seg3d_ccl26_select: hipMemsetAsync failed");
    return SEG3D_ERR_LAUNCH;
  }
  hipLaunchKernelGGL(ccl_init_kernel, grid, dim3(256), 0, s, mask, label, parent, total);
  hipLaunchKernelGGL(ccl_merge_kernel, grid, dim3(256), 0, s, parent, X, Y, Z);
  hipLaunchKernelGGL(ccl_flatten_kernel, grid, dim3(256), 0, s, parent, total);
  hipLaunchKernelGGL(ccl_count_kernel, dim3((unsigned)((total + 256 * 32 - 1) / (256 * 32))), dim3(256), 0, s, parent, sizes,
                     total);
  hipLaunchKernelGGL(ccl_best_kernel, grid, dim3(256), 0, s, parent, sizes, best, total);
  hipLaunchKernelGGL(ccl_select_kernel, grid, dim3(256), 0, s, parent, sizes, best, mode, threshold, (signed char)value,
                     combine, out, total);
  SEG3D_LAUNCH_CHECK("seg3d_ccl26_select");
  return SEG3D_OK;
}
