// metrics.hip -- label-overlap counts for the Dice evaluation metric.
//
// Reference restated (file:line relative to /root/reference/segmentation3d):  utils/metrics.py:5-37 `cal_dsc`
//   gt == label, seg == label, area_gt = sum, area_seg = sum, intersection = sum(gt & seg)
// evaluated by core/seg_eval.py:8-57 once per (case, label), i.e. three full passes over both volumes per label on
// the host.  Here ONE pass over the two label volumes yields (area_gt, area_seg, intersection) for all requested
// labels: integer counting, bit-exact, HBM-bound (algorithmic bytes = 2 * voxels * element size).
#include "seg3d_common.h"
#include "seg3d_hip.h"

#define METRIC_MAX_LABELS 16

struct MetricLabels {
  int v[METRIC_MAX_LABELS];
};

template <typename T>
__global__ __launch_bounds__(256) void label_overlap_kernel(const T* __restrict__ gt, const T* __restrict__ seg, i64 n,
                                                             MetricLabels labels, int nlabels,
                                                             unsigned long long* __restrict__ counts) {
  unsigned cg[METRIC_MAX_LABELS], cs[METRIC_MAX_LABELS], ci[METRIC_MAX_LABELS];
#pragma unroll
  for (int k = 0; k < METRIC_MAX_LABELS; ++k) cg[k] = cs[k] = ci[k] = 0u;
  // a thread visits at most 2^31 / (gridDim * 256) elements: 32-bit per-thread counters cannot overflow for n < 2^40
  for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n; i += (i64)gridDim.x * 256) {
    const T g = gt[i], s = seg[i];
#pragma unroll
    for (int k = 0; k < METRIC_MAX_LABELS; ++k) {
      if (k < nlabels) {
        const T l = (T)labels.v[k];
        const bool a = g == l, b = s == l;
        cg[k] += a ? 1u : 0u;
        cs[k] += b ? 1u : 0u;
        ci[k] += (a && b) ? 1u : 0u;
      }
    }
  }
  __shared__ unsigned red[4][METRIC_MAX_LABELS * 3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < METRIC_MAX_LABELS; ++k) {
    if (k < nlabels) {
      unsigned a = cg[k], b = cs[k], c = ci[k];
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) {
        a += __shfl_down(a, off, 64);
        b += __shfl_down(b, off, 64);
        c += __shfl_down(c, off, 64);
      }
      if (lane == 0) {
        red[wave][3 * k] = a;
        red[wave][3 * k + 1] = b;
        red[wave][3 * k + 2] = c;
      }
    }
  }
  __syncthreads();
  if ((int)threadIdx.x < 3 * nlabels) {
    const unsigned long long v = (unsigned long long)red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] +
                                 red[3][threadIdx.x];
    if (v) atomicAdd(counts + threadIdx.x, v);   // integer atomics: the result does not depend on the order
  }
}

template <typename T>
static void launch_overlap(const void* gt, const void* seg, i64 n, const MetricLabels& l, int nlabels,
                           unsigned long long* counts, hipStream_t s) {
  i64 blocks = (n + 256 * 16 - 1) / (256 * 16);   // >= 16 elements per thread
  if (blocks > 256 * 16) blocks = 256 * 16;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL((label_overlap_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, s, (const T*)gt, (const T*)seg, n, l,
                     nlabels, counts);
}

// counts[k] = (area_gt, area_seg, intersection) of labels[k]; the caller zeroes `counts` (3 * nlabels u64) first.
// dtype: 0 = int8, 1 = uint8, 2 = int16, 3 = int32, 4 = float32 (label volumes as the formats store them).
extern "C" int seg3d_label_overlap_counts(const void* gt, const void* seg, int dtype, long long n, const int* labels_host,
                                          int nlabels, unsigned long long* counts, void* stream) {
  SEG3D_REQUIRE(gt && seg && labels_host && counts, "seg3d_label_overlap_counts: null pointer");
  SEG3D_REQUIRE(n > 0 && n < (1ll << 40), "seg3d_label_overlap_counts: bad element count");
  SEG3D_REQUIRE(nlabels > 0 && nlabels <= METRIC_MAX_LABELS, "seg3d_label_overlap_counts: 1..%d labels per call (got %d)",
                METRIC_MAX_LABELS, nlabels);
  MetricLabels l;
  for (int k = 0; k < METRIC_MAX_LABELS; ++k) l.v[k] = k < nlabels ? labels_host[k] : 0;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case 0: launch_overlap<signed char>(gt, seg, n, l, nlabels, counts, s); break;
    case 1: launch_overlap<unsigned char>(gt, seg, n, l, nlabels, counts, s); break;
    case 2: launch_overlap<short>(gt, seg, n, l, nlabels, counts, s); break;
    case 3: launch_overlap<int>(gt, seg, n, l, nlabels, counts, s); break;
    case 4: launch_overlap<float>(gt, seg, n, l, nlabels, counts, s); break;
    default: SEG3D_UNSUPPORTED("seg3d_label_overlap_counts: unknown dtype code %d", dtype);
  }
  SEG3D_LAUNCH_CHECK("seg3d_label_overlap_counts");
  return SEG3D_OK;
}
