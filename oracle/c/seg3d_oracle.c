/* ORACLE (test infrastructure, not product code) -- plain C restatement of the arithmetic on the hot path.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; the product package
 * never does.  The reference (qinliuliuqin/Medical-Segmentation3d-Toolkit) is pure Python over torch.nn operators, so
 * there is nothing of the reference to compile (no oracle/_ref build); this file restates the published semantics of
 * the operators the reference composes, in the tensor layout the reference uses (NCDHW, weights [Cout][Cin][k][k][k],
 * ConvTranspose3d weights [Cin][Cout][k][k][k]), with double accumulation.  Citations are relative to
 * /root/reference/segmentation3d.  Pinned by tests/test_oracle_golden.py against fixtures produced by the real
 * reference modules (oracle/gen_golden.py).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define IDX5(n, c, z, y, x, C, D, H, W) (((((long)(n) * (C) + (c)) * (D) + (z)) * (H) + (y)) * (W) + (x))

/* nn.Conv3d, kernel k, stride s, zero padding p, with bias -- network/module/conv_gn_relu3.py:10,
 * vnet_inblock.py:9, vnet_downblock.py:11, vnet_outblock.py:13,16 */
void oracle_conv3d(const float* x, const float* w, const float* b, float* y, int N, int Cin, int D, int H, int W, int Cout,
                   int k, int s, int p) {
  const int Do = (D + 2 * p - k) / s + 1, Ho = (H + 2 * p - k) / s + 1, Wo = (W + 2 * p - k) / s + 1;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int zo = 0; zo < Do; ++zo)
        for (int yo = 0; yo < Ho; ++yo)
          for (int xo = 0; xo < Wo; ++xo) {
            double acc = b ? (double)b[co] : 0.0;
            for (int ci = 0; ci < Cin; ++ci)
              for (int kz = 0; kz < k; ++kz) {
                const int zi = zo * s + kz - p;
                if (zi < 0 || zi >= D) continue;
                for (int ky = 0; ky < k; ++ky) {
                  const int yi = yo * s + ky - p;
                  if (yi < 0 || yi >= H) continue;
                  for (int kx = 0; kx < k; ++kx) {
                    const int xi = xo * s + kx - p;
                    if (xi < 0 || xi >= W) continue;
                    acc += (double)x[IDX5(n, ci, zi, yi, xi, Cin, D, H, W)] *
                           (double)w[((((long)co * Cin + ci) * k + kz) * k + ky) * k + kx];
                  }
                }
              }
            y[IDX5(n, co, zo, yo, xo, Cout, Do, Ho, Wo)] = (float)acc;
          }
}

/* nn.ConvTranspose3d(kernel_size=2, stride=2) -- network/module/vnet_upblock.py:11 */
void oracle_conv_transpose3d_k2s2(const float* x, const float* w, const float* b, float* y, int N, int Cin, int D, int H,
                                  int W, int Cout) {
  const int Do = 2 * D, Ho = 2 * H, Wo = 2 * W;
  for (int n = 0; n < N; ++n)
    for (int co = 0; co < Cout; ++co)
      for (int zo = 0; zo < Do; ++zo)
        for (int yo = 0; yo < Ho; ++yo)
          for (int xo = 0; xo < Wo; ++xo) {
            double acc = b ? (double)b[co] : 0.0;
            const int kz = zo & 1, ky = yo & 1, kx = xo & 1;
            for (int ci = 0; ci < Cin; ++ci)
              acc += (double)x[IDX5(n, ci, zo >> 1, yo >> 1, xo >> 1, Cin, D, H, W)] *
                     (double)w[((((long)ci * Cout + co) * 2 + kz) * 2 + ky) * 2 + kx];
            y[IDX5(n, co, zo, yo, xo, Cout, Do, Ho, Wo)] = (float)acc;
          }
}

/* nn.GroupNorm(1, C), eps 1e-5, affine, optional residual add and ReLU -- conv_gn_relu3.py:11,14,
 * residual_block3.py:24 */
void oracle_group_norm1(const float* x, const float* gamma, const float* beta, const float* res, float* y, int N, int C,
                        long S, float eps, int relu) {
  const long M = (long)C * S;
  for (int n = 0; n < N; ++n) {
    const float* xn = x + n * M;
    double s = 0.0, ss = 0.0;
    for (long i = 0; i < M; ++i) s += xn[i];
    const double mean = s / (double)M;
    for (long i = 0; i < M; ++i) ss += ((double)xn[i] - mean) * ((double)xn[i] - mean);
    const double rstd = 1.0 / sqrt(ss / (double)M + (double)eps);
    for (int c = 0; c < C; ++c)
      for (long i = 0; i < S; ++i) {
        double v = ((double)xn[c * S + i] - mean) * rstd * (double)gamma[c] + (double)beta[c];
        if (res) v += (double)res[n * M + c * S + i];
        if (relu && v < 0.0) v = 0.0;
        y[n * M + c * S + i] = (float)v;
      }
  }
}

/* nn.Softmax(dim=1) -- vnet_outblock.py:18,23 */
void oracle_softmax_channels(const float* x, float* y, int N, int C, long S) {
  for (int n = 0; n < N; ++n)
    for (long i = 0; i < S; ++i) {
      double mx = -1e300, sum = 0.0;
      for (int c = 0; c < C; ++c) mx = fmax(mx, (double)x[((long)n * C + c) * S + i]);
      for (int c = 0; c < C; ++c) sum += exp((double)x[((long)n * C + c) * S + i] - mx);
      for (int c = 0; c < C; ++c) y[((long)n * C + c) * S + i] = (float)(exp((double)x[((long)n * C + c) * S + i] - mx) / sum);
    }
}

/* MultiDiceLoss -- loss/multi_dice_loss.py:24-43 with loss/binary_dice_loss.py:9-36 folded in:
 * sum_c w_c mean_n [1 - (2 sum(ph t) + 1e-6) / (sum(ph^2) + sum(t^2) + 1e-6)], ph = p * [p > (float)(1/C)] */
float oracle_multi_dice(const float* probs, const float* target, const float* weights, int N, int C, long S) {
  double wsum = 0.0, total = 0.0;
  for (int c = 0; c < C; ++c) wsum += weights[c];
  const float thr = (float)(1.0 / (double)C);
  for (int c = 0; c < C; ++c) {
    double mean_loss = 0.0;
    for (int n = 0; n < N; ++n) {
      double inter = 0.0, p2 = 0.0, t2 = 0.0;
      for (long i = 0; i < S; ++i) {
        const float p = probs[((long)n * C + c) * S + i];
        const double ph = p > thr ? (double)p : 0.0;
        const double t = target[(long)n * S + i] == (float)c ? 1.0 : 0.0;
        inter += ph * t;
        p2 += ph * ph;
        t2 += t * t;
      }
      mean_loss += 1.0 - (2.0 * inter + 1e-6) / (p2 + t2 + 1e-6);
    }
    total += (weights[c] / wsum) * (mean_loss / N);
  }
  return (float)total;
}

/* FocalLoss -- loss/focal_loss.py:27-61: mean_v [-alpha_t (1 - p_t)^gamma log p_t], p_t = p[target] + 1e-10 */
float oracle_focal(const float* probs, const float* target, const float* alpha, int N, int C, long S, float gamma,
                   int size_average) {
  double asum = 0.0, total = 0.0;
  for (int c = 0; c < C; ++c) asum += alpha ? alpha[c] : 1.0;
  for (int n = 0; n < N; ++n)
    for (long i = 0; i < S; ++i) {
      const int t = (int)target[(long)n * S + i];
      const double a = (alpha ? alpha[t] : 1.0) / asum;
      const double pt = (double)(float)(probs[((long)n * C + t) * S + i] + 1e-10f);
      double l = -a * log(pt);
      if (gamma > 0.f) l *= pow(1.0 - pt, (double)gamma);
      total += l;
    }
  return (float)(size_average ? total / ((double)N * S) : total);
}

/* image_partition_by_fixed_size -- utils/image_tools.py:163-218.  All triples are (x, y, z); bbox is updated in
 * place as the reference does.  Writes up to max_boxes start triples; returns the number of boxes; box receives the
 * box size in voxels. */
int oracle_partition(const int* image_size, const double* spacing, int* bbox_start, int* bbox_end,
                     const double* partition_size, const double* partition_stride, int max_stride, int* starts,
                     int max_boxes, int* box) {
  int bsize[3], stride[3], count[3];
  for (int d = 0; d < 3; ++d) {
    int sz = bbox_end[d] - bbox_start[d];
    if (sz > image_size[d]) sz = image_size[d];
    if (sz % max_stride) sz = max_stride * (sz / max_stride + 1);
    if (sz > image_size[d]) sz = image_size[d];
    bsize[d] = sz;
    bbox_end[d] = bbox_start[d] + sz;
    if (bbox_end[d] > image_size[d]) {
      bbox_end[d] = image_size[d];
      bbox_start[d] = bbox_end[d] - sz;
    }
    box[d] = (int)(partition_size[d] / spacing[d] + 0.5);
    if (box[d] % max_stride) box[d] = max_stride * (box[d] / max_stride + 1);
    if (box[d] > bsize[d]) box[d] = bsize[d];
    stride[d] = (int)(partition_stride[d] / spacing[d] + 0.5);
    if (stride[d] > bsize[d]) stride[d] = bsize[d];
    count[d] = (int)ceil((double)(bsize[d] - box[d]) / (double)stride[d]) + 1;
  }
  int nb = 0;
  for (int ix = 0; ix < count[0]; ++ix)
    for (int iy = 0; iy < count[1]; ++iy)
      for (int iz = 0; iz < count[2]; ++iz) {
        int s[3] = {bbox_start[0] + ix * stride[0], bbox_start[1] + iy * stride[1], bbox_start[2] + iz * stride[2]};
        for (int d = 0; d < 3; ++d)
          if (s[d] + box[d] > bbox_end[d]) s[d] = bbox_end[d] - box[d];
        if (nb < max_boxes) memcpy(starts + 3 * nb, s, sizeof(s));
        ++nb;
      }
  return nb;
}
