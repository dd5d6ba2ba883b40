// diagnostic harness: compiles the conv kernels with in-kernel s_memtime stamps (SEG3D_STAMPS) and prints the average
// cycles a workgroup of conv3d_k3_mfma2_kernel spends in each phase.  Not part of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DSEG3D_STAMPS -I include -I medical-segmentation3d-toolkit_amd/csrc \
//         tools/ubench/conv_stamp.hip medical-segmentation3d-toolkit_amd/csrc/seg3d_api.cpp -o tools/ubench/conv_stamp
#include "../../medical-segmentation3d-toolkit_amd/csrc/conv_mfma.hip"
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
  int N = 4, D = 96, H = 96, W = 96, Cin = 32, Cout = 32;
  if (argc >= 7) { N = atoi(argv[1]); D = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); Cin = atoi(argv[5]); Cout = atoi(argv[6]); }
  const size_t nx = (size_t)N * D * H * W * Cin, ny = (size_t)N * D * H * W * Cout;
  const size_t nw = (size_t)((Cout + 31) / 32) * ((Cin + 7) / 8) * SEG3D_W_CHUNK;
  std::vector<float> hx(nx), hw(nw);
  unsigned s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-4f; }
  float *x, *wp, *y, *st, *bias;
  long long* stamps;
  const long long cnt = seg3d_conv3d_k3_mfma_stats_count(N, D, H, W, Cin, Cout);
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&wp, nw * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&st, (size_t)N * cnt * 2 * 4);
  (void)hipMalloc(&bias, Cout * 4); (void)hipMemset(bias, 0, Cout * 4);
  (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(wp, hw.data(), nw * 4, hipMemcpyHostToDevice);
  const size_t nst = (size_t)1 << 20;
  (void)hipMalloc(&stamps, nst * 16 * 8); (void)hipMemset(stamps, 0, nst * 16 * 8);
  (void)hipMemcpyToSymbol(HIP_SYMBOL(seg3d_stamp_buf), &stamps, sizeof(stamps));
  for (int r = 0; r < 5; ++r) {
    int rc = seg3d_conv3d_k3_mfma_fwd(x, wp, bias, nullptr, y, st, nullptr, N, D, H, W, Cin, Cout, nullptr);
    if (rc) { printf("error %d: %s\n", rc, seg3d_last_error()); return 1; }
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) seg3d_conv3d_k3_mfma_fwd(x, wp, bias, nullptr, y, st, nullptr, N, D, H, W, Cin, Cout, nullptr);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  const int variant = seg3d_conv3d_k3_mfma_variant(N, D, H, W, Cin, Cout);
  printf("shape N=%d %dx%dx%d %d->%d variant %d: %.3f ms, %.1f TFLOP/s (stamped build)\n", N, D, H, W, Cin, Cout, variant, ms,
         2.0 * N * D * H * W * 27 * Cin * Cout / ms / 1e9);
  std::vector<long long> h(nst * 16);
  (void)hipMemcpy(h.data(), stamps, nst * 16 * 8, hipMemcpyDeviceToHost);
  // slots: [workgroup][0] first-item DMA issue, [1] first chunk landed;  [item][2] MFMA chunks done, [3] epilogue done
  double mfma = 0, epi = 0, pro = 0; long long nit = 0, nwg = 0;
  std::vector<long long> done;  // per item: (t2, t3)
  for (size_t b = 0; b < nst; ++b) {
    const long long* t = &h[b * 16];
    if (t[2] && t[3]) { epi += (double)(t[3] - t[2]); ++nit; }
    if (b < 256 && t[0] && t[1]) { pro += (double)(t[1] - t[0]); ++nwg; }
  }
  // per workgroup: items b, b+G, ...: time between consecutive epilogue ends = full item period
  const int G = 256;
  double period = 0; long long np = 0;
  for (size_t b = 0; b + G < nst; ++b) {
    if (h[b * 16 + 3] && h[(b + G) * 16 + 3]) { period += (double)(h[(b + G) * 16 + 3] - h[b * 16 + 3]); ++np; }
    if (h[b * 16 + 3] && h[(b + G) * 16 + 2]) mfma += (double)(h[(b + G) * 16 + 2] - h[b * 16 + 3]);
  }
  printf("  items %lld, workgroups %lld\n  first-item DMA prologue  %8.0f cycles\n  item period (steady)     %8.0f cycles\n"
         "    MFMA chunks            %8.0f\n    epilogue               %8.0f\n",
         nit, nwg, nwg ? pro / nwg : 0.0, np ? period / np : 0.0, np ? mfma / np : 0.0, nit ? epi / nit : 0.0);
  // ---- weight gradient (conv3d_k3_wgrad2_kernel): per-tile stamps of the first 5 tiles of every workgroup ----
  {
    float *dyb, *dw, *ws;
    (void)hipMalloc(&dyb, ny * 4); (void)hipMemcpy(dyb, x, (ny < nx ? ny : nx) * 4, hipMemcpyDeviceToDevice);
    (void)hipMalloc(&dw, (size_t)Cout * Cin * 27 * 4);
    (void)hipMalloc(&ws, (size_t)seg3d_conv3d_k3_mfma_wgrad_workspace_floats(N, D, H, W, Cin, Cout) * 4);
    (void)hipMemset(stamps, 0, nst * 16 * 8);
    for (int r = 0; r < 3; ++r) seg3d_conv3d_k3_mfma_wgrad(x, dyb, dw, ws, N, D, H, W, Cin, Cout, 0, nullptr);
    (void)hipEventRecord(e0);
    for (int r = 0; r < 10; ++r) seg3d_conv3d_k3_mfma_wgrad(x, dyb, dw, ws, N, D, H, W, Cin, Cout, 0, nullptr);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
    printf("wgrad: %.3f ms, %.1f TFLOP/s (stamped build)\n", ms, 2.0 * N * D * H * W * 27 * Cin * Cout / ms / 1e9);
    (void)hipMemcpy(h.data(), stamps, nst * 16 * 8, hipMemcpyDeviceToHost);
    double loop = 0, bar = 0, gap = 0; long long nl = 0, ng = 0;
    for (size_t b = 0; b < 256; ++b) {
      const long long* t = &h[b * 16];
      for (int k = 1; k < 4; ++k) {   // skip the first tile (cold)
        if (t[3 * k] && t[3 * k + 1] && t[3 * k + 2]) { loop += (double)(t[3 * k + 1] - t[3 * k]); bar += (double)(t[3 * k + 2] - t[3 * k + 1]); ++nl; }
        if (t[3 * k + 2] && t[3 * k + 3]) { gap += (double)(t[3 * k + 3] - t[3 * k + 2]); ++ng; }
      }
    }
    printf("  per tile: MFMA loop %8.0f cycles (ideal 28672), vmcnt+barrier %6.0f, next-tile setup %6.0f\n",
           nl ? loop / nl : 0.0, nl ? bar / nl : 0.0, ng ? gap / ng : 0.0);
  }
  return 0;
}
