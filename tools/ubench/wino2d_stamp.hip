// diagnostic harness: compiles conv_wino2d.hip with in-kernel s_memtime stamps (W2_STAMPS) and prints where a wave of
// conv3d_k3_wino2d_kernel (the 8^3-tile kernel; waves 0-3 and 4-7 share the SIMDs) spends its cycles per K chunk, also by the
// chunk's position inside its item.  Without -DW2_STAMPS it is a plain launch timer of seg3d_conv3d_k3_wino2d_fwd for any shape the
// entry takes (tiles or cells): argv = N D H W C [mode: 0 bias, 1 neither, 2 fused addend].  Not part of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DW2_STAMPS -I include -I medical-segmentation3d-toolkit_amd/csrc \
//         tools/ubench/wino2d_stamp.hip medical-segmentation3d-toolkit_amd/csrc/seg3d_api.cpp -o tools/ubench/wino2d_stamp
#ifndef W2_SRC   // -DW2_SRC='"/path/to/an/experimental/copy.hip"' builds the harness around a variant of the kernel file
#define W2_SRC "../../medical-segmentation3d-toolkit_amd/csrc/conv_wino2d.hip"
#endif
#include W2_SRC
#include <stdio.h>
#include <vector>
int main(int argc, char** argv) {
  int N = 8, D = 96, H = 96, W = 96, C = 32;
  if (argc >= 6) { N = atoi(argv[1]); D = atoi(argv[2]); H = atoi(argv[3]); W = atoi(argv[4]); C = atoi(argv[5]); }
  const int mode = argc >= 7 ? atoi(argv[6]) : 0;   // 0: bias (forward); 1: neither bias nor addend (plain data-gradient); 2: addend, no bias (fused data-gradient)
  const size_t nx = (size_t)N * D * H * W * C;
  const size_t nw = (size_t)(C / 32) * (C / 8) * 48 * 256;
  std::vector<float> hx(nx), hw(nw);
  unsigned s = 12345u;
  for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-3f; }
  for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = ((int)(s >> 8) % 2001 - 1000) * 1e-4f; }
  float *x, *wp, *y, *bias, *addend = nullptr;
  long long* stamps;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&wp, nw * 4); (void)hipMalloc(&y, nx * 4);
  (void)hipMalloc(&bias, C * 4); (void)hipMemset(bias, 0, C * 4);
  if (mode == 2) { (void)hipMalloc(&addend, nx * 4); (void)hipMemset(addend, 0, nx * 4); }
  if (mode != 0) bias = nullptr;
  (void)hipMemcpy(x, hx.data(), nx * 4, hipMemcpyHostToDevice); (void)hipMemcpy(wp, hw.data(), nw * 4, hipMemcpyHostToDevice);
  const size_t nst = 8 * 8 * 64 * 8;   // 8 workgroups x up to 8 waves x 64 chunks x 8 stamps
  (void)hipMalloc(&stamps, nst * 8); (void)hipMemset(stamps, 0, nst * 8);
#ifdef W2_STAMPS
  (void)hipMemcpyToSymbol(HIP_SYMBOL(w2_stamp_buf), &stamps, sizeof(stamps));
#endif
  for (int r = 0; r < 3; ++r) {
    int rc = seg3d_conv3d_k3_wino2d_fwd(x, wp, bias, addend, y, nullptr, N, D, H, W, C, C, nullptr);
    if (rc) { printf("error %d\n", rc); return 1; }
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) seg3d_conv3d_k3_wino2d_fwd(x, wp, bias, addend, y, nullptr, N, D, H, W, C, C, nullptr);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
#ifndef W2_STAMPS   // plain timing build: -DW2_STAMPS left out
  printf("N=%d %d^3 C=%d mode %d: %.3f ms, %.1f TFLOP/s algorithmic\n", N, D, C, mode, ms, 2.0 * N * D * H * W * 27 * C * C / ms / 1e9);
  return 0;
#endif
  printf("N=%d %d^3 C=%d mode %d: %.3f ms, %.1f TFLOP/s algorithmic (stamped build)\n", N, D, C, mode, ms, 2.0 * N * D * H * W * 27 * C * C / ms / 1e9);
  std::vector<long long> h(nst);
  (void)hipMemcpy(h.data(), stamps, nst * 8, hipMemcpyDeviceToHost);
  const char* names[4] = {"MFMA loop (+ DMA issue, transform stages)", "cursor advance + DMA wait", "barrier", "to next loop start (per-item work)"};
  for (int wv = 0; wv < 8; ++wv) {
    double acc[4] = {0, 0, 0, 0}; int n = 0; double period = 0;
    for (int b = 0; b < 8; ++b)
      for (int c = 1; c + 1 < 60; ++c) {
        const long long* t = &h[((b * 8 + wv) * 64 + c) * 8];
        const long long* tn = &h[((b * 8 + wv) * 64 + c + 1) * 8];
        if (!t[0] || !t[3] || !tn[0]) continue;
        acc[0] += t[1] - t[0]; acc[1] += t[2] - t[1]; acc[2] += t[3] - t[2]; acc[3] += tn[0] - t[3];
        period += tn[0] - t[0]; ++n;
      }
    if (!n) continue;
    printf("wave %d (%d chunks): period %.0f cycles", wv, n, n ? period / n : 0.0);
    for (int k = 0; k < 4; ++k) printf(" | %s %.0f", names[k], n ? acc[k] / n : 0.0);
    printf("\n");
  }
  {  // by position of the chunk inside its item (wave 4 = a slower wave of its SIMD): where do the waits sit?
    const int nsc = C / 4;
    printf("wave 4 by chunk position in the item:\n");
    for (int pos = 0; pos < nsc; ++pos) {
      double acc[4] = {0, 0, 0, 0}; int n = 0;
      for (int b = 0; b < 8; ++b)
        for (int c = 1; c + 1 < 60; ++c) {
          if (c % nsc != pos) continue;
          const long long* t = &h[((b * 8 + 4) * 64 + c) * 8];
          const long long* tn = &h[((b * 8 + 4) * 64 + c + 1) * 8];
          if (!t[0] || !t[3] || !tn[0]) continue;
          acc[0] += t[1] - t[0]; acc[1] += t[2] - t[1]; acc[2] += t[3] - t[2]; acc[3] += tn[0] - t[3]; ++n;
        }
      if (n) printf("  chunk %d: loop %.0f | advance + DMA wait %.0f | barrier %.0f | to next loop start %.0f\n", pos, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n);
    }
  }
  {  // per item: epilogue (after the last chunk's barrier -> stores issued) and accumulator init + decode of the next item
    double epi = 0, ini = 0; int n = 0;
    for (int b = 0; b < 8; ++b)
      for (int wv = 0; wv < 8; ++wv)
        for (int c = 1; c + 1 < 60; ++c) {
          const long long* t = &h[((b * 8 + wv) * 64 + c) * 8];
          const long long* tn = &h[((b * 8 + wv) * 64 + c + 1) * 8];
          if (!t[6] || !t[3] || !tn[0]) continue;
          epi += t[6] - t[3]; ini += tn[0] - t[6]; ++n;
        }
    printf("per item (%d samples): epilogue %.0f cycles, accumulator init + decode %.0f\n", n, n ? epi / n : 0.0, n ? ini / n : 0.0);
  }
  return 0;
}
