// diagnostic harness: compiles conv_k2_mfma.hip with s_memtime stamps (K2_STAMPS) in conv3d_k2s2_direct_kernel and prints, over ALL
// waves of one launch, where a wave's lifetime goes (prologue + load issue | K loop | stores | statistics) and how the workgroups'
// start times spread (dispatch rounds).  argv = N Do Cin Cout (cubic output extent Do).  Not part of the product library.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -DK2_STAMPS -I include -I medical-segmentation3d-toolkit_amd/csrc \
//         tools/ubench/k2_stamp.hip medical-segmentation3d-toolkit_amd/csrc/seg3d_api.cpp -o tools/ubench/k2_stamp
#include "../../medical-segmentation3d-toolkit_amd/csrc/conv_k2_mfma.hip"
#include <stdio.h>
#include <vector>
#include <algorithm>
int main(int argc, char** argv) {
  int N = 4, Do = 24, Cin = 32, Cout = 64;
  if (argc >= 5) { N = atoi(argv[1]); Do = atoi(argv[2]); Cin = atoi(argv[3]); Cout = atoi(argv[4]); }
  const size_t nx = (size_t)N * 8 * Do * Do * Do * Cin, ny = (size_t)N * Do * Do * Do * Cout;
  const size_t nw = (size_t)((Cout + 31) / 32) * ((Cin + 7) / 8) * 2048;
  float *x, *wp, *y, *bias, *st;
  (void)hipMalloc(&x, nx * 4); (void)hipMalloc(&wp, nw * 4); (void)hipMalloc(&y, ny * 4); (void)hipMalloc(&bias, Cout * 4);
  (void)hipMemset(x, 0, nx * 4); (void)hipMemset(wp, 0, nw * 4); (void)hipMemset(bias, 0, Cout * 4);
  const long long nst = seg3d_conv3d_k2s2_mfma_stats_count(Do, Do, Do, Cout);
  (void)hipMalloc(&st, (size_t)N * nst * 2 * 4);
  K2Tile t = k2_pick_tile(Do, Do, Do);
  const int tiles = N * seg3d_cdiv(Do, t.tz) * seg3d_cdiv(Do, t.ty) * seg3d_cdiv(Do, t.tx), ncob = (Cout + 31) / 32;
  const size_t nwaves = (size_t)tiles * ncob * 4;
  long long* stamps;
  (void)hipMalloc(&stamps, nwaves * 8 * 8); (void)hipMemset(stamps, 0, nwaves * 8 * 8);
#ifdef K2_STAMPS
  (void)hipMemcpyToSymbol(HIP_SYMBOL(k2_stamp_buf), &stamps, sizeof(stamps));
#endif
  for (int r = 0; r < 3; ++r)
    if (int rc = seg3d_conv3d_k2s2_mfma_fwd(x, wp, bias, y, st, N, Do, Do, Do, Cin, Cout, nullptr)) { printf("error %d: %s\n", rc, seg3d_last_error()); return 1; }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  for (int r = 0; r < 10; ++r) seg3d_conv3d_k2s2_mfma_fwd(x, wp, bias, y, st, N, Do, Do, Do, Cin, Cout, nullptr);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("N=%d out %d^3 %d->%d, tile %dx%dx%d, %d tiles x %d column blocks: %.1f us per launch\n", N, Do, Cin, Cout, t.tz, t.ty, t.tx, tiles, ncob, ms * 1e3);
#ifdef K2_STAMPS
  std::vector<long long> h(nwaves * 8);
  (void)hipMemcpy(h.data(), stamps, nwaves * 8 * 8, hipMemcpyDeviceToHost);
  long long t0 = -1, t1 = 0;
  for (size_t w = 0; w < nwaves; ++w) { if (t0 < 0 || h[w * 8] < t0) t0 = h[w * 8]; t1 = std::max(t1, h[w * 8 + 4]); }
  double seg[4] = {0, 0, 0, 0};
  std::vector<long long> starts, life;
  for (size_t w = 0; w < nwaves; ++w) {
    for (int k = 0; k < 4; ++k) seg[k] += (double)(h[w * 8 + k + 1] - h[w * 8 + k]);
    starts.push_back(h[w * 8] - t0);
    life.push_back(h[w * 8 + 4] - h[w * 8]);
  }
  std::sort(starts.begin(), starts.end()); std::sort(life.begin(), life.end());
  printf("s_memtime ticks first start -> last end: %lld (%.1f ticks per us if that is the launch)\n", t1 - t0, (t1 - t0) / (ms * 1e3));
  printf("per wave (mean ticks): prologue + first loads issued %.0f | K loop %.0f | epilogue stores issued %.0f | statistics %.0f\n", seg[0] / nwaves, seg[1] / nwaves, seg[2] / nwaves, seg[3] / nwaves);
  printf("wave lifetime ticks: min %lld median %lld p90 %lld max %lld\n", life.front(), life[life.size() / 2], life[life.size() * 9 / 10], life.back());
  printf("wave start ticks after the first: ");
  for (int q = 0; q <= 10; ++q) printf("%lld ", starts[std::min(starts.size() - 1, starts.size() * q / 10)]);
  printf("\n");
#endif
  return 0;
}
