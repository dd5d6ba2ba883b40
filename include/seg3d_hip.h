/* seg3d_hip.h -- C ABI of libseg3d_hip.so, the MI355X (gfx950) engine behind the segmentation3d plugin API.
 *
 * The reference (qinliuliuqin/Medical-Segmentation3d-Toolkit) has no native boundary: its hot path bottoms out in
 * torch.nn modules.  Each entry point below names the reference call site (file:line under
 * /root/reference/segmentation3d) whose arithmetic it replaces.  The Python host binds these with ctypes
 * (medical-segmentation3d-toolkit_amd/segmentation3d/_engine.py); INTEGRATION.md shows the stub a reference maintainer
 * would add.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer (hipMalloc'ed, e.g. torch.Tensor.data_ptr()) unless noted "host";
 *   - activations are fp32 NDHWC: element (n, z, y, x, c) at (((n*D + z)*H + y)*W + x)*C + c;
 *     probabilities / targets at the plugin API edge are fp32 NCDHW planar ([N][C][S], S = D*H*W);
 *   - weights are passed in the reference layouts (Conv3d [Cout][Cin][k][k][k], ConvTranspose3d [Cin][Cout][k][k][k])
 *     and re-packed on device by seg3d_pack_weights_*;
 *   - `stream` is a hipStream_t (0 = default stream); all work is enqueued on it, nothing synchronises;
 *   - the caller owns every buffer including workspaces (sizes via the *_count / *_floats / *_blocks helpers);
 *     the library keeps no mutable global state, so every call is safe under hipGraph stream capture;
 *   - return value: 0 = ok, <0 = error (SEG3D_ERR_*), message via seg3d_last_error() (thread-local).
 */
#ifndef SEG3D_HIP_H
#define SEG3D_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define SEG3D_ABI_VERSION 1

/* ---- library ------------------------------------------------------------------------------------------------ */
const char* seg3d_last_error(void);
int seg3d_abi_version(void);
const char* seg3d_target_arch(void);
int seg3d_device_count(void);

/* ---- layout bridges (utils/image_tools.py:274-326 convert_image_to_tensor / convert_tensor_to_image define the
 *      NCDHW <-> (x,y,z) contract; torch.cat((up, skip), 1) at network/module/vnet_upblock.py:21) ------------------ */
int seg3d_ncdhw_to_ndhwc(const float* in, float* out, int N, int C, long long S, void* stream);
int seg3d_ndhwc_to_ncdhw(const float* in, float* out, int N, int C, long long S, void* stream);
int seg3d_copy_channels(const float* src, float* dst, long long nvox, int C, int src_ld, int src_off, int dst_ld,
                        int dst_off, void* stream);

/* ---- weight packers: W(a, b, t) = w[a*sa + b*sb + t], a = reduction channel, b = output channel, t = tap ---------- */
int seg3d_pack_weights_tapmajor(const float* w, float* wp, int A, int B, int BP, int T, long long sa, long long sb,
                                int flip, void* stream);
int seg3d_pack_weights_mfma(const float* w, float* wp, int A, int B, int T, long long sa, long long sb, int flip,
                            void* stream);
long long seg3d_packed_mfma_floats(int A, int B, int T);
/* the same pack for many weight tensors in one launch: `jobs_device` is a DEVICE array of njobs descriptors sorted by
 * first_block, first_block[k+1] = first_block[k] + seg3d_pack_job_blocks(A, B, T) of job k; total_blocks = their sum */
typedef struct Seg3dPackJob {
  const float* w;        /* weight tensor (reference layout) */
  float* wp;             /* packed destination, seg3d_packed_mfma_floats(A, B, T) floats */
  long long sa, sb;      /* element strides of the reduction-side / output-side channel in w */
  long long first_block; /* first workgroup of this job */
  int A, B, T, flip;
} Seg3dPackJob;
long long seg3d_pack_job_blocks(int A, int B, int T);
int seg3d_pack_weights_mfma_multi(const Seg3dPackJob* jobs_device, int njobs, long long total_blocks, void* stream);

/* ---- convolutions ---------------------------------------------------------------------------------------------
 * nn.Conv3d k3 s1 p1  : network/module/conv_gn_relu3.py:10, vnet_inblock.py:9, vnet_outblock.py:13
 * nn.Conv3d k2 s2     : network/module/vnet_downblock.py:11
 * nn.Conv3d k1        : network/module/vnet_outblock.py:16
 * nn.ConvTranspose3d k2 s2 : network/module/vnet_upblock.py:11
 * Backward (autograd of the above, core/seg_train.py:124): dgrad = the adjoint op with re-packed weights
 * (k3: same kernel, flipped taps; k2s2 <-> convT), wgrad = seg3d_*_wgrad. */
int seg3d_conv3d_fwd_direct(const float* x, const float* wp_tapmajor, const float* bias, float* y, int N, int Di, int Hi,
                            int Wi, int Cin, int Cout, int ksize, int stride, void* stream);
int seg3d_convT3d_k2s2_fwd_direct(const float* x, const float* wp_tapmajor, const float* bias, float* y, int N, int Di,
                                  int Hi, int Wi, int Cin, int Cout, void* stream);
long long seg3d_wgrad_direct_workspace_floats(int N, int Dq, int Hq, int Wq, int CA, int CB, int ntaps);
int seg3d_wgrad_direct(const float* P, const float* Q, float* part, int N, int Dp, int Hp, int Wp, int CA, int CB,
                       int ksize, int stride, int* n_chunks_out /* host */, void* stream);
int seg3d_wgrad_reduce(const float* part, float* dw, int chunks, int T, int A, int B, long long sa, long long sb,
                       int accumulate /* dw += instead of = */, void* stream);

/* fp32 MFMA implicit-GEMM path for k3 s1 p1 with Cin % 4 == 0 (the FLOP-dominant C->C layers) */
long long seg3d_conv3d_k3_mfma_stats_count(int N, int D, int H, int W, int Cin, int Cout);
long long seg3d_conv3d_k3_mfma_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
/* kernel instantiation a shape runs: MA (1..4) = conv3d_k3_mfma_kernel<MA>; 100 + 10*MA + NB = conv3d_k3_mfma2_kernel<MA, NB> */
int seg3d_conv3d_k3_mfma_variant(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_mfma_fwd(const float* x, const float* wp_mfma, const float* bias,
                             const float* addend /* optional, shape of y: y = conv + bias + addend */, float* y,
                             float* stats_partial, float* workspace /* split-K partials; NULL when the query returns 0 */,
                             int N, int D, int H, int W, int Cin, int Cout, void* stream);
long long seg3d_conv3d_k3_mfma_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_mfma_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D, int H, int W,
                               int Cin, int Cout, int accumulate, void* stream);

/* bf16 mode (BASELINE config 5: bf16 activations and weights, fp32 accumulation, fp32 GroupNorm statistics, fp32 master
 * weights).  Conv INPUTS (x) and packed weights are bf16 (raw 16-bit patterns, `void*` at this boundary); bias, addend,
 * conv OUTPUT, statistics and workspaces are fp32 as above.  Same nn.Conv3d(k3, p1) call sites (conv_gn_relu3.py:10).
 * Needs Cin % 16 == 0 and Cout % 4 == 0. */
long long seg3d_packed_mfma_bf16_elems(int A, int B, int T);
int seg3d_pack_weights_mfma_bf16(const float* w, void* wp_bf16, int A, int B, int T, long long sa, long long sb, int flip,
                                 void* stream);
/* bf16 images of many weights in one launch; Seg3dPackJob.wp then points at bf16 storage and first_block advances by
 * seg3d_pack_job_blocks_bf16 */
long long seg3d_pack_job_blocks_bf16(int A, int B, int T);
int seg3d_pack_weights_mfma_bf16_multi(const Seg3dPackJob* jobs_device, int njobs, long long total_blocks, void* stream);
int seg3d_f32_to_bf16(const float* src, void* dst_bf16, long long n, void* stream);
int seg3d_bf16_to_f32(const void* src_bf16, float* dst, long long n, void* stream);
/* Winograd F(2, 3) along x form of the same C -> C 3x3x3 convolution (csrc/conv_wino.hip): 2/3 of the fp32 MFMAs, exact
 * fp32 transforms.  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 36): T = 36 selects the transformed image
 * (t = (kz * 3 + ky) * 4 + p) of the 27-tap weight.  Supported: whole 8 x 8 x 8 tiles, Cin % 8 == 0, Cout % 32 == 0; preferred
 * (the faster choice): also >= 192 (tile, column block) items; other shapes use seg3d_conv3d_k3_mfma_fwd.  replaces nn.Conv3d(C, C, 3, padding=1),
 * network/module/conv_gn_relu3.py:10, and its input gradient (flip = 1, transposed strides) */
int seg3d_conv3d_k3_wino_supported(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino_preferred(int N, int D, int H, int W, int Cin, int Cout);
long long seg3d_conv3d_k3_wino_stats_count(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino_fwd(const float* x, const float* wp_wino, const float* bias, const float* addend, float* y,
                             float* stats_partial, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* Winograd F(2x2, 3x3) over (y, x) form of the same convolution (csrc/conv_wino2d.hip): a 2 x 2 output quad of one z plane
 * from a 4 x 4 input patch with 16 multiplies instead of 36 per kz = 4/9 of the fp32 MFMAs of the direct kernel; coefficients
 * 1, 1/2, 1/4, exact fp32 transforms.  wp = seg3d_pack_weights_mfma(A = Cin, B = Cout, T = 48): T = 48 selects U = G g G^T per
 * kz (t = kz * 16 + py * 4 + px); the image is opaque to the caller (same size as any T = 48 pack; inside a 32 x 8 chunk it is
 * ordered [4-channel half][group of four K steps][channel][output channel][step] -- the LDS image of the kernels, straight copy).
 * Same arguments as the F(2, 3) form.  Supported: D, H, W multiples of 4 (whole 8^3 tiles run
 * the tile kernel, levels that are only whole 4^3 cells -- the 12^3 level -- the cell kernel), Cin % 8 == 0, Cout % 32 == 0;
 * preferred: at least 192 (tile | group of four cells, 32-channel column block) items.
 * replaces nn.Conv3d(C, C, 3, padding=1), network/module/conv_gn_relu3.py:10, and its input gradient */
int seg3d_conv3d_k3_wino2d_supported(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino2d_preferred(int N, int D, int H, int W, int Cin, int Cout);
long long seg3d_conv3d_k3_wino2d_stats_count(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino2d_fwd(const float* x, const float* wp_wino2d, const float* bias, const float* addend, float* y,
                               float* stats_partial, int N, int D, int H, int W, int Cin, int Cout, void* stream);
/* The same with a caller-owned workspace of seg3d_conv3d_k3_wino2d_fwd_workspace_floats(...) floats (0: none wanted, pass null):
 * launches whose (tile, column block) items would leave more than 6 % of the CU-rounds empty -- 432 items on 256 CUs, the 24^3
 * level of the 4 x 96^3 train step -- deal the K chunks of all items to the workgroups in equal contiguous ranges instead (stream-K);
 * the pieces of an item that a range boundary cuts go through the workspace and a finish pass (one more launch on `stream`).
 * Two calls that run concurrently (two streams) need two workspaces.  Same results up to the order of one fp32 addition per
 * output of a cut item.  replaces nn.Conv3d(C, C, 3, padding=1), network/module/conv_gn_relu3.py:10 */
long long seg3d_conv3d_k3_wino2d_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino2d_fwd_ws(const float* x, const float* wp_wino2d, const float* bias, const float* addend, float* y,
                                  float* stats_partial, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                                  void* stream);
/* Winograd F(3, 2) along x form of the weight gradient of the same layers (csrc/conv_wino.hip): 36 point accumulators per
 * (kz, ky, ci block, co block) instead of 27 taps at one voxel PAIR per K slot = 2/3 of the fp32 MFMAs; partial slabs are
 * reduced in fixed order and turned into the three kx taps by the reduce kernel.  Supported: D, H, W multiples of 4, Cin and
 * Cout multiples of 4; dw in the reference layout [Cout][Cin][3][3][3], written or (accumulate != 0) added to.
 * replaces the weight gradient of nn.Conv3d(C, C, 3, padding=1), network/module/conv_gn_relu3.py:10 (autograd) */
int seg3d_conv3d_k3_wino_wgrad_supported(int N, int D, int H, int W, int Cin, int Cout);
long long seg3d_conv3d_k3_wino_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D, int H, int W,
                               int Cin, int Cout, int accumulate, void* stream);
/* Winograd F(3x3, 2x2) over (y, x) form of the same weight gradient (csrc/conv_wino2d.hip): 48 accumulators [3 kz][16 points]
 * instead of 27 taps at one output QUAD per K slot = 4/9 of the fp32 MFMAs of the 27-tap kernel; same arguments, shapes and
 * layouts as seg3d_conv3d_k3_wino_wgrad */
int seg3d_conv3d_k3_wino2d_wgrad_supported(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino2d_wgrad_preferred(int N, int D, int H, int W, int Cin, int Cout);   /* else the F(3, 2) form */
long long seg3d_conv3d_k3_wino2d_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_wino2d_wgrad(const float* x, const float* dy, float* dw, float* workspace, int N, int D, int H, int W,
                                 int Cin, int Cout, int accumulate, void* stream);
long long seg3d_conv3d_k3_bf16_stats_count(int N, int D, int H, int W, int Cin, int Cout);
long long seg3d_conv3d_k3_bf16_fwd_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
/* 200 + 10*MA + NB = conv3d_k3_mfma2_bf16_kernel<MA, NB>; 0 = shape not supported */
int seg3d_conv3d_k3_bf16_variant(int N, int D, int H, int W, int Cin, int Cout);
/* out_bf16 = 1: y is bf16 storage (used for data-gradients, whose consumer is the GroupNorm backward of a bf16 unit);
 * addend stays fp32 and is added before the rounding */
int seg3d_conv3d_k3_bf16_fwd(const void* x_bf16, const void* wp_bf16, const float* bias, const float* addend, void* y,
                             float* stats_partial, float* workspace, int N, int D, int H, int W, int Cin, int Cout,
                             int out_bf16, void* stream);

/* bf16 mode, remaining conv entry points: the INPUT activations (and, for weight gradients, the output gradient) are
 * bf16 and are widened to fp32 while a tile is staged; weights (fp32 pack), accumulation and outputs are fp32.  Each one
 * mirrors the fp32 entry point of the same name without `bf16` (same call sites in the reference). */
long long seg3d_conv3d_k3_bf16_wgrad_workspace_floats(int N, int D, int H, int W, int Cin, int Cout);
int seg3d_conv3d_k3_bf16_wgrad(const void* x_bf16, const void* dy_bf16, float* dw, float* workspace, int N, int D, int H,
                               int W, int Cin, int Cout, int accumulate, void* stream);
/* w_bf16 = 1: wp is a seg3d_pack_weights_mfma_bf16(T = 8) image and the kernel runs the bf16 MFMA (Cin % 16 == 0);
 * w_bf16 = 0: fp32 image, the bf16 input is widened while staging */
int seg3d_conv3d_k2s2_bf16_fwd(const void* x_bf16, const void* wp_mfma, const float* bias, void* y, float* stats_partial,
                               int N, int Do, int Ho, int Wo, int Cin, int Cout, int out_bf16, int w_bf16, void* stream);
int seg3d_convT3d_k2s2_bf16_fwd(const void* x_bf16, const void* wp_mfma, const float* bias, void* y,
                                float* stats_partial, int N, int Di, int Hi, int Wi, int Cin, int Cout, int out_bf16,
                                int w_bf16, void* stream);
/* stride-2 conv data-gradient with the skip connection's gradient folded into the epilogue (y = scatter(x) + addend;
 * DownBlock.down_conv = nn.Conv3d(in, out, 2, stride=2), network/module/vnet_downblock.py:11, whose input also feeds
 * torch.cat in vnet_upblock.py:21).  x_mode 0: fp32; 1: bf16 x, fp32 weight image; 2: bf16 x and image */
int seg3d_convT3d_k2s2_scatter_addend(const void* x, int x_mode, const void* wp, const void* addend, int ld_addend, void* y,
                                      int N, int Di, int Hi, int Wi, int Cin, int Cout, int out_bf16, void* stream);
int seg3d_k2_bf16_wgrad(const void* P_bf16, const void* Q_bf16, float* dw, float* workspace, int N, int Dq, int Hq, int Wq,
                        int CA, int CB, long long sa, long long sb, int accumulate, void* stream);
int seg3d_conv3d_k3_thin_out_bf16_fwd(const void* x_bf16, const float* wq, const float* bias, float* y,
                                      float* stats_partial, int N, int D, int H, int W, int Cin, int Cout, int CO,
                                      void* stream);
/* thin-input conv on the bf16 matrix cores (bf16 mode: stem forward CT = 1, head data-gradient CT = 2; Cout % 4 == 0):
 * x fp32 and the weights enter as bf16 hi + lo pairs (three MFMAs per K-step, exact to 2^-16), y is bf16, statistics
 * slots as seg3d_conv3d_k3_thin_in_fwd.  replaces InputBlock.conv = nn.Conv3d(in, 16, 3, padding=1),
 * network/module/vnet_inblock.py:9, and the input gradient of OutputBlock.conv1, vnet_outblock.py:13 */
int seg3d_conv3d_k3_thin_in_mfma16_supported(int CT, int Cout);
long long seg3d_packed_thin_in16_elems(int CT, int B);
int seg3d_pack_weights_thin_in16(const float* w, void* wq_bf16, int CT, int B, long long sa, long long sb, int flip,
                                 void* stream);
int seg3d_conv3d_k3_thin_in_mfma16_fwd(const float* x, const void* wq_bf16, const float* bias, void* y_bf16,
                                       float* stats_partial, int N, int D, int H, int W, int CT, int Cout, void* stream);
/* the same head conv on the matrix cores (Cin in {16, 32}, Cout <= 3): x taps in the reduction dimension, (kz, ky)
 * taps in the output dimension, weights as a bf16 hi + lo pair (fp32-grade); statistics slots as the thin_out kernel.
 * replaces OutputBlock.conv1 = nn.Conv3d(in, out, 3, padding=1), network/module/vnet_outblock.py:13 */
int seg3d_conv3d_k3_thin_out_mfma_supported(int Cin, int Cout);
long long seg3d_thin_out_mfma_packed_elems(int Cin);
int seg3d_pack_weights_thin_out_mfma(const float* w, void* wp_bf16, int A, int B, long long sa, long long sb, int flip,
                                     void* stream);
int seg3d_conv3d_k3_thin_out_mfma_fwd(const void* x_bf16, const void* wp_bf16, const float* bias, float* y,
                                      float* stats_partial, int N, int D, int H, int W, int Cin, int Cout, void* stream);

/* fp32 MFMA path for the stride-2 2x2x2 layers (Cin % 4 == 0): gather = Conv3d k2s2 forward / ConvTranspose3d dgrad,
 * scatter = ConvTranspose3d k2s2 forward / Conv3d k2s2 dgrad, pair-reduce = weight gradient of both.
 * stats_partial: [N][*_stats_count(...)][2] partial (sum, sum of squares) of y; the slots are an opaque order (since round 4:
 * [tile][4 sub-tiles or waves][column block]) -- callers only ever sum a sample's slots (seg3d_gn_stats_finalize) */
long long seg3d_conv3d_k2s2_mfma_stats_count(int Do, int Ho, int Wo, int Cout);
int seg3d_conv3d_k2s2_mfma_fwd(const float* x, const float* wp_mfma, const float* bias, float* y, float* stats_partial,
                               int N, int Do, int Ho, int Wo, int Cin, int Cout, void* stream);
/* x is a channel slice of a wider NDHWC buffer: ld_x floats between consecutive voxel rows (>= Cin, multiple of 4).  Inference:
 * the encoder feature that feeds both the next DownBlock (vnet_downblock.py:11) and a decoder concatenation
 * (vnet_upblock.py:21) is normalised straight into its half of the concatenated buffer and read from there. */
int seg3d_conv3d_k2s2_mfma_fwd_ld(const float* x, int ld_x, const float* wp, const float* bias, float* y, float* stats, int N,
                                  int Do, int Ho, int Wo, int Cin, int Cout, void* stream);
long long seg3d_convT3d_k2s2_mfma_stats_count(int Di, int Hi, int Wi, int Cout);
int seg3d_convT3d_k2s2_mfma_fwd(const float* x, const float* wp_mfma, const float* bias, float* y, float* stats_partial,
                                int N, int Di, int Hi, int Wi, int Cin, int Cout, void* stream);
long long seg3d_k2_mfma_wgrad_workspace_floats(int N, int Dq, int Hq, int Wq, int CA, int CB);
int seg3d_k2_mfma_wgrad(const float* P, const float* Q, float* dw, float* workspace, int N, int Dq, int Hq, int Wq, int CA,
                        int CB, long long sa, long long sb, int accumulate, void* stream);

/* thin 3x3x3 layers at full resolution (stem Cin <= 8 -> 16, head 32 -> num_classes <= 8): HBM-bound special cases
 * of Conv3d k3 p1 (vnet_inblock.py:9, vnet_outblock.py:13) and of their autograd adjoints */
long long seg3d_packed_thin_in_floats(int CT, int B);
int seg3d_pack_weights_thin_in(const float* w, float* wp, int CT, int B, long long sa, long long sb, int flip, void* stream);
long long seg3d_conv3d_k3_thin_stats_count(int D, int H, int W, int Cout_blocks);
int seg3d_conv3d_k3_thin_in_fwd(const float* x, const float* wp_thin, const float* bias, float* y, float* stats_partial,
                                int N, int D, int H, int W, int CT, int Cout, void* stream);
/* bf16 mode: y is bf16 storage (the stem's conv output, the head's data-gradient); everything else as above */
int seg3d_conv3d_k3_thin_in_bf16out_fwd(const float* x, const float* wp_thin, const float* bias, void* y_bf16,
                                        float* stats_partial, int N, int D, int H, int W, int CT, int Cout, void* stream);
int seg3d_pack_weights_thin_out(const float* w, float* wq, int A, int B, int CO, long long sa, long long sb, int flip,
                                void* stream);
long long seg3d_conv3d_k3_thin_out_stats_count(int D, int H, int W);
int seg3d_conv3d_k3_thin_out_fwd(const float* x, const float* wq, const float* bias, float* y, float* stats_partial, int N,
                                 int D, int H, int W, int Cin, int Cout, int CO, void* stream);
/* persistent form of the thin-input conv (csrc/conv_thin_f32.hip): same arithmetic and packed weights as
 * seg3d_conv3d_k3_thin_in_fwd, a workgroup walks tiles with every tile-invariant index hoisted out of the tile loop;
 * statistics: one slot per wave.  y is fp32, or bf16 storage when out_bf16.  replaces InputBlock.conv =
 * nn.Conv3d(in, 16, 3, padding=1), network/module/vnet_inblock.py:9, and the input gradient of OutputBlock.conv1 */
long long seg3d_conv3d_k3_thin_in_persistent_stats_count(int D, int H, int W, int Cout_blocks);
int seg3d_conv3d_k3_thin_in_persistent_fwd(const float* x, const float* wp_thin, const float* bias, void* y,
                                           float* stats_partial, int N, int D, int H, int W, int CT, int Cout, int out_bf16,
                                           void* stream);
/* fp32 head conv on the fp32 matrix cores (csrc/conv_thin_f32.hip; Cin in {16, 32}, Cout <= 5): x taps in the reduction
 * dimension of v_mfma_f32_16x16x4_f32, (kz, ky) taps in its output rows, the ninth tap on the VALU when 8 taps fill the
 * rows exactly (2 and 4 classes); every voxel row is read once, no LDS staging of activations.  Exact fp32 arithmetic.
 * replaces OutputBlock.conv1 = nn.Conv3d(in, out, 3, padding=1), network/module/vnet_outblock.py:13 */
int seg3d_conv3d_k3_thin_out_f32mfma_supported(int Cin, int Cout);
long long seg3d_thin_out_f32mfma_packed_floats(int Cin, int Cout);
int seg3d_pack_weights_thin_out_f32mfma(const float* w, float* wpk, int A, int B, long long sa, long long sb, int flip,
                                        void* stream);
long long seg3d_conv3d_k3_thin_out_f32mfma_stats_count(int N, int D, int H, int W);
int seg3d_conv3d_k3_thin_out_f32mfma_fwd(const float* x, const float* wpk, const float* bias, float* y,
                                         float* stats_partial, int N, int D, int H, int W, int Cin, int Cout, void* stream);
long long seg3d_k3_thin_wgrad_workspace_floats(int N, int D, int H, int W, int CT, int CF);
int seg3d_k3_thin_wgrad(const float* thin, const float* fat, float* dw, float* workspace, int N, int D, int H, int W, int CT,
                        int CF, long long s_ct, long long s_cf, int flip, int accumulate, void* stream);
/* bf16 mode: the same with a bf16 `fat` operand (head weight gradient: fat = the bf16 input activation; stem weight
 * gradient: fat = the bf16 gradient of the conv output).  Runs on the bf16 matrix cores with `thin` split into a bf16
 * hi + lo pair (exact to 2^-17) */
int seg3d_k3_thin_wgrad_fatbf16(const float* thin, const void* fat_bf16, float* dw, float* workspace, int N, int D, int H,
                                int W, int CT, int CF, long long s_ct, long long s_cf, int flip, int accumulate,
                                void* stream);

/* ---- GroupNorm(1, C) [+ ReLU] [+ residual]  (network/module/conv_gn_relu3.py:11,14; residual_block3.py:24,46) ------ */
long long seg3d_gn_stats_count(long long M);
int seg3d_gn_stats_partial(const float* y, float* part, int N, long long M, void* stream);
int seg3d_gn_stats_finalize(const float* part, float* mean_rstd, int N, int count, long long M, float eps, void* stream);
int seg3d_gn_apply(const float* y, const float* mean_rstd, const float* gamma, const float* beta, const float* res,
                   float* out, int N, long long S, int C, int relu,
                   int ld_out /* floats between consecutive voxels of `out`; 0 = C (a channel slice of a wider buffer) */,
                   void* stream);
long long seg3d_gn_bwd_blocks(long long S);
int seg3d_gn_bwd_reduce(const float* dout, const float* out /* NULL: recompute the ReLU mask from y */, const float* y,
                        const float* mean_rstd, const float* gamma, const float* beta, float* part, int N, long long S,
                        int C, int relu, int ld_dout /* row stride of dout in floats, 0 = C */, void* stream);
int seg3d_gn_bwd_finalize(const float* part, const float* gamma, const float* mean_rstd, float* abx, float* s12,
                          float* dgamma, float* dbeta, float* dbias, int N, long long S, int C,
                          int acc_mask /* bit 0/1/2: accumulate into dgamma/dbeta/dbias */, void* stream);
/* both stages in one launch (last-ticket workgroup runs the parameter stage); ticket: one device int, zero before the
 * first call and left zero by every call */
int seg3d_gn_bwd_finalize_fused(const float* part, const float* gamma, const float* mean_rstd, float* abx, float* s12,
                                float* dgamma, float* dbeta, float* dbias, int* ticket, int N, long long S, int C,
                                int acc_mask, void* stream);
int seg3d_gn_bwd_apply(const float* dout, const float* out /* NULL: recompute */, const float* y, const float* mean_rstd,
                       const float* s12, const float* gamma, const float* beta, float* dy, float* dres, int N, long long S,
                       int C, int relu, int ld_dout /* row stride of dout in floats, 0 = C */, void* stream);

/* bf16 mode GroupNorm: the conv output y, the statistics and all arithmetic stay fp32; the activation-side tensors
 * (residual, unit output, incoming gradient) are bf16 where flagged.  ld_out / ld_dout count elements of that tensor. */
/* y_bf16: the conv output y itself is bf16 storage (the conv epilogue rounded it after taking the fp32 statistics) */
int seg3d_gn_apply_mixed(const void* y, const float* mean_rstd, const float* gamma, const float* beta, const void* res,
                         void* out, int N, long long S, int C, int relu, int ld_out, int res_bf16, int out_bf16,
                         int y_bf16, void* stream);
int seg3d_gn_bwd_reduce_bf16(const void* dout_bf16, const void* out_bf16, const void* y, const float* mean_rstd,
                             const float* gamma, const float* beta, float* part, int N, long long S, int C, int relu,
                             int ld_dout, int y_bf16, void* stream);
int seg3d_gn_bwd_apply_bf16(const void* dout_bf16, const void* out_bf16, const void* y, const float* mean_rstd,
                            const float* s12, const float* gamma, const float* beta, void* dy, float* dres, int N,
                            long long S, int C, int relu, int ld_dout, int dy_bf16, int y_bf16, void* stream);

/* ---- head softmax (network/module/vnet_outblock.py:18,23) ---------------------------------------------------------- */
int seg3d_softmax_fwd(const float* in_ndhwc, float* probs_ncdhw, int N, int C, long long S, void* stream);
int seg3d_softmax_bwd(const float* probs_ncdhw, const float* dprobs_ncdhw, float* din_ndhwc, int N, int C, long long S,
                      void* stream);

/* ---- losses: MultiDiceLoss (loss/multi_dice_loss.py:24-43 + loss/binary_dice_loss.py:9-36),
 *              FocalLoss (loss/focal_loss.py:27-61) ------------------------------------------------------------------ */
long long seg3d_dice_blocks(long long S);
int seg3d_dice_fwd(const float* probs, const float* target, const float* weights, float* part, float* sums, float* loss,
                   int N, int C, long long S, void* stream);
int seg3d_dice_bwd(const float* probs, const float* target, const float* sums, const float* weights, const float* gout,
                   float* dprobs, int N, int C, long long S, void* stream);
/* BinaryDiceLoss called on its own (loss/binary_dice_loss.py:9-36): probs [N][2][S], pred = p1 * [p1 > p0] (ties -> 0),
 * float target; part: [N][seg3d_dice_blocks(S)][3], sums: [N][2] (kept for backward), one: device float 1.0f */
int seg3d_binary_dice_fwd(const float* probs, const float* target, const float* one, float* part, float* sums, float* loss,
                          int N, long long S, void* stream);
int seg3d_binary_dice_bwd(const float* probs, const float* target, const float* sums, const float* gout, float* dprobs,
                          int N, long long S, void* stream);
long long seg3d_focal_blocks(long long total_vox);
int seg3d_focal_fwd(const float* probs, const float* target, const float* alpha, float* part, float* loss, int N, int C,
                    long long S, long long sn, long long sc, long long ss, float gamma, int size_average, void* stream);
int seg3d_focal_bwd(const float* probs, const float* target, const float* alpha, const float* gout, float* dprobs, int N,
                    int C, long long S, long long sn, long long sc, long long ss, float gamma, int size_average,
                    void* stream);

/* ---- optimizer: optim.Adam(...).step()  (core/seg_train.py:83,127) -------------------------------------------------- */
int seg3d_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n, int step, float lr,
                    float beta1, float beta2, float eps, float weight_decay, float grad_scale, void* stream);
/* the same update with the step count kept on the device (advanced by the call), for train steps captured in a hipGraph:
 * step_dev = steps taken so far, bc_dev = 2 floats of device scratch */
int seg3d_adam_step_devstep(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, long long n,
                            int* step_dev, float* bc_dev, float lr, float beta1, float beta2, float eps,
                            float weight_decay, float grad_scale, void* stream);

/* ---- sliding-window batcher (core/seg_infer.py:208-246, 313-327, 336-339; utils/image_tools.py:435-469;
 *      utils/normalizer.py:6-81) ---------------------------------------------------------------------------------- */
long long seg3d_patch_stats_blocks(int bx, int by, int bz);
int seg3d_patch_gather_normalize(const float* volume, const int* starts_xyz, float* batch, double* workspace,
                                 float* mean_std, int Z, int Y, int X, int bx, int by, int bz, int P, int normalizer_type,
                                 float mean, float stddev, int clip, float clip_sigma, void* stream);
int seg3d_patch_scatter_accumulate(const float* probs, const int* starts_xyz, const int* ctl /* device int32[7] */,
                                   float* acc, float* count, int Z, int Y, int X, int bx, int by, int bz, int C,
                                   long long max_box_voxels, void* stream);
int seg3d_finalize_argmax(float* acc, const float* count, signed char* mask, int C, long long voxels,
                          long long class_stride, void* stream);

/* ---- evaluation metric: utils/metrics.py:5-37 cal_dsc, core/seg_eval.py:8-57 ---------------
 * counts[3k..3k+2] += (area_gt, area_seg, intersection) of labels_host[k] over two label volumes of n elements;
 * the caller zeroes counts first.  dtype: 0 int8, 1 uint8, 2 int16, 3 int32, 4 float32.  1..16 labels per call. */
int seg3d_label_overlap_counts(const void* gt, const void* seg, int dtype, long long n, const int* labels_host, int nlabels,
                               unsigned long long* counts, void* stream);

/* ---- pre/post-processing around the patch path: utils/image_tools.py:329-432, 481-510 ----------
 * resample: dst[z][y][x] (Xo, Yo, Zo) = src sampled at the continuous index c = M * (x, y, z, 1), M = 12 doubles on the
 * HOST (row-major 3 x 4); ITK semantics: inside iff -0.5 <= c < size - 0.5, else `pad`; linear (clamped 8-neighbourhood)
 * or nearest neighbour (round half up). */
int seg3d_resample_affine(const float* src, float* dst, int Xi, int Yi, int Zi, int Xo, int Yo, int Zo,
                          const double* affine_host, int linear, float pad, void* stream);
/* box_device[6] initialised to {INT_MAX x3, -1 x3} -> inclusive (xmin, ymin, zmin, xmax, ymax, zmax) of the voxels whose
 * value is in labels_host (nlabels == 0: every voxel > 0); untouched when nothing is selected */
int seg3d_mask_bounding_box(const signed char* mask, int X, int Y, int Z, const int* labels_host, int nlabels,
                            int* box_device, void* stream);
/* 26-connected components of (mask == label): keep the largest (mode 0; ties: first in raster order) or every component
 * with >= threshold voxels (mode 1); out = (combine ? out : 0) + value * kept.  workspace: seg3d_ccl_workspace_ints ints */
long long seg3d_ccl_workspace_ints(long long voxels);
int seg3d_ccl26_select(const signed char* mask, int label, int X, int Y, int Z, int mode, int threshold, int value,
                       int combine, signed char* out, int* workspace, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SEG3D_HIP_H */
