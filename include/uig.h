/*
 * uig.h — C ABI of the MI355X-native CycleGAN train-step hot path (libuig.so).
 *
 * Drop-in boundary.  The reference snapshot defines no FFI of its own
 * (/root/reference/README.md:1 is the whole tree), so every entry point below names the
 * ATen operator schema (torch 2.10) whose arithmetic it replaces on the GPU.
 *
 * Conventions
 *   - plain pointers + ints, no torch types; every pointer is DEVICE memory owned by the caller
 *     (torch's caching allocator); the library allocates nothing and keeps no pointer past return.
 *   - all work is enqueued asynchronously on `stream` (a hipStream_t passed as void*).
 *   - return 0 = ok; negative = argument error (nothing launched, message in uig_last_error());
 *     positive = hipError_t of the failed launch.
 *   - activations: NHWC, channel count physically padded to a multiple of 8 (pad channels are 0);
 *     dtype: UIG_F32 (exact-f32 MFMA path, the L-inf < 1e-3 parity path) or UIG_BF16.
 *   - weights at the boundary: the torch layouts, fp32 (Conv2d: O,I,kH,kW; ConvTranspose2d: I,O,kH,kW);
 *     uig_pack_weight() produces the kernel-side [N][tap][C] operand in the compute dtype.
 */
#ifndef UIG_H
#define UIG_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

enum { UIG_F32 = 0, UIG_BF16 = 1 };
enum { UIG_ACT_NONE = 0, UIG_ACT_RELU = 1, UIG_ACT_LRELU = 2, UIG_ACT_TANH = 3 };
enum { UIG_PAD_ZERO = 0, UIG_PAD_REFLECT = 1 };
/* gather modes of uig_conv_gather: direct = strided cross-correlation (conv fwd, convT dgrad);
 * transposed = fractionally-strided gather in stride*stride sub-pixel phases (convT fwd, conv dgrad). */
enum { UIG_GATHER_DIRECT = 0, UIG_GATHER_TRANSPOSED = 1 };
/* weight packing: A: out[d0][tap][d1] (row = dim0 of the torch weight), B: out[d1][tap][d0], taps flipped=0/1 */
enum { UIG_PACK_ROW_DIM0 = 0, UIG_PACK_ROW_DIM1 = 1 };

const char* uig_version(void);
const char* uig_last_error(void);
int uig_device_ok(void); /* 1 if a gfx950 device is visible to the HIP runtime */
/* test hook: the kernel family the process's last uig_conv_gather* launch ran on (any thread: autograd's backward thread
 * included), so that a parity test can assert it covers the kernel it means to cover rather than a fallback */
enum { UIG_K_NONE = 0, UIG_K_IGEMM = 1, UIG_K_STRIP128 = 2, UIG_K_STRIP256 = 3, UIG_K_STRIP_PK = 4, UIG_K_ROWSTRIP = 5,
       UIG_K_HEADROW = 6, UIG_K_GEMV = 7, UIG_K_CIN8 = 8, UIG_K_TR2 = 9 };
int uig_debug_last_conv_kernel(void);
/* tuning / test hook: force the conv tile width for layers with >64 output channels (0 = auto, 128, 256) */
void uig_debug_set_tile(int bn);
/* tuning / test hook: 1 (default) = stride-1 3x3 convs use the LDS-resident-strip kernels, 0 = always the generic gather,
 * 2 = 128x128 single-buffer strip tiles, 3 = never the persistent (one block per CU, walks its tiles) 256x128 kernel.
 * All uig_debug_set_* hooks write process-global selection state: set them before launching, never concurrently with launches. */
void uig_debug_set_strip(int on);
void uig_debug_set_strip_wide(int on);   /* 1 (default): reflection-padded 3x3 forward convs on 128-pixel-wide maps on the persistent strip kernel (512-row strip), 0: generic kernel */
/* tuning hook of the persistent strip kernel: dm = K-loop variant (0 default: DMA issue at the top of the K-step, the next step's
 * strip fragments read behind this step's MFMAs; 12 = without that prefetch; 2 / 4 / 8 / 9 / 10 = the rejected variants of DESIGN
 * §3.2), grid = persistent grid size (0 = one block per CU) */
void uig_debug_set_strip_pk(int dm, int grid);      /* dm: 0 default (round 4: the phased schedule where it applies), 5 round 3's schedule, 12/20/21/23 older A/B variants */
long uig_debug_strip_pk_phased_count(void);          /* launches so far that ran the phased schedule (tests) */
/* tuning / test hook: 1 (default) = uig_reflect3x3_dgrad_mirror_applicable may say 1; 0 = it never does (A/B against the border GEMM);
 * 3 / 5 / 7 = diagnostic timing builds of the mirror kernel WITHOUT its per-chunk / first-chunk / any mirror sums (wrong results:
 * scripts/bench_dgrad_mirror.py and scripts/stamp_strip_pk.py only) */
void uig_debug_set_mirror(int on);
/* tuning / test hook: 1 (default) = 7x7 stride-1 convs with <= 16 output channels use the row-strip kernel */
void uig_debug_set_rowstrip(int on);
void uig_debug_set_infer_cs(int on);              /* inference InstanceNorm (uig_instnorm_act_fwd_infer): 1 (default) = blocks own 64 channels and finalise only those, 0 = round 2's whole-C form */
void uig_debug_set_strip_small(int mode);         /* 64x64-tile bf16 strip kernel (four waves of 64 px x 16 ch) for plain launches: 2 = never (default: training keeps a batch-independent kernel choice), 0 = auto (grids of <= 64 128x128 blocks; the inference path switches it on around its launches), 1 = wherever it applies */
void uig_debug_set_strip_small_stages(int n);     /* weight stages of the 64x64-tile kernel: 4 (default: tiles three K-steps ahead, counted waits) / 2 */
void uig_debug_set_strip_stages(int n);           /* 128x128-tile bf16 strip kernel: weight stages 2 (default) / 4 (three K-steps ahead, counted waits: measured slower, A-B hook) */
/* diagnostic build hook: device buffer (u64[blocks*8*4]) that receives in-kernel cycle stamps; NULL = off (default) */
void uig_debug_set_strip_stamps(void* dev_buf);
void uig_debug_set_mx_issuers(int n);            /* MX fp8 kernel A-B hook: waves of a block that issue the K loop's DMAs (4 default, 8 = all: the round-2 form) */
void uig_debug_set_mx_stamps(void* dev_buf);      /* the same for the MX fp8 kernel: 8 x u64 per wave {total, wait, issue, reads+MFMA, epilogue, row table, tiles, 0} */

/* aten::convolution / aten::convolution_backward(input grad) — implicit GEMM, LDS-staged im2col tiles -> MFMA.
 *   y[b, oh, ow, n] = act( bias[n] + sum_{tap,c} x[b, ih(tap), iw(tap), c] * wp[n][tap][c] )
 * direct:      ih = oh*stride + kh - pad            (pad_mode zero or reflect),   y is Ho x Wo
 * transposed:  oh = ih*stride - pad + kh  (all (ih,kh) pairs that hit oh; zero outside), y is Ho x Wo
 * x: (B,H,W,Cin) NHWC, Cin % 8 == 0;  wp: packed [Nrows][kH*kW][Cin];  y: (B,Ho,Wo,ldc), channels n < Nstore written
 * (rows n >= Nrows of wp are treated as zero).  bias may be NULL (fp32[>=Nstore]).                              */
int uig_conv_gather(const void* x, const void* wp, const float* bias, void* y,
                    int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                    int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                    int act, float slope, int dtype, void* stream);

/* Convenience boundary of SURVEY.md §8(b) (contiguous NCHW activations, OIHW fp32 weight, any repack internal and counted in
 * the call): aten::convolution for Conv2d (groups 1, dilation 1), pad_mode zero or reflect.  Three extra passes (NCHW ->
 * NHWC, weight pack, NHWC -> NCHW) run inside; the hot path avoids them by keeping NHWC between layers.
 *   x (B,Cin,H,W) dtype; w (Cout,Cin,kH,kW) fp32; bias fp32[Cout] or NULL; y (B,Cout,Ho,Wo) dtype; workspace: ..._workspace_bytes */
size_t uig_conv2d_fwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int dtype);
int uig_conv2d_fwd(const void* x, const float* w, const float* bias, void* y,
                   int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int pad_mode, int dtype,
                   void* workspace, size_t workspace_bytes, void* stream);
/* The other two operators on the same contiguous-NCHW boundary (round 4).
 * aten::convolution, transposed: ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) - the up-sampling layers of the path.
 *   x (B,Cin,H,W) dtype; w (Cin,Cout,3,3) fp32; bias fp32[Cout] or NULL; y (B,Cout,2H,2W) dtype. */
size_t uig_conv_transpose2d_fwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int dtype);
int uig_conv_transpose2d_fwd(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int H, int W, int Cout, int dtype,
                             void* workspace, size_t workspace_bytes, void* stream);
/* aten::convolution_backward for Conv2d (groups 1, dilation 1, stride 1 or 2); output_mask = which of dx / dW / db are non-NULL.
 *   dy (B,Cout,Ho,Wo), x (B,Cin,H,W) dtype; w (Cout,Cin,kH,kW) fp32; dx (B,Cin,H,W) dtype; dW (Cout,Cin,kH,kW) fp32; db fp32[Cout].
 * pad_mode UIG_PAD_REFLECT: the gradient of ReflectionPad2d(pad) + Conv2d(padding 0), folded back onto the H x W input. */
size_t uig_conv2d_bwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int dtype);
int uig_conv2d_bwd(const void* dy, const void* x, const float* w, void* dx, float* dW, float* db,
                   int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int pad_mode, int dtype,
                   void* workspace, size_t workspace_bytes, void* stream);

/* Paired launch: the same convolution for TWO networks of identical architecture in one grid (CycleGAN's G_A/G_B and
 * D_A/D_B always process same-shaped batches).  Images b < group_images use (wp, bias), the others (wp2, bias2). */
int uig_conv_gather_pair(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                         int group_images, void* y,
                         int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                         int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                         int act, float slope, int dtype, void* stream);

/* Superset entry: optional second network (wp2 != NULL) and optional fused InstanceNorm statistics: in_partial
 * (fp32[B * (Ho*Wo/64) * Nstore * 2]) receives per-64-pixel (sum, sum of squares) of the stored output per channel, consumed by
 * uig_instnorm_act_fwd_pre.  Needs Nrows a multiple of 64 (> 64), Nstore == Nrows and (gather grid) % 64 == 0.
 * border_add (optional, strip kernel only: uig_conv_strip_applicable): the compact border buffer of
 * uig_reflect3x3_dgrad_border, added to output rows 1 / H-2 and columns 1 / W-2 in the epilogue.
 * res_add (optional, strip kernel only): a tensor of y's shape and dtype added to the output in the epilogue (the ResBlock's
 * skip gradient in the input-gradient launch: saves the separate gradient-accumulation pass). */
int uig_conv_gather_ex(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                       int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                       int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                       int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                       int act, float slope, int dtype, void* stream);

/* uig_conv_gather_ex that ALSO emits the statistics of the InstanceNorm BACKWARD which consumes this launch's output as its dy
 * (the output of a ResBlock conv's input-gradient launch is the dy of the norm in front of that conv): bst_x = that norm's
 * saved input (shape and dtype of y), bst_stats = its (mean, rstd) fp32[B][ldc][2], bst_act / bst_slope = its activation;
 * bst_partial fp32[B][Ho*Wo/64][Nstore][2] receives (sum g, sum g*xhat) per 64-pixel slab, g = dy * act'(xhat), computed on
 * the values as stored - exactly what the norm's own statistics pass would read back.  uig_instnorm_act_bwd_colsum_pre then
 * skips that pass (one full read of dy and x).  bf16 strip-kernel launches with border_add and / or res_add, Ho*Wo % 64 == 0. */
int uig_conv_gather_bst(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                        int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                        int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                        int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                        int act, float slope, int dtype,
                        const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial, void* stream);

/* Round 4 - statistics that come out of the convolution launch FINAL (no finalize launch between a convolution and the InstanceNorm
 * apply pass behind it).  uig_conv_gather_ex plus: in_stats fp32[B][Nstore][2] receives (mean, rstd) with eps = in_eps, bit-identical
 * to uig_instnorm_finalize(in_partial): the blocks of the launch take one ARRIVAL TICKET per tile and image after their statistics
 * slabs are out (write-through stores, drained, one relaxed agent-scope fetch_add per block) and the block that draws an image's
 * last ticket reduces that image's slabs in the finalize kernel's fixed order.  tickets: >= B zero-initialised 32-bit words owned
 * by the caller, left zero by the launch (one arena per device and stream of launches is enough: launches are stream-ordered).
 * Kernel families without the in-launch form (and tickets == NULL) run the finalize launch behind the convolution: same result.
 * uig_debug_set_in_tickets(0) forces that everywhere (A/B and parity hook).  Consumer: uig_instnorm_apply_fwd. */
void uig_debug_set_in_tickets(int on);
void uig_debug_set_colsum_slabs(int n);         /* tuning hook: pixel slabs per image of uig_instnorm_act_bwd_colsum* (default 32) */
int uig_conv_gather_fin(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                        int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                        int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                        int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                        int act, float slope, int dtype, float* in_stats, float in_eps, unsigned* tickets, void* stream);

/* Input gradient of a reflection-padded (pad 1) 3x3 stride-1 conv WITHOUT the padded (H+2)x(W+2) gradient + fold:
 * this computes the mirrored-border terms (8 groups: top/bottom/left/right lines + 4 corners) into bord[B][8][H][ldc];
 * then uig_conv_gather_ex(dy, ..., transposed, pad=1, zero, border_add=bord) produces dx on the exact HxW grid. */
int uig_reflect3x3_dgrad_border(const void* dy, const void* wp, const void* wp2, int group_images, void* bord,
                                int B, int H, int W, int C, int Nrows, int ldc, int dtype, void* stream);
int uig_conv_strip_applicable(int B, int H, int W, int Cin, int Nrows, int Ho, int Wo, int dh_min, int dh_max, int dtype);

/* The same input gradient in ONE launch where the persistent strip kernel can fold the mirrored-border terms itself ("mirror
 * pixels": sums of two - at the corners four - pixels of dy placed behind the tile's LDS strip and addressed by the taps that
 * would have read a mirrored line or column; bf16, 64-pixel-wide maps of >= 8 lines, C % 64 == 0, Nrows == ldc, Nrows % 128 == 0):
 * dx (B, H, W, ldc) = zero-padded transposed conv of dy (B, H, W, C) + mirrored terms [+ res_add of dx's shape].  No border
 * buffer, no border GEMM in front.  uig_reflect3x3_dgrad_mirror_applicable: 1 where it applies, else use the two launches above
 * (which also carry the optional fused norm-backward statistics of uig_conv_gather_bst; uig_reflect3x3_dgrad_mirror_bst below is this launch with them). */
int uig_reflect3x3_dgrad_mirror_applicable(int B, int H, int W, int C, int Nrows, int ldc, int dtype);
int uig_reflect3x3_dgrad_mirror(const void* dy, const void* wp, const void* wp2, int group_images, const void* res_add, void* dx,
                                int B, int H, int W, int C, int Nrows, int ldc, int dtype, void* stream);
/* Round 4: the same launch (a variant of the mirror-pixel kernel) ALSO emits the statistics of the InstanceNorm backward that consumes
 * dx as its dy - the arguments of uig_conv_gather_bst - and delivers them FINAL: bst_gm fp32[B][ldc][2] = (mean g, mean g*xhat),
 * finalised inside the launch through `tickets` (>= B zero words, left zero; see uig_conv_gather_fin) or, with tickets == NULL, by
 * a finalize launch behind it.  uig_instnorm_act_bwd_colsum_t(pre_gm = bst_gm) is then ONE launch: neither the norm's statistics
 * pass (a full read of dy and x) nor a finalize launch.  dx is bitwise what uig_reflect3x3_dgrad_mirror writes. */
int uig_reflect3x3_dgrad_mirror_bst(const void* dy, const void* wp, const void* wp2, int group_images, const void* res_add, void* dx,
                                    int B, int H, int W, int C, int Nrows, int ldc, int dtype,
                                    const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial, float* bst_gm,
                                    unsigned* tickets, void* stream);

/* ---- 3x3 stride-1 pad-1 convolution (zero or reflection padding) that applies the InstanceNorm(+ReLU / LeakyReLU) IN FRONT of
 * it to its own input as it is staged (round 3; replaces aten::instance_norm's apply pass + aten::convolution of a ResBlock's
 * second convolution by ONE launch):
 *   y = conv(act((x_raw - mean) * rstd), wp) + bias,   (mean, rstd) = nrm_stats fp32[B][Cin][2] (uig_instnorm_finalize)
 * x_raw (B,H,W,Cin): the raw output of the previous convolution; nrm_act / nrm_slope: the norm's activation.
 * h_out (optional, shape of x_raw): receives the normalised activations (bitwise what uig_instnorm_act_fwd_pre would write) - the
 * backward pass's weight-gradient operand.
 * wp / bias / wp2 / bias2 / group_images / in_partial / y / ldc / act / slope: as uig_conv_gather_ex (direct gather, Nstore = Nrows).
 * bf16, persistent strip kernel shapes only: uig_conv3x3_innorm_applicable. */
int uig_conv3x3_innorm_applicable(int B, int H, int W, int Cin, int Nrows, int pad_mode, int ldc, int dtype);
int uig_conv3x3_innorm_fwd(const void* x_raw, const float* nrm_stats, int nrm_act, float nrm_slope, void* h_out,
                           const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images, float* in_partial, void* y,
                           int B, int H, int W, int Cin, int Nrows, int pad_mode, int ldc, int act, float slope, int dtype, void* stream);
void uig_debug_set_normconv(int on);
/* the finalize launch of uig_instnorm_act_fwd_pre on its own: partial (in_partial of uig_conv_gather_ex, nslab per image) ->
 * stats (mean, rstd) fp32[B][C][2] */
int uig_instnorm_finalize(const float* partial, int nslab, float* stats, int B, int64_t HW, int C, float eps, void* stream);

/* ---- MX block-scaled fp8 path (BASELINE.json configs[4]): the 3x3 stride-1 pad-1 convolutions (forward: gather_mode direct,
 * zero or reflection padding; input gradient: gather_mode transposed, zero padding + border_add) on
 * v_mfma_scale_f32_16x16x128_f8f6f4.  Operands: OCP e4m3 bytes with one E8M0 scale byte per 32 consecutive channels
 * (uig_mx_quantize): xq (B,H,W,Cin) + xs (B,H,W,Cin/32); wq [Nrows][9][Cin] + ws [Nrows][9][Cin/32] (quantise the packed
 * bf16 operand of uig_pack_weight as a [Nrows*9][Cin] matrix).  fp32 accumulate, bf16 output y (B,H,W,ldc); bias, paired
 * launch (wq2 / ws2 / bias2 / group_images), in_partial, border_add, res_add as in uig_conv_gather_ex.
 * Needs Cin and Nrows multiples of 128 and 256-pixel strips of at most 448 rows (uig_conv3x3_mx_fp8_applicable). */
int uig_conv3x3_mx_fp8_applicable(int B, int H, int W, int Cin, int Nrows);
int uig_conv3x3_mx_fp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                       const void* wq2, const void* ws2, const float* bias2, int group_images,
                       float* in_partial, const void* border_add, const void* res_add, void* y,
                       int B, int H, int W, int Cin, int Nrows, int pad_mode, int gather_mode, int ldc,
                       int act, float slope,
                       const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial /* all NULL / 0: off; see uig_conv_gather_bst */,
                       void* stream);
/* InstanceNorm forward / backward that ALSO emit the MX fp8 form (mx_q [B*HW][C] e4m3, mx_s [B*HW][C/32] E8M0) of exactly the
 * bf16 tensor they write - the operand of the fp8 convolution that consumes it, without the stand-alone quantiser's extra pass.
 * fwd: partial != NULL as uig_instnorm_act_fwd_pre, else as uig_instnorm_act_fwd (workspace).  bf16, C a multiple of 32. */
int uig_instnorm_act_fwd_mx(const void* x, const void* residual, void* y, float* stats, const float* partial, int nslab,
                            float* workspace, void* mx_q, void* mx_s,
                            int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream);
int uig_instnorm_act_bwd_colsum_mx(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                   float* colsum_partial, void* mx_q, void* mx_s, int B, int64_t HW, int C, int act, float slope,
                                   int dtype, void* stream);
/* MX quantisation along the last axis of a [P][C] matrix (dtype UIG_F32 / UIG_BF16), C a multiple of 32: per block of 32,
 * scale = 2^(floor(log2(max|x|)) - 8) as an E8M0 byte (127 for an all-zero block), elements = RNE(x / scale) in e4m3,
 * saturated to +-448.  q: [P][C] bytes, scales: [P][C/32] bytes. */
int uig_mx_quantize(const void* x, void* q, void* scales, long P, int C, int dtype, void* stream);
/* the same for many bf16 matrices in one launch (all fp8 weight operands after the per-step repack): items_dev = device array of
 * 40-byte records {const void* x; void* q; void* scales; int64 n8 = P*C/8; int64 block_end = inclusive prefix sum of ceil(n8/256)} */
int uig_mx_quantize_multi(const void* items_dev, int nitems, long total_blocks, void* stream);
/* Input gradient of a pad-1 REFLECTION 3x3 convolution on the MX fp8 kernel in ONE launch (the backward of reference
 * networks.py:ResnetBlock's ReflectionPad2d(1)+Conv2d, models/networks.py:349-377): dq / ds = the MX-quantised output gradient
 * [B,H,W,C], wq / ws (+ second network's wq2 / ws2 from image group_images on) = the MX-quantised tap-major weights of the transposed
 * gather.  The mirrored lines / columns are "mirror pixels" built in LDS: sums of two (four at the corners) de-quantised pixels,
 * re-quantised per 32-channel block with the rule of uig_mx_quantize.  res_add (nullable, [B,H,W,ldc] bf16): summed into dx in the
 * epilogue.  Shapes: W == 64, H % 4 == 0, H >= 8, C and Nrows multiples of 128 (…_applicable returns 1). */
int uig_conv3x3_mx_fp8_dgrad_mirror_applicable(int B, int H, int W, int Cin, int Nrows);
int uig_conv3x3_mx_fp8_dgrad_mirror(const void* dq, const void* ds, const void* wq, const void* ws, const void* wq2, const void* ws2,
                                    int group_images, const void* res_add, void* dx, int B, int H, int W, int Cin, int Nrows, int ldc,
                                    void* stream);
/* which strip kernel that launch runs on: 0 none (generic gather), 128 = 128x128 tiles, 256 = 256x128 tiles one per block,
 * 257 = 256x128 tiles on persistent blocks (tests assert the variant they mean to cover) */
int uig_conv_strip_tile(int B, int H, int W, int Cin, int Nrows, int Ho, int Wo, int dh_min, int dh_max, int dtype);

/* aten::convolution_backward(weight grad) — dW partials by split-K MFMA GEMM over pixels, then uig_wgrad_reduce.
 *   part[s][n][tap][c] = sum_{pixels m in split s} P[m][n] * Q[pix(m,tap)][c]
 * P: dense operand (B,Mh,Mw,Np) (dy for Conv2d, x for ConvTranspose2d); Q: gathered operand (B,Hq,Wq,Cq) read at
 * (i*stride + kh - pad, j*stride + kw - pad) with zero/reflect padding.  workspace: fp32[splits*Np*kH*kW*Cq].   */
size_t uig_wgrad_workspace_bytes(int Np, int Cq, int kH, int kW, int splits);
/* rows of the dense operand covered by one block of uig_wgrad_partial for this shape (16 / 128 / 256): the number of
 * output tiles, from which the caller chooses `splits`, is ceil(Np / rows) * ceil(kH*kW*Cq / 128). */
int uig_wgrad_tile_rows(int Np, int Mw, int dtype);
void uig_debug_set_wgrad_wide(int on);   /* tuning hook: 0 = never use the 256-row tile */
void uig_debug_set_wgrad_rows(int on);   /* A/B hook: 0 = never use the image-row kernel of the stride-1 3x3 convs */
void uig_debug_set_wgrad_rows_s2(int on);          /* image-row weight-gradient kernel, stride-2 form (round 3): 1 (default) / 0 = stride-2 layers on the generic split-K kernel */
/* 1 if a stride-2 3x3 transposed gather of this shape runs on the phase-fused kernel (conv_tr2.hip): such launches may emit
 * InstanceNorm statistics (in_partial of uig_conv_gather_ex) for 64 output channels too */
int uig_conv_tr2_applicable(int B, int H, int W, int Cin, int Nrows, int Nstore, int ldc, int dtype);
void uig_debug_set_tr2(int on);        /* 0: stride-2 transposed layers stay on the generic gather kernel */
void uig_debug_set_cin8(int on);         /* A/B hook: 0 = never use the LDS-resident-weights kernel of the 8-input-channel 7x7 convs */
void uig_debug_set_gemv(int on);         /* A/B hook: 0 = never use the one-wave-per-pixel kernel of the 1..4-output-channel convs */
void uig_debug_set_wgrad_head(int on);   /* A/B hook: 0 = never use the kernel of the 7x7 64 -> <=8 channel output conv */
/* number of splits (fp32 partial slabs) uig_wgrad_partial should be run with for this shape; target_blocks sizes the grid
 * of the generic kernel (the image-row kernel of the 3x3 stride-1 convs always runs one block per CU) */
int uig_wgrad_splits(int B, int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad,
                     int dtype, int target_blocks);
int uig_wgrad_partial(const void* P, const void* Q, float* workspace, int B, int Mh, int Mw, int Np,
                      int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int pad_mode,
                      int splits, int dtype, void* stream);
/* dW[d0][d1][tap] (+)= sum_s part[s][d0][tap][d1]  for d0 < D0, d1 < D1 (the real, unpadded channel counts) */
/* Two networks of the same layer shape in one partial launch (images [0, group_images) are the first network's): twice the
 * tiles, so half the splits and half the partial-slab traffic per network.  uig_wgrad_pair_splits returns the split count to
 * use (0 = bad grouping).  Workspace [2][splits][Np][kH*kW*Cq] floats; reduce each half with uig_wgrad_reduce*. */
int uig_wgrad_pair_splits(int B, int group_images, int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW,
                          int stride, int pad, int dtype, int target_blocks);
int uig_wgrad_partial_pair(const void* P, const void* Q, float* workspace, int B, int group_images, int Mh, int Mw,
                           int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int pad_mode,
                           int splits, int dtype, void* stream);
/* The same over TWO batches in one launch: the step's two generator passes use the same two weight sets - pass 1 on (P, Q) =
 * B1 images of which the first g1 are network 0's, pass 2 on (P2, Q2) = B2 images of which the first g2 are network
 * (swap2 ? 1 : 0)'s.  The fixed cost of a split-K launch (fill / drain, partial slabs, reduce) is paid once.
 * uig_wgrad_pair2_splits: the split count to use (the generic kernel accepts exactly this value), 0 = run the two launches
 * (bad grouping, or the 7x7 head kernel's shape).  Workspace and reduce as for uig_wgrad_partial_pair. */
int uig_wgrad_pair2_splits(int B1, int g1, int B2, int g2, int swap2, int Mh, int Mw, int Np, int Hq, int Wq, int Cq,
                           int kH, int kW, int stride, int pad, int dtype);
int uig_wgrad_partial_pair2(const void* P, const void* Q, const void* P2, const void* Q2, float* workspace,
                            int B1, int g1, int B2, int g2, int swap2, int Mh, int Mw, int Np, int Hq, int Wq, int Cq,
                            int kH, int kW, int stride, int pad, int pad_mode, int splits, int dtype, void* stream);
/* both halves of a uig_wgrad_partial_pair workspace in one launch (colsum_* NULL = no bias gradient on this launch) */
int uig_wgrad_reduce_pair(const float* workspace, float* dW_a, float* dW_b, int Np, int Cq, int taps, int splits,
                          int D0, int D1, int accumulate, const float* colsum_a, const float* colsum_b,
                          int nslab_a, int nslab_b, int C, int Nreal, float* db_a, float* db_b, int accumulate_db,
                          void* stream);
/* The same with TWO runs of column-sum slabs per network in the bias riders (the reduce behind uig_wgrad_partial_pair2, whose
 * partials cover both generator passes): db_x (+)= sum of colsum_x's nslab_x slabs + sum of colsum_x2's nslab_x2 slabs. */
int uig_wgrad_reduce_pair2(const float* workspace, float* dW_a, float* dW_b, int Np, int Cq, int taps, int splits,
                           int D0, int D1, int accumulate, const float* colsum_a, const float* colsum_b, int nslab_a, int nslab_b,
                           const float* colsum_a2, const float* colsum_b2, int nslab_a2, int nslab_b2,
                           int C, int Nreal, float* db_a, float* db_b, int accumulate_db, void* stream);
int uig_wgrad_reduce(const float* workspace, float* dW, int Np, int Cq, int taps, int splits,
                     int D0, int D1, int accumulate, void* stream);
/* uig_wgrad_reduce + the layer's bias gradient from the InstanceNorm backward's column-sum partials, in one launch */
int uig_wgrad_reduce_bias(const float* workspace, float* dW, int Np, int Cq, int taps, int splits,
                          int D0, int D1, int accumulate, const float* colsum_partial, int nslab_total, int C,
                          int Nreal, float* db, int accumulate_db, void* stream);
/* db[n] (+)= sum over all B*H*W pixels of dy[m][n], n < Nreal (aten::convolution_backward bias grad).
 * workspace: fp32[uig_colsum_workspace_floats(C)] */
size_t uig_colsum_workspace_floats(int C);
int uig_bias_grad(const void* dy, float* db, float* workspace, int64_t pixels, int C, int Nreal,
                  int accumulate, int dtype, void* stream);

/* fp32 torch-layout weight (D0,D1,kH,kW) -> packed [rows_padded][taps][cols_padded] in `dtype`, zero padded.
 * row_dim selects which torch dim becomes the GEMM row (UIG_PACK_ROW_DIM0 / _DIM1); flip=1 reverses taps. */
int uig_pack_weight(const float* w, void* wp, int D0, int D1, int kH, int kW, int row_dim, int flip,
                    int rows_padded, int cols_padded, int dtype, void* stream);

/* Every layer of every network in one launch.  items_dev: device array of nitems 48-byte records
 * {const float* w; void* dst; int D0, D1, taps, row_dim, rows_padded, cols_padded; int64 work_end} where work_end is the
 * inclusive prefix sum of uig_pack_tiles(...) of the records (one block transposes one tile); total_work = the last
 * work_end.  rows_padded must equal the real row count (only the column dimension is zero padded); taps <= 64. */
int uig_pack_tiles(int D0, int D1, int kH, int kW, int row_dim, int rows_padded, int cols_padded);
int uig_pack_weights_multi(const void* items_dev, int nitems, int64_t total_work, int dtype, void* stream);
/* The same with BOTH kernel-side operands of a layer from one pass over its fp32 weights (round 3).  56-byte records
 * {const float* w; void* dst; void* dst2; int D0, D1, taps, row_dim, cols_padded, cols2_padded; int64 work_end}: dst has rows = torch
 * dim row_dim and its columns padded to cols_padded, dst2 rows = the other dim, columns padded to cols2_padded; work_end = inclusive
 * prefix sum of uig_pack_tiles2(...). */
int uig_pack_tiles2(int D0, int D1, int kH, int kW, int row_dim, int cols_padded, int cols2_padded);
int uig_pack_weights_multi2(const void* items_dev, int nitems, int64_t total_work, int dtype, void* stream);

/* aten::instance_norm(use_input_stats=True, weight=None, eps) fused with ReLU / LeakyReLU and residual add:
 *   y = act((x - mean_bc) * rstd_bc) + (residual ? residual : 0);  stats fp32[B*C*2] = (mean, rstd) saved for bwd.
 * workspace: fp32[uig_instnorm_workspace_floats(B,HW,C)].                                                    */
size_t uig_instnorm_workspace_floats(int B, int64_t HW, int C);
int uig_instnorm_act_fwd(const void* x, const void* residual, void* y, float* stats, float* workspace,
                         int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream);
/* the same forward with the statistics partials already produced by uig_conv_gather_ex (nslab = HW/64 per image) */
int uig_instnorm_act_fwd_pre(const void* x, const void* residual, void* y, float* stats, const float* partial, int nslab,
                             int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream);
/* Inference form (no autograd state: nothing is saved for a backward pass): the apply kernel finalises the statistics
 * itself.  partial != NULL: the np-per-image epilogue partials of uig_conv_gather_ex -> ONE launch; partial == NULL: a
 * statistics pass into workspace (uig_instnorm_workspace_floats), then the fused apply.  Bit-identical to
 * uig_instnorm_act_fwd[_pre] (same fp64 association order). */
int uig_instnorm_act_fwd_infer(const void* x, const void* residual, void* y, const float* partial, int np, float* workspace,
                               int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream);
/* Round 4 (see uig_conv_gather_fin): the forward apply pass alone on FINAL statistics (stats fp32[B][C][2] = (mean, rstd));
 * mx_q / mx_s optional (both or none; bf16, C % 32 == 0: as uig_instnorm_act_fwd_mx). */
int uig_instnorm_apply_fwd(const void* x, const void* residual, void* y, const float* stats, void* mx_q, void* mx_s,
                           int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);
/* uig_instnorm_act_fwd whose statistics pass finalises itself through arrival tickets (>= B zero words, left zero): two launches
 * instead of three, bit-identical.  mx_q / mx_s optional. */
int uig_instnorm_act_fwd_t(const void* x, const void* residual, void* y, float* stats, float* workspace, unsigned* tickets,
                           void* mx_q, void* mx_s, int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream);
/* aten::native_batch_norm_backward (on the (1,B*C,H,W) view) fused with the activation's backward:
 *   g = dy * act'(xhat);  dx = rstd * (g - mean(g) - xhat * mean(g*xhat))                                   */
int uig_instnorm_act_bwd(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                         int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);

/* Same, and additionally emits per-block column sums of the dx it writes into colsum_partial
 * (fp32[uig_instnorm_bwd_colsum_slabs(B,HW,C,dtype) * C * 2]): the bias gradient of the convolution in front of this
 * InstanceNorm is then uig_bias_grad_from_partials(colsum_partial, db, slabs, ...) with no second pass over dx. */
int uig_instnorm_bwd_colsum_slabs(int B, int64_t HW, int C, int dtype);
/* the same with the backward statistics already produced by the launch that wrote dy (uig_conv_gather_bst / uig_conv3x3_mx_fp8:
 * partial fp32[B][nslab][C][2], nslab = HW/64): no statistics pass.  mx_q / mx_s optional (NULL, or as uig_instnorm_act_bwd_colsum_mx). */
int uig_instnorm_act_bwd_colsum_pre(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                    float* colsum_partial, const float* partial, int nslab, void* mx_q, void* mx_s,
                                    int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);
int uig_instnorm_act_bwd_colsum(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                float* colsum_partial, int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);
/* Round 4, the general form.  Where (mean g, mean g*xhat) come from: pre_gm != NULL - fp32[B][C][2], already final (finalised inside
 * the launch that wrote dy: uig_reflect3x3_dgrad_mirror_bst): the apply launch alone; else pre_partial != NULL - that launch's epilogue
 * partials (pre_nslab per image): finalize launch + apply (= uig_instnorm_act_bwd_colsum_pre); else this norm's own statistics pass,
 * which finalises itself through `tickets` (>= B zero words, left zero): two launches instead of three.  Bit-identical to the forms
 * above.  mx_q / mx_s optional (as uig_instnorm_act_bwd_colsum_mx). */
int uig_instnorm_act_bwd_colsum_t(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                  float* colsum_partial, const float* pre_partial, int pre_nslab, const float* pre_gm,
                                  unsigned* tickets, void* mx_q, void* mx_s,
                                  int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);
/* Round 4: the whole InstanceNorm backward in ONE launch and ONE pass over dy and x (statistics, finalize and apply fused; the blocks
 * of an image synchronise inside the kernel through arrival counters and keep their pixels in registers across the waits): replaces the
 * three launches of uig_instnorm_act_bwd_colsum.  uig_instnorm_bwd_fused_applicable -> NB (blocks = column-sum slabs per image) or 0
 * (shape too large for the whole grid to be resident at once: use the three-launch form).  partial fp32[B][NB][C][2], gm fp32[B][C][2]:
 * scratch; colsum_partial fp32[B][NB][C][2]; sync: >= 4*B zero 32-bit words, left zero; err: one word set to 1 if a bounded in-kernel
 * wait ran out (results invalid; never a hang).  Same arithmetic as the three-launch form; the statistics' fp32 partial sums cover other
 * pixel ranges, so results agree to rounding, not bitwise.  uig_debug_set_in_fused(0): the query answers 0 (A/B hook). */
void uig_debug_set_in_fused(int on);
int uig_instnorm_bwd_fused_applicable(int B, int64_t HW, int C, int dtype);
int uig_instnorm_act_bwd_fused(const void* dy, const void* x, const float* stats, void* dx, float* partial, float* gm,
                               float* colsum_partial, unsigned* sync, unsigned* err, void* mx_q, void* mx_s,
                               int B, int64_t HW, int C, int act, float slope, int dtype, void* stream);
int uig_bias_grad_from_partials(const float* colsum_partial, float* db, int nslab_total, int C, int Nreal,
                                int accumulate, void* stream);

/* aten::reflection_pad2d_backward: fold (B,H+2p,W+2p,C) -> (B,H,W,C) */
int uig_reflect_fold(const void* dyp, void* dx, int B, int H, int W, int C, int pad, int dtype, void* stream);
/* aten::tanh_backward / leaky_relu_backward / threshold_backward on the activation OUTPUT y: dx = dy * act'(y) */
int uig_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, float slope, int dtype, void* stream);

/* aten::l1_loss(mean) forward + gradient in one pass:  loss[0] = weight*mean|a-b| over n_real elements;
 * grad_a = weight*sign(a-b)/n_real (may be NULL).  n = physical element count (padded channels are 0 in both). */
int uig_l1_loss_fwd_bwd(const void* a, const void* b, float* loss, void* grad_a, float* workspace,
                        int64_t n, int64_t n_real, float weight, int dtype, void* stream);
/* aten::mse_loss(a, full_like(a, target), mean) + gradient: loss = weight*mean((a-t)^2); grad = weight*2(a-t)/n */
int uig_mse_const_fwd_bwd(const void* a, float target, float* loss, void* grad_a, float* workspace,
                          int64_t n, float weight, int dtype, void* stream);
size_t uig_loss_workspace_floats(void);
/* g_out = g * scalar[0]  (loss backward when the upstream gradient is a device scalar) */
int uig_scale_by_scalar(const void* g, const float* scalar, void* g_out, int64_t n, int dtype, void* stream);

/* aten::_fused_adam (amsgrad=False, maximize=False, weight_decay=0) over one flat fp32 buffer.
 * g is multiplied by grad_scale first (1/world_size after an all-reduce(sum)). step = 1,2,... */
int uig_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                  float eps, int step, float grad_scale, void* stream);

/* Same update, replayable from a HIP graph: the step counter and bias-correction scalars live in the 16-byte device
 * record state16 = {int step; float lr*lr_scale/bc1; float 1/sqrt(bc2); float lr_scale}; each call increments step on
 * the device.  lr_scale (the LR schedule's multiplier) is written by the host between replays; initialise it to 1.0f. */
int uig_adam_flat_graph(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                        float eps, void* state16, float grad_scale, void* stream);

/* layout plumbing at the module surface: arbitrary-strided fp32/bf16 (B,C,H,W) logical tensor <-> NHWC(Cp) */
int uig_to_nhwc(const void* src, int src_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                void* dst, int B, int C, int H, int W, int Cp, int dtype, void* stream);
int uig_from_nhwc(const void* src, int B, int C, int H, int W, int Cp, int dtype,
                  void* dst, int dst_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw, void* stream);

/* Input pipeline tail (the upstream recipe is torchvision's Resize(286, BICUBIC) on a PIL image ->
 * RandomCrop(256) -> RandomHorizontalFlip -> ToTensor -> Normalize(0.5, 0.5)) in one launch over a batch of decoded
 * images  src u8[B][Hs][Ws][3]  ->  out dtype[B][Ho][Wo][8] (channels 3..7 zero), values (x/255 - 0.5)/0.5.
 * The resize is separable resampling in Pillow's 8-bit convention (libImaging/Resample.c): horizontal pass, then
 * vertical pass, each  clip8((2^21 + sum_i px_i * k_i) >> 22)  with integer coefficients k = round(w * 2^22).
 *   kh int32[Wr][ksh], bh int32[Wr][2] = (first source column, tap count) per resized column; kv/bv likewise per
 *   resized row; crop_flip int32[B][3] = (x0, y0, flip) in the resized image, DEVICE memory (clamped to the valid range
 *   by the kernel): out[b][oy][ox] = resized[b][y0+oy][x0 + (flip ? Wo-1-ox : ox)].                              */
int uig_resize_crop_flip_normalize(const uint8_t* src, int B, int Hs, int Ws,
                                   const int32_t* kh, const int32_t* bh, int ksh,
                                   const int32_t* kv, const int32_t* bv, int ksv, int Hr, int Wr,
                                   const int32_t* crop_flip, void* out, int Ho, int Wo, int dtype, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UIG_H */
