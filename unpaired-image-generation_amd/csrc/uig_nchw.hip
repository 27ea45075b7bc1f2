// uig_nchw.hip — the contiguous-NCHW convenience boundary of SURVEY.md §8(b): one call = aten::convolution (Conv2d forward,
// optional reflection padding) on NCHW activations and the OIHW fp32 weight, with the internal repacks (NCHW -> padded NHWC,
// weight -> [N][tap][C], padded NHWC -> NCHW) INSIDE the call and therefore inside its time.  The hot path does not use this
// entry: it keeps activations NHWC between layers and repacks the weights once per optimiser step.
#include "uig_common.h"

extern "C" int uig_conv_gather(const void* x, const void* wp, const float* bias, void* y,
                               int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                               int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                               int act, float slope, int dtype, void* stream);

static size_t al256(size_t v) { return (v + 255) / 256 * 256; }
static int pad8i(int c) { return (c + 7) / 8 * 8; }
// input channels as the gather kernels take them: a multiple of the K-step (64 bf16 / 32 f32), or a power of two below it
static int pad_cin(int c, int dtype) {
    const int bk = dtype == UIG_BF16 ? 64 : 32;
    if (c >= bk) return (c + bk - 1) / bk * bk;
    int p = 8;
    while (p < c) p *= 2;
    return p;
}

extern "C" size_t uig_conv2d_fwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int dtype) {
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    const int Ho = (H + 2 * pad - kH) / stride + 1, Wo = (W + 2 * pad - kW) / stride + 1;
    return al256((size_t)B * H * W * pad_cin(Cin, dtype) * es) + al256((size_t)Cout * kH * kW * pad_cin(Cin, dtype) * es) + al256((size_t)B * Ho * Wo * pad8i(Cout) * es);
}

// x: contiguous (B,Cin,H,W) in `dtype`; w: contiguous fp32 (Cout,Cin,kH,kW); bias: fp32[Cout] or NULL; y: contiguous
// (B,Cout,Ho,Wo) in `dtype`; pad_mode UIG_PAD_ZERO / UIG_PAD_REFLECT; workspace: uig_conv2d_fwd_workspace_bytes(...) bytes.
extern "C" int uig_conv2d_fwd(const void* x, const float* w, const float* bias, void* y,
                              int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int pad_mode, int dtype,
                              void* workspace, size_t workspace_bytes, void* stream) {
    UIG_CHECK_ARG(x && w && y && workspace, "uig_conv2d_fwd: null pointer");
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_conv2d_fwd: bad dtype %d", dtype);
    UIG_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && kH > 0 && kW > 0 && stride > 0 && pad >= 0, "uig_conv2d_fwd: bad shape");
    UIG_CHECK_ARG(workspace_bytes >= uig_conv2d_fwd_workspace_bytes(B, Cin, H, W, Cout, kH, kW, stride, pad, dtype),
                  "uig_conv2d_fwd: workspace too small (%zu bytes)", workspace_bytes);
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    const int Cp = pad_cin(Cin, dtype), Np = pad8i(Cout);
    const int Ho = (H + 2 * pad - kH) / stride + 1, Wo = (W + 2 * pad - kW) / stride + 1;
    char* xp = static_cast<char*>(workspace);
    char* wp = xp + al256((size_t)B * H * W * Cp * es);
    char* yp = wp + al256((size_t)Cout * kH * kW * Cp * es);
    int rc = uig_to_nhwc(x, dtype, (int64_t)Cin * H * W, (int64_t)H * W, W, 1, xp, B, Cin, H, W, Cp, dtype, stream);
    if (rc) return rc;
    rc = uig_pack_weight(w, wp, Cout, Cin, kH, kW, UIG_PACK_ROW_DIM0, 0, Cout, Cp, dtype, stream);
    if (rc) return rc;
    rc = uig_conv_gather(xp, wp, bias, yp, B, H, W, Cp, Cout, kH, kW, stride, pad, pad_mode, UIG_GATHER_DIRECT, Ho, Wo, Np, Np, UIG_ACT_NONE, 0.f, dtype, stream);
    if (rc) return rc;
    return uig_from_nhwc(yp, B, Cout, Ho, Wo, Np, dtype, y, dtype, (int64_t)Cout * Ho * Wo, (int64_t)Ho * Wo, Wo, 1, stream);
}
