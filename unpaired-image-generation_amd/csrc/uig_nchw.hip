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

// ---- round 4: the other two operators of §8(b) on the same contiguous-NCHW boundary (missing #6) ------------------
// Channel padding of a tensor that is BOTH a gather operand and an InstanceNorm-style column-sum operand (dy of the backward): the
// smallest power of two >= max(c, 8) - a power of two below the K-step or a multiple of it above (gather kernels), and a divisor
// pattern the column-sum kernel accepts (C / chunk divides 256).
static int pad_pow2(int c) { int p = 8; while (p < c) p *= 2; return p; }

// aten::convolution for ConvTranspose2d(kernel 3, stride 2, padding 1, output_padding 1) - the up-sampling layers of the path:
// x (B,Cin,H,W), w (Cin,Cout,3,3) fp32 (torch's transposed-conv layout), bias fp32[Cout] or NULL, y (B,Cout,2H,2W).
extern "C" size_t uig_conv_transpose2d_fwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int dtype) {
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    return al256((size_t)B * H * W * pad_cin(Cin, dtype) * es) + al256((size_t)Cout * 9 * pad_cin(Cin, dtype) * es) + al256((size_t)B * 4 * H * W * pad8i(Cout) * es);
}
extern "C" int uig_conv_transpose2d_fwd(const void* x, const float* w, const float* bias, void* y, int B, int Cin, int H, int W, int Cout, int dtype,
                                        void* workspace, size_t workspace_bytes, void* stream) {
    UIG_CHECK_ARG(x && w && y && workspace, "uig_conv_transpose2d_fwd: null pointer");
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_conv_transpose2d_fwd: bad dtype %d", dtype);
    UIG_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0, "uig_conv_transpose2d_fwd: bad shape");
    UIG_CHECK_ARG(workspace_bytes >= uig_conv_transpose2d_fwd_workspace_bytes(B, Cin, H, W, Cout, dtype), "uig_conv_transpose2d_fwd: workspace too small (%zu bytes)", workspace_bytes);
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    const int Cp = pad_cin(Cin, dtype), Np = pad8i(Cout);
    char* xp = static_cast<char*>(workspace);
    char* wp = xp + al256((size_t)B * H * W * Cp * es);
    char* yp = wp + al256((size_t)Cout * 9 * Cp * es);
    int rc = uig_to_nhwc(x, dtype, (int64_t)Cin * H * W, (int64_t)H * W, W, 1, xp, B, Cin, H, W, Cp, dtype, stream);
    if (rc) return rc;
    rc = uig_pack_weight(w, wp, Cin, Cout, 3, 3, UIG_PACK_ROW_DIM1, 0, Cout, Cp, dtype, stream);      // rows = Cout = dim 1 of the (Cin, Cout, k, k) weight
    if (rc) return rc;
    rc = uig_conv_gather(xp, wp, bias, yp, B, H, W, Cp, Cout, 3, 3, 2, 1, UIG_PAD_ZERO, UIG_GATHER_TRANSPOSED, 2 * H, 2 * W, Np, Np, UIG_ACT_NONE, 0.f, dtype, stream);
    if (rc) return rc;
    return uig_from_nhwc(yp, B, Cout, 2 * H, 2 * W, Np, dtype, y, dtype, (int64_t)Cout * 4 * H * W, (int64_t)4 * H * W, 2 * W, 1, stream);
}

// aten::convolution_backward for Conv2d (groups 1, dilation 1), output_mask = which of dx / dW / db are non-NULL:
// dy (B,Cout,Ho,Wo) and x (B,Cin,H,W) in `dtype`, w (Cout,Cin,kH,kW) fp32; dx (B,Cin,H,W) in `dtype`, dW (Cout,Cin,kH,kW) fp32,
// db fp32[Cout].  pad_mode UIG_PAD_REFLECT: the gradient w.r.t. the reflection-padded input, folded back (aten::reflection_pad2d_backward).
extern "C" size_t uig_conv2d_bwd_workspace_bytes(int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int dtype) {
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    const int Ho = (H + 2 * pad - kH) / stride + 1, Wo = (W + 2 * pad - kW) / stride + 1;
    const int Cp = pad_cin(Cin, dtype), Cop = pad_pow2(Cout), c8 = pad8i(Cin);
    const int splits = uig_wgrad_splits(B, Ho, Wo, Cop, H, W, Cp, kH, kW, stride, pad, dtype, 512);
    return al256((size_t)B * H * W * Cp * es) + al256((size_t)B * Ho * Wo * Cop * es) + al256((size_t)Cin * kH * kW * Cop * es) +
           al256((size_t)B * (H + 2 * pad) * (W + 2 * pad) * c8 * es) + al256((size_t)B * H * W * c8 * es) +
           al256(uig_wgrad_workspace_bytes(Cop, Cp, kH, kW, splits)) + al256(uig_colsum_workspace_floats(Cop) * sizeof(float));
}
extern "C" int uig_conv2d_bwd(const void* dy, const void* x, const float* w, void* dx, float* dW, float* db,
                              int B, int Cin, int H, int W, int Cout, int kH, int kW, int stride, int pad, int pad_mode, int dtype,
                              void* workspace, size_t workspace_bytes, void* stream) {
    UIG_CHECK_ARG(dy && workspace && (dx || dW || db), "uig_conv2d_bwd: null pointer");
    UIG_CHECK_ARG((dx == nullptr || w) && (dW == nullptr || x), "uig_conv2d_bwd: dx needs w, dW needs x");
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_conv2d_bwd: bad dtype %d", dtype);
    UIG_CHECK_ARG(B > 0 && Cin > 0 && Cout > 0 && H > 0 && W > 0 && kH > 0 && kW > 0 && (stride == 1 || stride == 2) && pad >= 0, "uig_conv2d_bwd: bad shape");
    UIG_CHECK_ARG(pad_mode == UIG_PAD_ZERO || (pad_mode == UIG_PAD_REFLECT && pad < H && pad < W), "uig_conv2d_bwd: bad pad_mode %d", pad_mode);
    UIG_CHECK_ARG(pad_pow2(Cout) <= (dtype == UIG_BF16 ? 2048 : 1024), "uig_conv2d_bwd: Cout=%d too wide for the column-sum pass", Cout);
    UIG_CHECK_ARG(workspace_bytes >= uig_conv2d_bwd_workspace_bytes(B, Cin, H, W, Cout, kH, kW, stride, pad, dtype), "uig_conv2d_bwd: workspace too small (%zu bytes)", workspace_bytes);
    const size_t es = dtype == UIG_BF16 ? 2 : 4;
    const int Ho = (H + 2 * pad - kH) / stride + 1, Wo = (W + 2 * pad - kW) / stride + 1;
    const int Cp = pad_cin(Cin, dtype), Cop = pad_pow2(Cout), c8 = pad8i(Cin);
    const int splits = uig_wgrad_splits(B, Ho, Wo, Cop, H, W, Cp, kH, kW, stride, pad, dtype, 512);
    char* xp = static_cast<char*>(workspace);
    char* dyp = xp + al256((size_t)B * H * W * Cp * es);
    char* wpd = dyp + al256((size_t)B * Ho * Wo * Cop * es);
    char* dxpad = wpd + al256((size_t)Cin * kH * kW * Cop * es);
    char* dxp = dxpad + al256((size_t)B * (H + 2 * pad) * (W + 2 * pad) * c8 * es);
    char* wsw = dxp + al256((size_t)B * H * W * c8 * es);
    char* wsc = wsw + al256(uig_wgrad_workspace_bytes(Cop, Cp, kH, kW, splits));
    int rc = uig_to_nhwc(dy, dtype, (int64_t)Cout * Ho * Wo, (int64_t)Ho * Wo, Wo, 1, dyp, B, Cout, Ho, Wo, Cop, dtype, stream);
    if (rc) return rc;
    if (dx != nullptr) {
        rc = uig_pack_weight(w, wpd, Cout, Cin, kH, kW, UIG_PACK_ROW_DIM1, 0, Cin, Cop, dtype, stream);      // rows = Cin, reduction over (tap, Cout)
        if (rc) return rc;
        if (pad_mode == UIG_PAD_REFLECT && pad > 0) {
            rc = uig_conv_gather(dyp, wpd, nullptr, dxpad, B, Ho, Wo, Cop, Cin, kH, kW, stride, 0, UIG_PAD_ZERO, UIG_GATHER_TRANSPOSED,
                                 H + 2 * pad, W + 2 * pad, c8, c8, UIG_ACT_NONE, 0.f, dtype, stream);
            if (rc) return rc;
            rc = uig_reflect_fold(dxpad, dxp, B, H, W, c8, pad, dtype, stream);
        } else {
            rc = uig_conv_gather(dyp, wpd, nullptr, dxp, B, Ho, Wo, Cop, Cin, kH, kW, stride, pad, UIG_PAD_ZERO, UIG_GATHER_TRANSPOSED, H, W, c8, c8,
                                 UIG_ACT_NONE, 0.f, dtype, stream);
        }
        if (rc) return rc;
        rc = uig_from_nhwc(dxp, B, Cin, H, W, c8, dtype, dx, dtype, (int64_t)Cin * H * W, (int64_t)H * W, W, 1, stream);
        if (rc) return rc;
    }
    if (dW != nullptr) {
        rc = uig_to_nhwc(x, dtype, (int64_t)Cin * H * W, (int64_t)H * W, W, 1, xp, B, Cin, H, W, Cp, dtype, stream);
        if (rc) return rc;
        rc = uig_wgrad_partial(dyp, xp, reinterpret_cast<float*>(wsw), B, Ho, Wo, Cop, H, W, Cp, kH, kW, stride, pad, pad_mode, splits, dtype, stream);
        if (rc) return rc;
        rc = uig_wgrad_reduce(reinterpret_cast<float*>(wsw), dW, Cop, Cp, kH * kW, splits, Cout, Cin, 0, stream);
        if (rc) return rc;
    }
    if (db != nullptr) rc = uig_bias_grad(dyp, db, reinterpret_cast<float*>(wsc), (int64_t)B * Ho * Wo, Cop, Cout, 0, dtype, stream);
    return rc;
}
