// augment.hip — input pipeline tail on the device: decoded 8-bit RGB images ->
//   resize (separable resampling in Pillow's 8-bit fixed-point convention, horizontal pass then vertical pass, each
//   rounded and clipped to 8 bits)  ->  crop  ->  horizontal flip  ->  x/255, (x-0.5)/0.5  ->  NHWC, 8 channels (3 real)
// in the compute dtype, i.e. exactly the tensor the stem convolution gathers from.  Only the cropped pixels are
// resampled: an output pixel evaluates its <= ksv x ksh source window directly (integer MACs), so nothing but the
// source bytes is read and the result is written once with 16-byte stores.  The coefficient and bound tables come from
// the host (pipeline.py builds them in float64 like Pillow does); every table value is re-clamped here before it is
// used as an address, because the crop parameters live in device memory and cannot be validated by the host.
#include "uig_common.h"

#define AUG_BITS 22   // Pillow: PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ int aug_clip8(int v) {
    v >>= AUG_BITS;                 // arithmetic shift, as the C reference does on a signed int
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

template <typename T>
__global__ __launch_bounds__(256) void augment_kernel(const uint8_t* __restrict__ src, int Hs, int Ws,
                                                      const int32_t* __restrict__ kh, const int32_t* __restrict__ bh, int ksh,
                                                      const int32_t* __restrict__ kv, const int32_t* __restrict__ bv, int ksv,
                                                      int Hr, int Wr, const int32_t* __restrict__ params,
                                                      T* __restrict__ out, int Ho, int Wo) {
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y, b = blockIdx.z;
    if (ox >= Wo) return;
    int x0 = params[b * 3], y0 = params[b * 3 + 1];
    const int flip = params[b * 3 + 2];
    x0 = min(max(x0, 0), Wr - Wo);
    y0 = min(max(y0, 0), Hr - Ho);
    const int xr = x0 + (flip ? Wo - 1 - ox : ox), yr = y0 + oy;
    const int xmin = min(max(bh[xr * 2], 0), Ws - 1), xn = min(min(bh[xr * 2 + 1], ksh), Ws - xmin);
    const int ymin = min(max(bv[yr * 2], 0), Hs - 1), yn = min(min(bv[yr * 2 + 1], ksv), Hs - ymin);
    const uint8_t* sb = src + (long)b * Hs * Ws * 3;
    const int32_t* khx = kh + (long)xr * ksh;
    const int32_t* kvy = kv + (long)yr * ksv;
    int v0 = 1 << (AUG_BITS - 1), v1 = v0, v2 = v0;
    for (int j = 0; j < yn; ++j) {
        const uint8_t* row = sb + ((long)(ymin + j) * Ws + xmin) * 3;
        int h0 = 1 << (AUG_BITS - 1), h1 = h0, h2 = h0;
        for (int i = 0; i < xn; ++i) {
            const int k = khx[i];
            h0 += (int)row[i * 3] * k; h1 += (int)row[i * 3 + 1] * k; h2 += (int)row[i * 3 + 2] * k;
        }
        const int k = kvy[j];
        v0 += aug_clip8(h0) * k; v1 += aug_clip8(h1) * k; v2 += aug_clip8(h2) * k;
    }
    // ToTensor (x / 255 in fp32) then Normalize(0.5, 0.5): (t - 0.5) / 0.5; the division by 0.5 is an exact doubling
    float f[8];
    f[0] = (__fdiv_rn((float)aug_clip8(v0), 255.f) - 0.5f) * 2.f;
    f[1] = (__fdiv_rn((float)aug_clip8(v1), 255.f) - 0.5f) * 2.f;
    f[2] = (__fdiv_rn((float)aug_clip8(v2), 255.f) - 0.5f) * 2.f;
#pragma unroll
    for (int e = 3; e < 8; ++e) f[e] = 0.f;
    constexpr int E = ElemTraits<T>::E;
    T* o = out + (((long)b * Ho + oy) * Wo + ox) * 8;
#pragma unroll
    for (int c = 0; c < 8 / E; ++c) reinterpret_cast<u32x4_t*>(o)[c] = f32_to_chunk<T>(f + c * E);
}

extern "C" int uig_resize_crop_flip_normalize(const uint8_t* src, int B, int Hs, int Ws,
                                              const int32_t* kh, const int32_t* bh, int ksh,
                                              const int32_t* kv, const int32_t* bv, int ksv, int Hr, int Wr,
                                              const int32_t* crop_flip, void* out, int Ho, int Wo, int dtype, void* stream) {
    UIG_CHECK_ARG(src && kh && bh && kv && bv && crop_flip && out, "uig_resize_crop_flip_normalize: null pointer");
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_resize_crop_flip_normalize: bad dtype %d", dtype);
    UIG_CHECK_ARG(B > 0 && B <= 65535 && Hs > 0 && Ws > 0 && ksh > 0 && ksv > 0,
                  "uig_resize_crop_flip_normalize: bad shape B=%d Hs=%d Ws=%d ksh=%d ksv=%d", B, Hs, Ws, ksh, ksv);
    UIG_CHECK_ARG(Ho > 0 && Wo > 0 && Ho <= Hr && Wo <= Wr && Ho <= 65535,
                  "uig_resize_crop_flip_normalize: crop %dx%d does not fit the resized image %dx%d", Ho, Wo, Hr, Wr);
    const dim3 grid((Wo + 255) / 256, Ho, B);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((augment_kernel<bf16_t>), grid, dim3(256), 0, s, src, Hs, Ws, kh, bh, ksh, kv, bv, ksv, Hr, Wr, crop_flip, (bf16_t*)out, Ho, Wo);
    else
        hipLaunchKernelGGL((augment_kernel<float>), grid, dim3(256), 0, s, src, Hs, Ws, kh, bh, ksh, kv, bv, ksv, Hr, Wr, crop_flip, (float*)out, Ho, Wo);
    UIG_LAUNCH_CHECK("uig_resize_crop_flip_normalize");
    return 0;
}
