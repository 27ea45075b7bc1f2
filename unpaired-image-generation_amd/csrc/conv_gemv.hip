// conv_gemv.hip — direct convolution with 1..4 output channels and many input channels (the PatchGAN's last layer:
// 4x4, 512 -> 1 on 31x31 -> 30x30): every output value is a 8192-long dot product, i.e. a batch of GEMVs.
// On the tiled MFMA kernel this layer fills 29 blocks (7200 pixels / 256-row tiles), each walking 128 dependent K-steps:
// 71-87 us per launch for 0.12 GFLOP.  Here ONE WAVE owns one output pixel: per tap every lane loads 16 bytes of the input
// pixel (64 lanes x 8 bf16 = 512 channels per pass) and of each weight row, multiplies in fp32 and the wave reduces by
// shuffles.  Thousands of independent waves, no LDS, no barriers: the launch is bandwidth-shaped (L2 hits: neighbouring
// output pixels share 12 of their 16 taps).
#include "uig_common.h"
#include <algorithm>

struct GemvDesc {
    int B, H, W, Cin, Ho, Wo, Nrows, kH, kW, stride, pad, pad_mode, ldw, ldc, Nstore, act, group_images;
    float slope;
    const void* wp2; const float* bias2;
};

template <typename T, int NR>
__global__ __launch_bounds__(256) void conv_gemv_kernel(const T* __restrict__ x, const T* __restrict__ wp_, const float* __restrict__ bias_,
                                                         T* __restrict__ y, const GemvDesc d) {
    constexpr int E = ElemTraits<T>::E;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long M = (long)d.B * d.Ho * d.Wo;
    const int ncp = d.Cin / (64 * E);                      // 64-lane passes per tap
    for (long m = (long)blockIdx.x * 4 + wave; m < M; m += (long)gridDim.x * 4) {
        const int wo = (int)(m % d.Wo); const long t = m / d.Wo; const int ho = (int)(t % d.Ho), img = (int)(t / d.Ho);
        const bool g2 = d.wp2 != nullptr && img >= d.group_images;
        const T* wp = g2 ? static_cast<const T*>(d.wp2) : wp_;
        const float* bias = g2 ? d.bias2 : bias_;
        float acc[NR];
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[n] = 0.f;
        for (int kh = 0; kh < d.kH; ++kh) {
            int hi = ho * d.stride + kh - d.pad;
            bool okh = (unsigned)hi < (unsigned)d.H;
            if (d.pad_mode == UIG_PAD_REFLECT) { hi = reflect_idx(hi, d.H); okh = true; }
            for (int kw = 0; kw < d.kW; ++kw) {
                int wi = wo * d.stride + kw - d.pad;
                bool ok = okh && (unsigned)wi < (unsigned)d.W;
                if (d.pad_mode == UIG_PAD_REFLECT) { wi = reflect_idx(wi, d.W); ok = true; }
                if (!ok) continue;                          // wave-uniform: the whole wave shares (ho, wo)
                const T* xp = x + (((long)img * d.H + hi) * d.W + wi) * d.Cin + lane * E;
                const T* wq = wp + (long)(kh * d.kW + kw) * d.Cin + lane * E;
                for (int c = 0; c < ncp; ++c) {
                    float xv[E];
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(xp + c * 64 * E), xv);
#pragma unroll
                    for (int n = 0; n < NR; ++n) {
                        if (n < d.Nrows) {
                            float wv[E];
                            chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(wq + (long)n * d.ldw + c * 64 * E), wv);
#pragma unroll
                            for (int e = 0; e < E; ++e) acc[n] = fmaf(xv[e], wv[e], acc[n]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int n = 0; n < NR; ++n) acc[n] = wave_sum(acc[n]);
        if (lane == 0) {
            T* yp = y + m * d.ldc;
#pragma unroll
            for (int n = 0; n < NR; ++n)
                if (n < d.Nstore) ElemTraits<T>::st(yp + n, n < d.Nrows ? apply_act(acc[n] + (bias ? bias[n] : 0.f), d.act, d.slope) : 0.f);
        }
    }
}

static int g_gemv_mode = 1;     // A/B and parity hook
extern "C" void uig_debug_set_gemv(int on) { g_gemv_mode = on; }

// Returns 1 if this kernel took the launch, 0 if the shape does not qualify (caller falls back).
int uig_try_conv_gemv(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                      void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad, int pad_mode,
                      int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype, hipStream_t s, int* rc_out) {
    const int E = dtype == UIG_BF16 ? 8 : 4;
    if (!g_gemv_mode || Nrows > 4 || Nstore > 4 || Cin % (64 * E) != 0 || Cin < 256) return 0;
    GemvDesc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Ho = Ho; d.Wo = Wo; d.Nrows = Nrows; d.kH = kH; d.kW = kW; d.stride = stride; d.pad = pad;
    d.pad_mode = pad_mode; d.ldw = kH * kW * Cin; d.ldc = ldc; d.Nstore = Nstore; d.act = act; d.slope = slope;
    d.group_images = group_images; d.wp2 = wp2; d.bias2 = bias2;
    const long M = (long)B * Ho * Wo;
    const int blocks = (int)std::min<long>((M + 3) / 4, 16384);
    uig_note_conv_kernel(UIG_K_GEMV);
    if (dtype == UIG_BF16) hipLaunchKernelGGL((conv_gemv_kernel<bf16_t, 4>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (bf16_t*)y, d);
    else hipLaunchKernelGGL((conv_gemv_kernel<float, 4>), dim3(blocks), dim3(256), 0, s, (const float*)x, (const float*)wp, bias, (float*)y, d);
    hipError_t e_ = hipGetLastError();
    *rc_out = e_ == hipSuccess ? 0 : uig_set_error((int)e_, "uig_conv_gather(gemv): launch failed: %s", hipGetErrorString(e_));
    return 1;
}
