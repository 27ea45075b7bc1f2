// conv_rowstrip.hip — stride-1 k x k convolution with FEW output channels (N <= 16) and a large tap count: the 7x7
// generator head (64 -> 3, forward) and the input gradient of the 7x7 stem (64 -> 3).
//
// The generic implicit-GEMM kernel stages a 256-pixel x 128-byte im2col tile for every one of the 49 taps while a wave
// only has 8 MFMAs to run on it (N = 16): 49 x 34 KB of L2->LDS traffic per 256 output pixels (267 us per 8 images).
// Here a block owns one 256-pixel segment of an output row and walks the kernel ROWS: per (channel chunk, kh) it loads
// ONE input row segment (256 + k - 1 pixels x 128 B, reflection / zero padding resolved per pixel in the DMA source
// address) plus the k weight tiles of that kernel row (k x 16 x 128 B) and generates the k taps of the row by shifted
// fragment reads - 47 KB per 7 taps instead of 238 KB, one barrier per kernel row.  LDS-DMA staging, double buffered,
// source-side XOR swizzle and MFMA 16x16 tiles exactly as in conv_igemm.hip / conv_strip.hip.
#include "uig_common.h"
#include <algorithm>

struct RowStripDesc {
    int B, H, W, Cin;
    int Ho, Wo;
    int k, R;                 // kernel size, halo (max |dw|)
    int pad_mode;
    int Nrows, ldw, ldc, Nstore;
    int act; float slope;
    unsigned x_bytes, w_bytes;
    int tap[64];              // (dh + 128) | (dw + 128) << 8 | weight-tap-index << 16, kernel-row major
    const void* wp2;          // paired launch: images >= group_images use wp2 / bias2
    const float* bias2;
    int group_images;
};

template <typename T> struct MmaR;
template <> struct MmaR<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct MmaR<float> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
    }
};

template <typename T, int KMAX, int NW>
__global__ __launch_bounds__(64 * NW, 2)
void conv_rowstrip_kernel(const T* __restrict__ x, const T* __restrict__ wp_, const float* __restrict__ bias_, T* __restrict__ y,
                          const RowStripDesc d) {
    constexpr int E = ElemTraits<T>::E;
    constexpr int BK = 8 * E;
    constexpr int BM = 256, MT = BM / (16 * NW);          // NW waves x (256 / NW) pixels, one 16-channel MFMA column
    constexpr int SROWS = (BM + KMAX - 1 + 7) / 8 * 8;    // strip rows (pixels), padded to whole 1-KiB DMA pieces
    constexpr int SPIECES = SROWS / 8;
    constexpr int SBUF = SROWS * 128;
    constexpr int WBUF = KMAX * 16 * 128;                 // k weight tiles of 16 rows
    constexpr int STAGE = SBUF + WBUF;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [stage 0: strip | weights][stage 1: ...]
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l8 = lane >> 3, ls = lane & 7, l16 = lane & 15, q = lane >> 4;

    // ---- tile: (image, output row, 256-pixel segment)
    const int segs = d.Wo / BM;
    int t = blockIdx.x;
    const int seg = t % segs; t /= segs;
    const int ho = t % d.Ho, img = t / d.Ho;
    const int w0 = seg * BM;
    const int Cin = d.Cin, k = d.k, R = d.R;
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    const bool g2 = d.wp2 != nullptr && img >= d.group_images;
    const T* wp = g2 ? static_cast<const T*>(d.wp2) : wp_;
    const float* bias = g2 ? d.bias2 : bias_;

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, d.w_bytes, 0x00020000);

    // group g = (channel chunk cc, kernel row kh): the strip is input row (ho + dh[kh]), pixels w0 - R .. w0 + BM - 1 + R
    const int ncc = Cin / BK;
    const int G = ncc * k;
    auto issue_group = [&](int g, int stage) {
        const int cc = g / k, kh = g - cc * k;
        const int te0 = __builtin_amdgcn_readfirstlane(d.tap[kh * k]);
        const int hi = ho + (te0 & 255) - 128;
        const bool hin = (unsigned)hi < (unsigned)d.H;
        const int hr = refl ? reflect_idx(hi, d.H) : hi;
        const int rowbase = (img * d.H + hr) * d.W;
        lds_ptr_t sdst = (lds_ptr_t)smem + stage * STAGE;
        const int soff = __builtin_amdgcn_readfirstlane(cc * BK * (int)sizeof(T));
        for (int j = wave; j < SPIECES; j += NW) {          // strip pieces: rows 8j .. 8j+7 of the strip
            const int s = 8 * j + l8;
            const int wi = w0 - R + s;
            const bool win = (unsigned)wi < (unsigned)d.W;
            const int wr = refl ? reflect_idx(wi, d.W) : wi;
            const bool ok = (s < BM + 2 * R) & (refl | (hin & win));
            const int csrc = ls ^ ((s >> 1) & 7);
            const unsigned off = ok ? (unsigned)(((rowbase + wr) * Cin + csrc * E) * (int)sizeof(T)) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(sdst + j * 1024), 16, (int)off, soff, 0, 0);
        }
        lds_ptr_t wdst = sdst + SBUF;
        for (int j = wave; j < 2 * k; j += NW) {            // weight pieces: tap kw = j / 2, rows 8 (j % 2) .. + 7
            const int kw = j >> 1, n = (j & 1) * 8 + l8;
            const int te = d.tap[kh * k + kw];
            const int csrc = ls ^ ((n >> 1) & 7);
            const unsigned off = (n < d.Nrows) ? (unsigned)((n * d.ldw + (te >> 16) * Cin + csrc * E) * (int)sizeof(T)) : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(wdst + j * 1024), 16, (int)off, soff, 0, 0);
        }
    };

    f32x4_t acc[MT];
#pragma unroll
    for (int b = 0; b < MT; ++b) acc[b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    issue_group(0, 0);
    const int pl = wave * (BM / NW) + l16;                 // this lane's pixel (tile-local) for m-tile 0
    const int wsw = (l16 >> 1) & 7;
    for (int g = 0; g < G; ++g) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        if (g + 1 < G) issue_group(g + 1, (g + 1) & 1);
        const int kh = g % k;
        const unsigned char* sx = smem + (g & 1) * STAGE;
        const unsigned char* sw = sx + SBUF + l16 * 128;
        for (int kw = 0; kw < k; ++kw) {
            const int sft = ((d.tap[kh * k + kw] >> 8) & 255) - 128 + R;      // strip shift of this tap (block-uniform)
            unsigned xa[MT];                                                   // byte address of chunk half 0; half 1 is ^ 64
#pragma unroll
            for (int b = 0; b < MT; ++b) {
                const int s = pl + b * 16 + sft;
                xa[b] = (unsigned)(s * 128 + ((q ^ ((s >> 1) & 7)) << 4));
            }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const u32x4_t wf = *reinterpret_cast<const u32x4_t*>(sw + kw * 2048 + (((q + 4 * h) ^ wsw) << 4));
#pragma unroll
                for (int b = 0; b < MT; ++b) {
                    const u32x4_t xf = *reinterpret_cast<const u32x4_t*>(sx + (xa[b] ^ (unsigned)(h << 6)));
                    MmaR<T>::run(wf, xf, acc[b]);
                }
            }
        }
    }

    // ---- epilogue: lane holds channels 4q .. 4q+3 of pixel column l16 (N <= 16: direct stores)
    const bool vec_ok = ((d.Nstore & 3) == 0) && ((d.ldc & 3) == 0);
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        const int wo = w0 + pl + b * 16;
        T* yp = y + ((long)(img * d.Ho + ho) * d.Wo + wo) * d.ldc;
        const int n = 4 * q;
        if (n >= d.Nstore) continue;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float bv = (bias != nullptr && n + e < d.Nrows) ? bias[n + e] : 0.f;
            v[e] = apply_act(acc[b][e] + bv, d.act, d.slope);
        }
        if (vec_ok) {
            if constexpr (sizeof(T) == 4) {
                *reinterpret_cast<f32x4_t*>(yp + n) = f32x4_t{v[0], v[1], v[2], v[3]};
            } else {
                u32x2_t pk;
                pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                *reinterpret_cast<u32x2_t*>(yp + n) = pk;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (n + e < d.Nstore) ElemTraits<T>::st(yp + n + e, v[e]);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// <= 4 output channels (the 64 -> 3 head): with the channels alone on the MFMA's 16-wide N side 13 of 16 columns are padding.
// Put the HORIZONTAL TAPS there instead: per kernel row kh one GEMM  P[s][kw*Nc + co] += X[row ho+kh-3][s][ci] * W[co][kh][kw][ci]
// over all 262 input pixels s of the segment (N = 7*Nc = 21 of 32 columns used, K accumulates over kh and ci), and ONE
// shift-add at the end,  y[w][co] = sum_kw P[w + kw][kw*Nc + co],  through LDS.  252 MFMAs per 256 output pixels instead of
// 784, a quarter of the LDS fragment reads, and a 3-stage DMA ring (48 KB per kernel row) because a kernel row is now only
// ~0.4 us of math.  bf16, k = 7, zero or reflection padding, taps in natural order.
// epilogue activation of the <= 4-channel kernels: one thread finishes a whole pixel, so the library tanhf (~150 instructions,
// measured 2.5 us per output row of a block) is replaced by 1 - 2 / (exp(2x) + 1) on the fast exponential (abs. error ~1e-6,
// far below the bf16 output's rounding); the padded channels are written as zeros without evaluating anything
__device__ __forceinline__ float head_act(float v, int act, float slope) {
    if (act == UIG_ACT_TANH) {
        const float a = fminf(fabsf(v), 15.f);
        const float r = 1.f - 2.f / (__expf(2.f * a) + 1.f);
        return v < 0.f ? -r : r;
    }
    return apply_act(v, act, slope);
}

namespace {
constexpr int HR_ROWS = 4;                            // output rows per block: 10 input-row strips serve 4 output rows (28 for 4 single rows)
constexpr int HR_SROWS = 320;                         // strip rows per stage: 262 used, padded so that every wave issues 5 pieces
constexpr int HR_SBUF = HR_SROWS * 128;               // 40960 B
constexpr int HR_NST = 3;
constexpr int HR_WOFF = HR_NST * HR_SBUF;             // weights: 7 kh x 32 (kw, co) rows x 128 B, resident for the block
constexpr int HR_SMEM = HR_WOFF + 7 * 32 * 128;       // 151552 B
constexpr int HR_PER = 5;                             // DMA instructions per wave per input row
}

__global__ __launch_bounds__(512, 2)
void conv_headrow_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp_, const float* __restrict__ bias_,
                         bf16_t* __restrict__ y, const RowStripDesc d, int P_, int flip, int segw) {
    // P_: input pixel of (output o, kernel index k) is o + k - P_ (3: same-size conv; 6: the "full" correlation that yields the
    // padded input gradient of a 7x7 conv).  flip: kernel index k addresses weight tap 6 - k (transposed gather).  segw: output
    // pixels per segment (256, or the whole row when it has <= 266 pixels: 266 + 6 strip pixels still fit the 17 MFMA tiles).
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l8 = lane >> 3, ls = lane & 7, l16 = lane & 15, q = lane >> 4;
    const int segs = (d.Wo + segw - 1) / segw, nrg = (d.Ho + HR_ROWS - 1) / HR_ROWS;
    int t = blockIdx.x;
    const int seg = t % segs; t /= segs;
    const int rg = t % nrg, img = t / nrg;
    const int w0 = seg * segw, ho0 = rg * HR_ROWS;
    const int Nc = d.Nrows;
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    const bool g2 = d.wp2 != nullptr && img >= d.group_images;
    const bf16_t* wp = g2 ? static_cast<const bf16_t*>(d.wp2) : wp_;
    const float* bias = g2 ? d.bias2 : bias_;
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, d.x_bytes, 0x00020000);

    // group g = input row ho0 - P + g (g = 0 .. ROWS + 5): pixels w0 - P .. w0 - P + segw + 5 as strip rows 0 .. segw + 5
    constexpr int G = HR_ROWS + 6;
    auto issue_group = [&](int g, int stage) {
        const int hi = ho0 - P_ + g;
        int hr = refl ? reflect_idx(hi, d.H) : hi;
        const bool hin = (unsigned)hr < (unsigned)d.H;      // (reflected rows of a ragged last row group can still fall outside)
        hr = hin ? hr : 0;
        const int rowbase = (img * d.H + hr) * d.W;
        lds_ptr_t sdst = (lds_ptr_t)smem + stage * HR_SBUF;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int j = wave + 8 * i;
            const int s = 8 * j + l8;
            const int wi = w0 - P_ + s;
            const int wr = refl ? reflect_idx(wi, d.W) : wi;
            const bool win = (unsigned)wr < (unsigned)d.W;
            const bool ok = (s < segw + 6) & hin & win;
            const unsigned off = ok ? (unsigned)((rowbase + wr) * 64 + ((ls ^ ((s >> 1) & 7)) << 3)) * 2u : 0xFFFFFFFFu;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(sdst + j * 1024), 16, (int)off, 0, 0, 0);
        }
    };
    issue_group(0, 0);
    issue_group(1, 1);
    // ---- weights -> LDS once: row (kh, j = kw * Nc + co), chunk c at slot c ^ ((j >> 1) & 7); rows j >= 7 * Nc are zero
    for (int i = tid; i < 7 * 32 * 8; i += 512) {
        const int c = i & 7, j = (i >> 3) & 31, kh = i >> 8;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (j < 7 * Nc) {
            const int kw = j / Nc, co = j - kw * Nc;
            const int wt = flip ? 48 - (kh * 7 + kw) : kh * 7 + kw;
            v = *reinterpret_cast<const u32x4_t*>(wp + (long)co * d.ldw + wt * 64 + c * 8);
        }
        *reinterpret_cast<u32x4_t*>(smem + HR_WOFF + (kh * 32 + j) * 128 + ((c ^ ((j >> 1) & 7)) << 4)) = v;
    }

    // wave w owns pixel tiles w and w + 8 (strip pixels 16 mt .. 16 mt + 15) and, wave 0, tile 16 (pixels 256 .. 261); both N tiles
    const int nmt = wave == 0 ? 3 : 2;
    f32x4_t acc[HR_ROWS][3][2];
#pragma unroll
    for (int r = 0; r < HR_ROWS; ++r)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int a = 0; a < 2; ++a) acc[r][b][a] = f32x4_t{0.f, 0.f, 0.f, 0.f};
    unsigned xa[3], wa[2];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const int s = (wave + 8 * b) * 16 + l16;
        xa[b] = (unsigned)(s * 128 + ((q ^ ((s >> 1) & 7)) << 4));
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const int j = a * 16 + l16;
        wa[a] = (unsigned)(HR_WOFF + j * 128 + ((q ^ ((j >> 1) & 7)) << 4));
    }

#pragma unroll
    for (int g = 0; g < G; ++g) {
        // loads retire in order: with the newest HR_PER DMA instructions (group g+1) outstanding, group g has landed
        if (g + 1 < G) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(HR_PER) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        if (g == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the weight rows written above
        __builtin_amdgcn_s_barrier();                       // everyone's pieces of group g are in LDS; all waves are past group g-1
        if (g + 2 < G) issue_group(g + 2, (g + 2) % HR_NST);
        const unsigned char* sx = smem + (g % HR_NST) * HR_SBUF;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            u32x4_t xf[3];
#pragma unroll
            for (int b = 0; b < 3; ++b)
                if (b < nmt) xf[b] = *reinterpret_cast<const u32x4_t*>(sx + (xa[b] ^ (unsigned)(h << 6)));
#pragma unroll
            for (int r = 0; r < HR_ROWS; ++r) {
                constexpr int dummy = 0; (void)dummy;
                const int kh = g - r;                       // input row ho0-3+g is kernel row kh of output row ho0+r
                if (kh < 0 || kh > 6) continue;             // compile-time after unrolling
                u32x4_t wf[2];
#pragma unroll
                for (int a = 0; a < 2; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(smem + kh * (32 * 128) + (wa[a] ^ (unsigned)(h << 6)));
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    if (b < nmt) {
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            acc[r][b][a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]),
                                                                                   acc[r][b][a], 0, 0, 0);
                    }
            }
        }
    }
    // ---- per output row: P -> LDS as float [272 pixels][32] (over the strip stages), then shift-add, bias, activation, store
    float* P = reinterpret_cast<float*>(smem);
    float bv[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) bv[c] = (bias != nullptr && c < Nc) ? bias[c] : 0.f;
#pragma unroll
    for (int r = 0; r < HR_ROWS; ++r) {
        __syncthreads();                                    // strips (r = 0) / the previous row's P have been consumed
#pragma unroll
        for (int b = 0; b < 3; ++b)
            if (b < nmt) {
                const int s = (wave + 8 * b) * 16 + l16;
#pragma unroll
                for (int a = 0; a < 2; ++a) *reinterpret_cast<f32x4_t*>(P + s * 32 + a * 16 + 4 * q) = acc[r][b][a];
            }
        __syncthreads();
        const int ho = ho0 + r;
        if (tid < segw && w0 + tid < d.Wo && ho < d.Ho) {   // thread t owns output pixel w0 + t: the pixel leaves as one 16-byte store
            float v[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) v[c] = 0.f;
#pragma unroll
            for (int kw = 0; kw < 7; ++kw) {
                const float* pr = P + (tid + kw) * 32 + kw * Nc;
#pragma unroll
                for (int c = 0; c < 4; ++c) if (c < Nc) v[c] += pr[c];          // static register indices: a runtime-bounded loop spills v[]
            }
#pragma unroll
            for (int c = 0; c < 4; ++c) if (c < Nc) v[c] = head_act(v[c] + bv[c], d.act, d.slope);      // act(0) == 0 for the pads
            bf16_t* yp = y + ((long)(img * d.Ho + ho) * d.Wo + w0 + tid) * d.ldc;
            if (d.Nstore == 8) {
                *reinterpret_cast<u32x4_t*>(yp) = f32_to_chunk<bf16_t>(v);
            } else {
#pragma unroll
                for (int c = 0; c < 8; ++c) if (c < d.Nstore) yp[c].v = f32_to_bf16(v[c]);
            }
        }
    }
}

template __global__ void conv_rowstrip_kernel<bf16_t, 7, 8>(const bf16_t*, const bf16_t*, const float*, bf16_t*, const RowStripDesc);
template __global__ void conv_rowstrip_kernel<float, 7, 8>(const float*, const float*, const float*, float*, const RowStripDesc);

static int g_rowstrip_mode = 1;
extern "C" void uig_debug_set_rowstrip(int on) { g_rowstrip_mode = on; }

template <typename T>
static int launch_rowstrip(const void* x, const void* wp, const float* bias, void* y, const RowStripDesc& d, hipStream_t s) {
    constexpr int SROWS = (256 + 7 - 1 + 7) / 8 * 8;
    const size_t smem = 2 * (size_t)(SROWS * 128 + 7 * 16 * 128);
    auto kern = conv_rowstrip_kernel<T, 7, 8>;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(kern), (size_t)(int)smem);
        if (e != hipSuccess) return uig_set_error((int)e, "conv_rowstrip: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(d.B * d.Ho * (d.Wo / 256)), dim3(512), smem, s, (const T*)x, (const T*)wp, bias, (T*)y, d);
    UIG_LAUNCH_CHECK("uig_conv_gather(rowstrip)");
    return 0;
}

// Returns 1 if this kernel took the launch, 0 if the shape does not qualify (caller falls back).
int uig_try_conv_rowstrip(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                          void* y, int B, int H, int W, int Cin, int Nrows, int k,
                          int pad_mode, const int* taps, int ntaps, int Ho, int Wo, int ldc, int Nstore, int act, float slope,
                          int dtype, long x_bytes, long w_bytes, hipStream_t s, int* rc_out) {
    const int BKe = dtype == UIG_BF16 ? 64 : 32;
    if (!g_rowstrip_mode || k != 7 || ntaps != 49 || Nrows > 16 || Cin % BKe != 0) return 0;
    // <= 4 output channels, bf16, Cin == 64: taps-on-N kernel.  Natural taps (direct conv, pad P: dh = kh - P, dw = kw - P) or the
    // mirrored order of a transposed gather (dh = Pt - kh, dw = Pt - kw: kernel index 6 - kh with P = 6 - Pt): the 7x7 head forward
    // (P = 3) and the padded input gradient of the 7x7 stem (Pt = 0 -> P = 6, output 6 pixels larger than the input).
    if (dtype == UIG_BF16 && g_rowstrip_mode == 1 && Nrows <= 4 && Nstore <= 8 && (ldc & 7) == 0 && Cin == 64 && (Wo % 256 == 0 || Wo <= 266)) {
        const int dh0 = (taps[0] & 255) - 128, dw0 = ((taps[0] >> 8) & 255) - 128;
        const int flip = dh0 > 0 || (dh0 == 0 && ((taps[48] & 255) - 128) < 0) ? 1 : 0;
        const int P = flip ? 6 - dh0 : -dh0;
        bool ok = dh0 == dw0 && P >= 0 && P <= 6 && (pad_mode == UIG_PAD_ZERO || (P <= 3 && H > 3 && W > 3));
        for (int i = 0; i < ntaps && ok; ++i) {
            const int kh = flip ? 6 - i / 7 : i / 7, kw = flip ? 6 - i % 7 : i % 7;
            ok = (taps[i] >> 16) == i && ((taps[i] & 255) - 128) == kh - P && (((taps[i] >> 8) & 255) - 128) == kw - P;
        }
        if (ok) {
            RowStripDesc d{};
            d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Ho = Ho; d.Wo = Wo; d.k = k; d.pad_mode = pad_mode;
            d.Nrows = Nrows; d.ldw = ntaps * Cin; d.ldc = ldc; d.Nstore = Nstore; d.act = act; d.slope = slope;
            d.x_bytes = (unsigned)x_bytes; d.w_bytes = (unsigned)w_bytes;
            d.wp2 = wp2; d.bias2 = bias2; d.group_images = group_images;
            static SmemAttrOnce attr_once2;
            {
                hipError_t e = attr_once2.ensure(reinterpret_cast<const void*>(conv_headrow_kernel), (size_t)HR_SMEM);
                if (e != hipSuccess) { *rc_out = uig_set_error((int)e, "conv_headrow: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 1; }
            }
            uig_note_conv_kernel(UIG_K_HEADROW);
            const int segw = Wo % 256 == 0 ? 256 : Wo;
            hipLaunchKernelGGL(conv_headrow_kernel, dim3(B * ((Ho + HR_ROWS - 1) / HR_ROWS) * ((Wo + segw - 1) / segw)), dim3(512), HR_SMEM, s,
                               (const bf16_t*)x, (const bf16_t*)wp, bias, (bf16_t*)y, d, P, flip, segw);
            hipError_t e = hipGetLastError();
            *rc_out = e == hipSuccess ? 0 : uig_set_error((int)e, "conv_headrow: launch failed: %s", hipGetErrorString(e));
            return 1;
        }
    }
    if (Wo % 256 != 0 || Ho != H || Wo != W) return 0;
    RowStripDesc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Ho = Ho; d.Wo = Wo; d.k = k; d.pad_mode = pad_mode;
    d.Nrows = Nrows; d.ldw = ntaps * Cin; d.ldc = ldc; d.Nstore = Nstore; d.act = act; d.slope = slope;
    d.x_bytes = (unsigned)x_bytes; d.w_bytes = (unsigned)w_bytes;
    d.wp2 = wp2; d.bias2 = bias2; d.group_images = group_images;
    int R = 0;
    for (int i = 0; i < ntaps; ++i) {
        d.tap[i] = taps[i];
        R = std::max(R, std::abs(((taps[i] >> 8) & 255) - 128));
        if (((taps[i] & 255) - 128) != ((taps[(i / k) * k] & 255) - 128)) return 0;      // taps must be kernel-row major
    }
    if (R > (k - 1)) return 0;
    d.R = R;
    if (2 * R + 256 > (256 + 7 - 1 + 7) / 8 * 8) return 0;
    uig_note_conv_kernel(UIG_K_ROWSTRIP);
    *rc_out = dtype == UIG_BF16 ? launch_rowstrip<bf16_t>(x, wp, bias, y, d, s) : launch_rowstrip<float>(x, wp, bias, y, d, s);
    return 1;
}
