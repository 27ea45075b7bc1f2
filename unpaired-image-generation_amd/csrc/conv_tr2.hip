// conv_tr2.hip — transposed gather with stride 2, 3x3, pad 1 (ConvTranspose2d(k3, s2, p1, output_padding 1) forward and the
// input gradient of Conv2d(k3, s2, p1)), bf16: all four output phases of a tile in ONE block.
//
// On the generic gather kernel each output phase (py, px) is its own GEMM with 1, 2, 2 or 4 taps: at 128 -> 64 channels
// that is 8192 tiles of 128 x 64 with 2..8 K-steps each - prologue / epilogue dominated (245 TFLOP/s, 158 us for the
// batch-16 up-sampling layer whose output write alone allows ~35 us), the input is staged once per phase and tap, and
// the output leaves as 128-byte pixels 256 bytes apart.  Here a block owns 128 input pixels (whole rows) of one image:
// per 64-channel chunk the (TI+1) input rows it needs sit in LDS once (a strip, as in conv_strip.hip: the +1 column of the
// taps falls off the right edge of a whole row, so the column halo is a zero row), the nine 64 x 64 weight tiles of the
// chunk sit beside it, and every wave owns one phase of 64 input pixels: its taps are SHIFTED fragment reads of the strip,
// 32 MFMAs per tap and chunk with no barrier in between.  Waves are paired on the SIMDs so that the 1/2/2/4-tap phases
// balance (5,5,4,4 tap units).  The four phases of a pixel row leave as whole output rows through the LDS transposition.
#include "uig_common.h"
#include <algorithm>

struct Tr2Desc {
    int B, H, W, Cin;             // input (B, H, W, Cin); output (B, 2H, 2W, ldc)
    int Nrows, ldw, ldc, Nstore, act;
    float slope;
    unsigned x_bytes, w_bytes;
    int TI;                       // input rows per tile: TI * W == 128
    int ph_tap0[5];
    int tap[9];                   // dh | dw << 8 | weight tap index << 16, grouped by phase (py * 2 + px); dh, dw in {0, 1}
    const void* wp2; const float* bias2; int group_images;
    float* in_partial;            // optional InstanceNorm partial statistics [img][Ho*Wo/64][Nstore][2]
};

namespace {
constexpr int TR_PIX = 128;                          // input pixels per tile
constexpr int TR_KC = 32;                            // channels per K-chunk = one MFMA K-step; LDS rows are 64 B
constexpr int TR_ROWB = TR_KC * 2;
constexpr int TR_SROWS = 2 * TR_PIX + 16;            // strip rows: (TI+1) * W <= 256, + the zero rows
constexpr int TR_SBUF = TR_SROWS * TR_ROWB;          // 17408 B per strip buffer
constexpr int TR_WBUF = 9 * 64 * TR_ROWB;            // 36864 B: nine 64 x 32 weight tiles of one chunk
constexpr int TR_SMEM = 2 * TR_SBUF + TR_WBUF;       // 71680 B: TWO blocks per CU cover each other's load latency (a 64-channel
                                                     // chunk would need 138 KB: one block per CU, every load exposed: 97 us vs 158)
}

__global__ __launch_bounds__(512, 2) void conv_tr2_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp_,
                                                           const float* __restrict__ bias_, bf16_t* __restrict__ y, const Tr2Desc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // [strip 0][strip 1][weights / epilogue scratch]
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l16 = lane & 15, q = lane >> 4;

    // ---- tile coordinates (XCD-aware: blocks that share an XCD get a contiguous run of tiles -> shared halo rows hit its L2)
    int bid;
    {
        const int nwg = gridDim.x, o = blockIdx.x, xcd = o & 7, qq = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (o >> 3);
    }
    const int ntn = d.Nrows / 64, ntiles = d.H / d.TI;
    const int n_base = (bid % ntn) * 64; bid /= ntn;
    const int ti = bid % ntiles, img = bid / ntiles;
    const int i0 = ti * d.TI, W = d.W, Cin = d.Cin;
    const bool g2 = d.wp2 != nullptr && img >= d.group_images;
    const bf16_t* wp = g2 ? static_cast<const bf16_t*>(d.wp2) : wp_;
    const float* bias = g2 ? d.bias2 : bias_;
    const int nrows_s = (d.TI + 1) * W;              // strip rows holding pixels; rows nrows_s .. +15 are zero
    const int npiece_s = nrows_s / 16;               // 1-KiB DMA pieces per strip chunk (16 or 12)

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(wp), 0, d.w_bytes, 0x00020000);

    // the zero rows of both strip buffers (never touched by the DMA)
    if (tid < 128) {
        unsigned char* z = smem + (tid >> 6) * TR_SBUF + nrows_s * TR_ROWB + (tid & 63) * 16;
        *reinterpret_cast<u32x4_t*>(z) = u32x4_t{0u, 0u, 0u, 0u};
    }

    // LDS-DMA: one wave instruction writes 1 KiB = 16 rows x 64 B, lane L -> row L/4, slot L%4; slot s of row r holds the
    // global 16-byte chunk s ^ ((r >> 2) & 3) (source-side swizzle: the 16 rows of a ds_read_b128 lane group hit 16 distinct
    // 16-byte bank groups)
    const int lr8 = lane >> 2, sl = lane & 3;
    auto issue_strip = [&](int chunk, int buf) {
        for (int pc = wave; pc < npiece_s; pc += 8) {
            const int p = pc * 16 + lr8;                              // strip row = pixel i0 * W + p of the image
            const int gi = i0 + p / W;
            const unsigned off = gi < d.H ? (unsigned)(((img * d.H + i0) * W + p) * Cin + chunk * TR_KC + ((sl ^ ((p >> 2) & 3)) << 3)) * 2u
                                          : 0xFFFFFFFFu;              // rows below the image: the hardware returns zeros
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)((lds_ptr_t)smem + buf * TR_SBUF + pc * 1024),
                                                     16, (int)off, 0, 0, 0);
        }
    };
    auto issue_weights = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int pc = wave + 8 * k;                              // 36 pieces: tap slot pc / 4, rows (pc % 4) * 16 ..
            if (pc >= 36) break;
            const int ts = pc >> 2, n = (pc & 3) * 16 + lr8;
            const int widx = __builtin_amdgcn_readfirstlane(d.tap[ts] >> 16);
            const unsigned off = (unsigned)(((n_base + n) * d.ldw) + widx * Cin + chunk * TR_KC + ((sl ^ ((n >> 2) & 3)) << 3)) * 2u;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)((lds_ptr_t)smem + 2 * TR_SBUF + pc * 1024),
                                                     16, (int)off, 0, 0, 0);
        }
    };

    // ---- this wave's role: phase (py, px) and which 64 of the 128 input pixels
    // SIMD s hosts waves s and s + 4: (w0 3, w4 0) (w1 3, w5 0) (w2 1, w6 1) (w3 2, w7 2) -> 5, 5, 4, 4 taps per SIMD
    const int phase = __builtin_amdgcn_readfirstlane((0x21002133u >> (4 * wave)) & 3);
    const int half = __builtin_amdgcn_readfirstlane((0xE2u >> wave) & 1);                    // w1, w5, w6, w7 take pixels 64..127
    const int tap0 = d.ph_tap0[phase], ntap = d.ph_tap0[phase + 1] - tap0;
    const int py = phase >> 1, px = phase & 1;

    // strip row of this lane's pixel for the four 16-pixel M-tiles, unshifted, and whether dw = 1 falls off the row
    int prow[4]; bool edge[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int m = half * 64 + b * 16 + l16;
        prow[b] = m;                                   // tile pixel m = (m / W) * W + m % W: rows are contiguous in the strip
        edge[b] = (m % W) == W - 1;
    }

    f32x4_t acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int nchunk = Cin / TR_KC;
    issue_strip(0, 0);
    issue_weights(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int c = 0; c < nchunk; ++c) {
        if (c + 1 < nchunk) issue_strip(c + 1, (c + 1) & 1);          // lands behind this chunk's MFMAs
        const unsigned char* sb = smem + (c & 1) * TR_SBUF;
        const unsigned char* wb = smem + 2 * TR_SBUF;
        for (int t = 0; t < ntap; ++t) {
            const int te = __builtin_amdgcn_readfirstlane(d.tap[tap0 + t]);
            const int dh = te & 255, dw = (te >> 8) & 255;
            const unsigned char* wt = wb + (tap0 + t) * (64 * TR_ROWB);
            u32x4_t wf[4], xf[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const int n = a * 16 + l16;
                wf[a] = *reinterpret_cast<const u32x4_t*>(wt + n * TR_ROWB + ((q ^ ((n >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int b = 0; b < 4; ++b) {
                const int r = (dw && edge[b]) ? nrows_s : prow[b] + dh * W + dw;
                xf[b] = *reinterpret_cast<const u32x4_t*>(sb + r * TR_ROWB + ((q ^ ((r >> 2) & 3)) << 4));
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b)
                    acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]),
                                                                        acc[a][b], 0, 0, 0);
        }
        __syncthreads();                                              // every wave is done with this chunk's weights and strip
        if (c + 1 < nchunk) {
            issue_weights(c + 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }

    // ---- epilogue: 64 pixels x 64 channels of one phase per wave, through a wave-private 8-KB scratch (the whole block's LDS:
    //      the barrier that ended the last chunk guarantees nobody reads strips or weights any more)
    float b4[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int n = n_base + a * 16 + 4 * q + e;
            b4[a * 4 + e] = (bias != nullptr && n < d.Nrows) ? bias[n] : 0.f;
        }
    const int Ho = 2 * d.H, Wo = 2 * W;
    float* so = nullptr;
    if (d.in_partial != nullptr)
        so = d.in_partial + ((((long)img * (Ho * Wo / 64)) + (ti * 4 + phase) * 2 + half) * d.Nstore + n_base) * 2;
    store_tile_via_lds<bf16_t, 4, 4>(acc, smem + wave * 8192, lane, b4, d.act, d.slope,
                                     [&](int r) -> bf16_t* {
                                         const int m = half * 64 + r;
                                         const int oy = 2 * (i0 + m / W) + py, ox = 2 * (m % W) + px;
                                         return y + (((long)img * Ho + oy) * Wo + ox) * d.ldc + n_base;
                                     },
                                     so, 64);
}

static int g_tr2_mode = 1;
extern "C" void uig_debug_set_tr2(int on) { g_tr2_mode = on; }

// shape rule of this kernel (also exported: the host uses it to decide whether a 64-channel transposed layer can emit the
// following InstanceNorm's statistics from its epilogue)
extern "C" int uig_conv_tr2_applicable(int B, int H, int W, int Cin, int Nrows, int Nstore, int ldc, int dtype) {
    if (!g_tr2_mode || dtype != UIG_BF16) return 0;
    if ((W != 64 && W != 128) || Cin % TR_KC != 0 || Cin < 64) return 0;
    if (Nrows % 64 != 0 || Nstore != Nrows || (ldc * 2) % 16 != 0 || ldc < Nstore) return 0;
    const int TI = TR_PIX / W;
    if (H % TI != 0 || (long)(TI + 1) * W > 2 * TR_PIX) return 0;
    if ((long)B * (H / TI) * (Nrows / 64) > 0x7fffffffL) return 0;
    return 1;
}

// returns 1 if the launch was handled here (*rc_out = status), 0 if the shape is not this kernel's
int uig_try_conv_tr2(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                     float* in_partial, void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                     const int* ph_tap0, const int* taps, int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype,
                     long x_bytes, long w_bytes, hipStream_t s, int* rc_out) {
    if (kH != 3 || kW != 3 || stride != 2 || pad != 1 || Ho != 2 * H || Wo != 2 * W) return 0;
    if (!uig_conv_tr2_applicable(B, H, W, Cin, Nrows, Nstore, ldc, dtype)) return 0;
    const int TI = TR_PIX / W;
    Tr2Desc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Nrows = Nrows; d.ldw = 9 * Cin; d.ldc = ldc; d.Nstore = Nstore; d.act = act; d.slope = slope;
    d.x_bytes = (unsigned)x_bytes; d.w_bytes = (unsigned)w_bytes; d.TI = TI;
    for (int p = 0; p <= 4; ++p) d.ph_tap0[p] = ph_tap0[p];
    if (d.ph_tap0[4] != 9) return 0;
    for (int t = 0; t < 9; ++t) {
        const int dh = (taps[t] & 255) - 128, dw = ((taps[t] >> 8) & 255) - 128;
        if (dh < 0 || dh > 1 || dw < 0 || dw > 1) return 0;
        d.tap[t] = dh | (dw << 8) | ((taps[t] >> 16) << 16);
    }
    // the wave roles assume 1 / 2 / 2 / 4 taps for phases 0..3
    if (d.ph_tap0[1] - d.ph_tap0[0] != 1 || d.ph_tap0[2] - d.ph_tap0[1] != 2 || d.ph_tap0[3] - d.ph_tap0[2] != 2) return 0;
    d.wp2 = wp2; d.bias2 = bias2; d.group_images = group_images; d.in_partial = in_partial;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(conv_tr2_kernel), (size_t)TR_SMEM);
        if (e != hipSuccess) { *rc_out = uig_set_error((int)e, "conv_tr2: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 1; }
    }
    uig_note_conv_kernel(UIG_K_TR2);
    hipLaunchKernelGGL(conv_tr2_kernel, dim3(B * (H / TI) * (Nrows / 64)), dim3(512), TR_SMEM, s,
                       (const bf16_t*)x, (const bf16_t*)wp, bias, (bf16_t*)y, d);
    hipError_t e = hipGetLastError();
    *rc_out = e == hipSuccess ? 0 : uig_set_error((int)e, "conv_tr2: launch failed: %s", hipGetErrorString(e));
    return 1;
}
