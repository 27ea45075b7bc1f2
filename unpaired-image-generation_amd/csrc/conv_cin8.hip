// conv_cin8.hip — stride-1 k x k convolution (k <= 7) whose INPUT has one 16-byte chunk of channels per pixel (8 bf16: the
// 3-channel image padded to 8) and up to 64 output channels: the generator's 7x7 stem forward (3 -> 64) and the input
// gradient of its 7x7 head (3 -> 64 through the transposed taps), bf16.
//
// On the generic gather kernel these layers take the "small Cin" path: every 128-byte K-row of the im2col tile is 8 taps x 8
// channels gathered chunk by chunk, with a per-lane tap lookup and reflection arithmetic per chunk and K-step - 170 us for the
// batch-16 stem (109 TFLOP/s) where the output write alone allows ~35 us.  Here the whole weight tensor (64 x 49 x 16 B = 50 KB)
// is LDS-resident for the life of a block, laid out [n][tap row r][slot s = 0..7][8 ch] (slot = horizontal tap position,
// the 8th slot is zero), and a block walks RPB consecutive output rows of a 128-pixel segment keeping the k input rows it
// needs in an LDS ring (one new 2-KB row per output row).  One MFMA K-group of 32 = 4 horizontal taps x 8 channels: the
// B fragment of output pixel j is the 16-byte chunk of input pixel j + s - a plain shifted read of the ring row, and the A
// fragment is chunk (r, s) of the weight row.  14 MFMA K-groups per output pixel tile, no im2col, no per-chunk address math.
#include "uig_common.h"
#include <algorithm>

struct Cin8Desc {
    int B, H, W, Ho, Wo;
    int pad_mode, dh_min, dw_min, KR;     // input pixel of slot (r, s) for output (ho, wo): (ho + dh_min + r, wo + dw_min + s)
    int Nrows, ldw, ldc, Nstore, act;
    float slope;
    int rpb, group_images, segw;           // segw: output pixels per segment (multiple of 32, <= 128): balanced over the row
    unsigned w_bytes;
    int wmap[56];                           // weight tap index of slot r * 8 + s, or -1 (zero weights)
    const void* wp2; const float* bias2;
};

namespace {
constexpr int C8_BM = 128;                  // output pixels per segment: 4 waves x 32
constexpr int C8_RINGW = (C8_BM + 8) * 16;  // bytes per ring row: pixels wo0 + dw_min + [0, BM + 8)
constexpr int C8_RING = 8;                  // ring slots (>= KR + 1)
constexpr int C8_WROW = 7 * 128;            // weight row stride (bytes): 7 tap rows x 8 slots x 16 B; slot s of row n sits at s ^ ((n >> 1) & 7)
                                            // (unpadded rows are 896 B = 128 mod 256: the XOR makes the 16 chunks of a ds_read_b128 lane group distinct)
}

template <int KRT>      // tap rows (compile-time: the 14 K-groups of a 7x7 stencil are fully unrolled so that fragment reads run ahead of the MFMAs)
__global__ __launch_bounds__(256, 2) void conv_cin8_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ wp_,
                                                            const float* __restrict__ bias_, bf16_t* __restrict__ y, const Cin8Desc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];     // [weights 64 x WROW][ring 8 x RINGW][scratch 4 x 1 KB]  = 78848 B: two blocks per CU
    unsigned char* sw = smem;
    unsigned char* sr = smem + 64 * C8_WROW;
    unsigned char* sc = sr + C8_RING * C8_RINGW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l16 = lane & 15, q = lane >> 4;
    const int nseg = (d.Wo + d.segw - 1) / d.segw, nrg = (d.Ho + d.rpb - 1) / d.rpb;
    int bid = blockIdx.x;
    const int seg = bid % nseg; bid /= nseg;
    const int rg = bid % nrg; const int img = bid / nrg;
    const int wo0 = seg * d.segw, ho0 = rg * d.rpb, ho1 = min(d.Ho, ho0 + d.rpb);
    const bool g2 = d.wp2 != nullptr && img >= d.group_images;
    const bf16_t* wp = g2 ? static_cast<const bf16_t*>(d.wp2) : wp_;
    const float* bias = g2 ? d.bias2 : bias_;
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;

    // ---- weights -> LDS (once per block): chunk (n, r, s) = wp[n][wmap[r*8+s]][0..7] or zeros
    for (int i = tid; i < 64 * d.KR * 8; i += 256) {
        const int s = i & 7, r = (i >> 3) % d.KR, n = i / (8 * d.KR);
        const int wt = d.wmap[r * 8 + s];
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (wt >= 0 && n < d.Nrows) v = *reinterpret_cast<const u32x4_t*>(wp + (long)n * d.ldw + wt * 8);
        *reinterpret_cast<u32x4_t*>(sw + n * C8_WROW + r * 128 + ((s ^ ((n >> 1) & 7)) << 4)) = v;
    }
    // ---- input rows -> ring: row hi (any integer) goes to slot (hi - (ho0 + dh_min)) % 8; thread t < 136 loads pixel t of the row
    const int wi_t = wo0 + d.dw_min + tid;                      // this thread's pixel column of a ring row
    bool colok = tid < C8_BM + 8;
    int wr = wi_t;
    if (refl) wr = reflect_idx(wi_t, d.W); else colok = colok && (unsigned)wi_t < (unsigned)d.W;
    if (refl && ((unsigned)wr >= (unsigned)d.W)) colok = false;     // (only for columns far outside: never read by a valid pixel)
    auto load_row = [&](int k) -> u32x4_t {                    // ring row k = input row ho0 + dh_min + k
        int hi = ho0 + d.dh_min + k;
        bool ok = colok;
        if (refl) { hi = reflect_idx(hi, d.H); ok = ok && (unsigned)hi < (unsigned)d.H; }
        else ok = ok && (unsigned)hi < (unsigned)d.H;
        u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
        if (ok) v = *reinterpret_cast<const u32x4_t*>(x + (((long)img * d.H + hi) * d.W + wr) * 8);
        return v;
    };
    auto store_row = [&](int k, const u32x4_t& v) {
        if (tid < C8_BM + 8) *reinterpret_cast<u32x4_t*>(sr + (k & (C8_RING - 1)) * C8_RINGW + tid * 16) = v;
    };
    for (int k = 0; k < d.KR - 1; ++k) store_row(k, load_row(k));

    // ---- per-lane fragment offsets
    unsigned aoff[4][2];                                        // weight chunk (n = a*16 + l16, r = 0, s = 4h + q), swizzled
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const int n = a * 16 + l16, g = (n >> 1) & 7;
#pragma unroll
        for (int h = 0; h < 2; ++h) aoff[a][h] = (unsigned)(n * C8_WROW + (((4 * h + q) ^ g) << 4));
    }
    unsigned boff[2];                                           // ring chunk of pixel (wave*32 + b*16 + l16) + s, s = q
#pragma unroll
    for (int b = 0; b < 2; ++b) boff[b] = (unsigned)((wave * 32 + b * 16 + l16 + q) * 16);
    float bv[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int e = 0; e < 4; ++e) { const int n = a * 16 + 4 * q + e; bv[a * 4 + e] = (bias != nullptr && n < d.Nrows) ? bias[n] : 0.f; }
    const int nvalid = min(d.segw, d.Wo - wo0);                 // valid pixels of this segment
    const bool wave_live = wave * 32 < nvalid;                  // waves past the end of a ragged segment only help with loads / barriers

    for (int ho = ho0; ho < ho1; ++ho) {
        const int k0 = ho - ho0;                                // ring rows k0 .. k0 + KR - 1
        // the one new input row of this output row: its slot held row k0 + KR - 9, last read two output rows ago, and the barrier of
        // the previous row guarantees nobody is further back than one row - so one barrier per output row is enough
        store_row(k0 + d.KR - 1, load_row(k0 + d.KR - 1));
        __syncthreads();
        if (wave_live) {
            f32x4_t acc[4][2];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int r = 0; r < KRT; ++r) {
                const unsigned char* rowp = sr + ((k0 + r) & (C8_RING - 1)) * C8_RINGW;
                const unsigned char* wrp = sw + r * 128;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4_t wf[4], xf[2];
#pragma unroll
                    for (int a = 0; a < 4; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(wrp + aoff[a][h]);
#pragma unroll
                    for (int b = 0; b < 2; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(rowp + boff[b] + h * 64);
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, wf[a]), __builtin_bit_cast(bf16x8_t, xf[b]), acc[a][b], 0, 0, 0);
                }
            }
            // ---- epilogue: bias + activation; the 32 x 64 tile goes through a wave-private 1-KB LDS scratch 8 pixel rows at a time and
            //      leaves as whole 128-byte pixel rows (16 B per lane)
            unsigned char* scr = sc + wave * 1024;
            const int c = lane & 7, r0 = lane >> 3;
            bf16_t* yrow = y + (((long)img * d.Ho + ho) * d.Wo + wo0 + wave * 32) * d.ldc;
            with_act(d.act, d.slope, [&](auto actf) {       // one activation branch per output row, not one switch per element
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int b = ps >> 1;
                if ((l16 >> 3) == (ps & 1)) {                       // lanes whose pixel row lies in this pass
                    const int m = l16 & 7;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        float v[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = actf(acc[a][b][e] + bv[a * 4 + e]);
                        u32x2_t pk;
                        pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                        pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                        *reinterpret_cast<u32x2_t*>(scr + m * 128 + (((2 * a + (q >> 1)) ^ m) << 4) + (q & 1) * 8) = pk;
                    }
                }
                // compiler + LDS ordering between the 8-byte writes and the 16-byte reads of the scratch (different vector types: without
                // the barrier the reads of a pass were scheduled above its writes - stale rows of the previous pass)
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const int r = 8 * ps + r0;                          // lane stores chunk c of pixel row r
                const u32x4_t val = *reinterpret_cast<const u32x4_t*>(scr + r0 * 128 + ((c ^ r0) << 4));
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                if (wave * 32 + r < nvalid) {
                    if (d.Nstore == 64 && (d.ldc & 7) == 0) {
                        *reinterpret_cast<u32x4_t*>(yrow + (long)r * d.ldc + c * 8) = val;
                    } else {
                        const unsigned short* hv = reinterpret_cast<const unsigned short*>(&val);
#pragma unroll
                        for (int e = 0; e < 8; ++e)
                            if (c * 8 + e < d.Nstore) yrow[(long)r * d.ldc + c * 8 + e].v = hv[e];
                    }
                }
            }
            });
        }
    }
}

static int g_cin8_mode = 1;      // A/B and parity hook
extern "C" void uig_debug_set_cin8(int on) { g_cin8_mode = on; }

// Returns 1 if this kernel took the launch (rc in *rc_out), 0 if the shape does not qualify (caller falls back).
int uig_try_conv_cin8(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                      void* y, int B, int H, int W, int Cin, int Nrows, int pad_mode, const int* taps, int ntaps,
                      int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype, long w_bytes, hipStream_t s, int* rc_out) {
    if (!g_cin8_mode || dtype != UIG_BF16 || Cin != 8 || Nrows > 64 || Nstore > 64 || ntaps < 9 || ntaps > 49) return 0;
    Cin8Desc d{};
    int dh0 = 127, dh1 = -127, dw0 = 127, dw1 = -127;
    for (int i = 0; i < ntaps; ++i) {
        const int dh = (taps[i] & 255) - 128, dw = ((taps[i] >> 8) & 255) - 128;
        dh0 = std::min(dh0, dh); dh1 = std::max(dh1, dh); dw0 = std::min(dw0, dw); dw1 = std::max(dw1, dw);
    }
    const int KR = dh1 - dh0 + 1, KS = dw1 - dw0 + 1;
    if (KR > 7 || KS > 7 || KR * KS != ntaps) return 0;
    for (int i = 0; i < 56; ++i) d.wmap[i] = -1;
    for (int i = 0; i < ntaps; ++i) {
        const int dh = (taps[i] & 255) - 128, dw = ((taps[i] >> 8) & 255) - 128, slot = (dh - dh0) * 8 + (dw - dw0);
        if (d.wmap[slot] != -1) return 0;                      // duplicate offsets: not a plain k x k stencil
        d.wmap[slot] = taps[i] >> 16;
    }
    if (pad_mode == UIG_PAD_REFLECT && (KR > H || KS > W)) return 0;
    d.B = B; d.H = H; d.W = W; d.Ho = Ho; d.Wo = Wo; d.pad_mode = pad_mode; d.dh_min = dh0; d.dw_min = dw0; d.KR = KR;
    d.Nrows = Nrows; d.ldw = ntaps * Cin; d.ldc = ldc; d.Nstore = Nstore; d.act = act; d.slope = slope;
    d.rpb = 1;      // set below
    d.group_images = group_images; d.wp2 = wp2; d.bias2 = bias2; d.w_bytes = (unsigned)w_bytes;
    int nseg = (Wo + C8_BM - 1) / C8_BM;
    d.segw = std::min(C8_BM, ((Wo + nseg - 1) / nseg + 31) / 32 * 32);      // e.g. Wo = 262: 3 segments of 96 instead of 128 + 128 + 6
    nseg = (Wo + d.segw - 1) / d.segw;
    // (rows that leave > 10 % of the 4-wave segments idle, e.g. the 262-wide padded gradient of the head, used to stay on the generic
    // kernel: 109 vs 102 us in round 1.  Since the epilogue lost its per-element activation switch this kernel wins there too:
    // 163 vs 190 us at 16 images, 77 vs 91 at 8, fold excluded.)
    // output rows per block: whole rounds of 512 resident blocks (2 per CU), fewest row-steps per CU; ties go to more rows per
    // block (the 50-KB weight image is loaded once per block)
    {
        long best = -1;
        for (int r = 4; r <= 32; ++r) {
            if (r > Ho && r != 4) break;
            const long blocks = (long)B * ((Ho + r - 1) / r) * nseg, cost = ((blocks + 511) / 512) * r;
            if (best < 0 || cost <= best) { best = cost; d.rpb = r; }
        }
        d.rpb = std::min(d.rpb, std::max(1, Ho));
    }
    const int nrg = (Ho + d.rpb - 1) / d.rpb;
    const size_t smem = 64 * C8_WROW + C8_RING * C8_RINGW + 4 * 1024;
    if (KR != 7) return 0;                                     // only the 7-row stencil is instantiated
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(conv_cin8_kernel<7>), (size_t)(int)smem);
        if (e != hipSuccess) { *rc_out = uig_set_error((int)e, "conv_cin8: hipFuncSetAttribute failed: %s", hipGetErrorString(e)); return 1; }
    }
    uig_note_conv_kernel(UIG_K_CIN8);
    hipLaunchKernelGGL(conv_cin8_kernel<7>, dim3(B * nrg * nseg), dim3(256), smem, s, (const bf16_t*)x, (const bf16_t*)wp, bias, (bf16_t*)y, d);
    hipError_t e_ = hipGetLastError();
    *rc_out = e_ == hipSuccess ? 0 : uig_set_error((int)e_, "uig_conv_gather(cin8): launch failed: %s", hipGetErrorString(e_));
    return 1;
}
