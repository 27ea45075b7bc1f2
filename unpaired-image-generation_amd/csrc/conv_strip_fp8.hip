// conv_strip_fp8.hip — the ResBlock 3x3 convolution (forward and input gradient) on the CDNA4 block-scaled fp8 matrix
// instruction v_mfma_scale_f32_16x16x128_f8f6f4: OCP e4m3 operands with one E8M0 scale per 32 K elements (MX format),
// fp32 accumulate, bf16 output.  BASELINE.json configs[4]; MX-scaled fp8 MFMA runs at twice the bf16 rate
// (MI355X_MICROARCH.md: Matrix cores).
//
// Same structure as conv_strip_pk.hip (persistent blocks walking 256-pixel x 128-channel tiles, LDS-resident input strip,
// taps as shifted fragment reads, next chunk / next tile streamed in behind the MFMAs), with these differences:
//   * a 128-byte LDS row is 128 fp8 channels: one K-step consumes 128 channels of one tap (K = 128 per MFMA), a 256-channel
//     layer has 2 chunks instead of 4, half the strip and weight bytes per FLOP;
//   * operand layout of the instruction (found with scripts/probes/probe_mx_*.hip, exact small-integer data): lane l holds row /
//     column l & 15; its operand registers 0-3 are K = 16g .. 16g+15 and registers 4-7 are K = 64+16g .. 64+16g+15 (g = l >> 4) -
//     the 16-byte chunks g and g+4 of the row, i.e. exactly the two fragment reads of the bf16 kernel; byte OPSEL of the
//     lane's scale register is the E8M0 scale of K block g (K = 32g .. 32g+31) of its row;
//   * scales travel with their data: the strip buffer is followed by one dword per strip row (the 4 block scales of that
//     row's 128-channel chunk) and the weight stage by one dword per weight row, DMA'd together with them; a lane reads the
//     dword of its row and shifts its block's byte down (the zero row's scales are 1.0: E8M0 0xff would be NaN).
// Quantisation (uig_mx_quantize, below): per 32-channel block, scale = 2^(floor(log2(amax)) - 8) (e4m3: emax = 8), elements
// = RNE(x / scale) clamped to +-448 (the conversion instruction returns NaN above 464).  The test suite holds a CPU emulation with the same rounding, byte for byte.
#include "conv_strip_desc.h"
#include <algorithm>

typedef __attribute__((ext_vector_type(8))) int i32x8_t;

struct MxDesc {
    StripDesc d;                        // geometry / epilogue fields as for the bf16 kernels (x_bytes, w_bytes = fp8 bytes)
    const unsigned char* xs;            // x scales  [B*H*W][Cin/32]
    const unsigned char* ws;            // w scales  [Nrows][9][Cin/32]
    const unsigned char* ws2;           // second network's (paired launch)
    unsigned xs_bytes, ws_bytes;
};

// MIRROR (round 3): the input gradient of a REFLECTION-padded convolution in one launch, as in conv_strip_pk.hip - the taps that would
// read a mirrored line / column read "mirror pixels" placed behind the tile's NS strip rows (same slot layout: NS + 8r + 2 / 5 column
// mirrors, NS + 48 + c line mirror, + 66 / 69 its corners).  In fp8 a mirror pixel is the sum of two (four) pixels with DIFFERENT block
// scales: each source chunk is de-quantised (e4m3 -> f32, times its block's E8M0 scale), the sum is taken in fp32 and re-quantised with
// the MX rule (one new scale per 32 channels = two adjacent lanes' chunks: one xor-shuffle), data and scale byte written to the slot.
// The scale dwords of the next chunk's strip are DMA'd in the chunk's FIRST step instead of its last, so that they have landed when
// the mirror pixels are built at the top of the last step.  No border GEMM in front, no border loads in the epilogue.
// STAMP: diagnostic build (scripts/stamp_fp8.py): per wave, s_memtime sums {total, step wait + barrier, DMA issue (+ mirror pixels), fragment
// reads + MFMAs, epilogue, row table, tiles} to d.dbg.
template <int CAP, bool MIRROR = false, bool STAMP = false, int NISS = 4>
__global__ __launch_bounds__(512, 2)
void conv_strip_fp8_kernel(const unsigned char* __restrict__ x, const unsigned char* __restrict__ wp1, const float* __restrict__ bias1,
                           bf16_t* __restrict__ y, const MxDesc m) {
    const StripDesc& d = m.d;
    constexpr int BK = 128;                                                        // fp8 channels per K-step (one 128-byte row)
    constexpr int BM = 256, BN = 128, NW = 8, NTAPS = 9, WM = 64, WN = 64, MT = 4, NT = 4;
    constexpr int PIECES = CAP / 8, SPIECES = (CAP + 63) / 64;
    constexpr int SBUF = (CAP + 8) * 128, XSB = (CAP + 8) * 4, WSTG = BN * 128, WSB = BN * 4;
    constexpr int REG = SBUF + XSB + WSTG + WSB;                                   // LDS: 2 x [strip][strip scales][weights][weight scales]
    constexpr int SCRW = 64 * 64 * 2;                                              // one wave's epilogue scratch (bf16 output tile)
    constexpr int ZW = CAP * 128 / SCRW;
    static_assert(NW * SCRW <= REG, "the epilogue scratch must fit the region the last K-step used");
    static_assert(CAP % 8 == 0 && PIECES <= NTAPS * NW && SPIECES <= NW, "one strip piece per wave per K-step");
    static_assert((CAP * 128) % SCRW + 1024 <= SCRW && SBUF + XSB <= (ZW + 1) * SCRW, "zero row and its scales inside one wave's scratch");
    static_assert(!MIRROR || (6 * 64 + 48 <= CAP && 5 * 64 + 48 + 72 <= CAP), "mirror pixels: 64-wide maps");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = tid >> 3, ch = tid & 7;
    const int wm = wave & 3, wn = wave >> 2;
    const int l16 = lane & 15, q = lane >> 4, l8 = lane >> 3, ls = lane & 7;
    const int Cin = d.Cin, HoWo = d.Ho * d.Wo, CB = Cin / 32;                      // CB: scale bytes per pixel / per weight tap row
    const int ncc = Cin / BK;
    const int ntn = d.Nrows / BN, tpi = (HoWo + BM - 1) / BM;
    const int ntiles = d.B * tpi * ntn, G = gridDim.x;

    struct Tile { int img, p0, lo, NS, n_base, ti; bool g2, valid; };
    auto get_tile = [&](int r) -> Tile {
        Tile t{};
        const int base = r * G, nwg = min(G, ntiles - base), o = blockIdx.x;
        t.valid = nwg > 0 && o < nwg;
        if (!t.valid) return t;
        const int xcd = o & 7, qq = nwg >> 3, rr = nwg & 7;
        const int bid = base + (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (o >> 3);
        t.n_base = (bid % ntn) * BN;
        const int mtile = bid / ntn;
        t.img = mtile / tpi; t.ti = mtile - t.img * tpi; t.p0 = t.ti * BM;
        const int p_last = min(t.p0 + BM, HoWo) - 1;
        t.lo = max(0, t.p0 / d.Wo + d.dh_min);
        const int hi = min(d.H - 1, p_last / d.Wo + d.dh_max);
        t.NS = (hi - t.lo + 1) * d.W;
        t.g2 = d.wp2 != nullptr && t.img >= d.group_images;
        return t;
    };

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsxs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(m.xs), 0, m.xs_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(wp1), 0, d.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(d.wp2 != nullptr ? d.wp2 : static_cast<const void*>(wp1)), 0, d.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsws1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(m.ws), 0, m.ws_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsws2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<unsigned char*>(m.ws2 != nullptr ? m.ws2 : m.ws), 0, m.ws_bytes, 0x00020000);

    auto zero_rows = [&]() {                                  // rows CAP .. CAP+7 of both strip buffers: data 0, scales 1.0
        for (int i = tid; i < 2 * 64; i += 64 * NW)
            *reinterpret_cast<u32x4_t*>(smem + (i >> 6) * REG + CAP * 128 + (i & 63) * 16) = u32x4_t{0u, 0u, 0u, 0u};
        if (tid < 2 * 8) *reinterpret_cast<unsigned*>(smem + (tid >> 3) * REG + SBUF + (CAP + (tid & 7)) * 4) = 0x7f7f7f7fu;
    };
    zero_rows();

    // ---- strip DMA (data): piece j = strip rows 8j..8j+7, rotation swizzle as in conv_strip_pk.hip (1 byte per channel here)
    const unsigned svl = (unsigned)(l8 * Cin + ((ls - (l8 & 6)) & 7) * 16);
    auto strip_base = [&](const Tile& t, int cc) -> unsigned {
        return (unsigned)__builtin_amdgcn_readfirstlane(((t.img * d.H + t.lo) * d.W) * Cin + cc * BK);
    };
    auto issue_strip_piece = [&](int j, unsigned sbase, int NS, int region) {
        const unsigned off = (8 * j + l8 < NS) ? svl + (unsigned)(8 * j * Cin) : 0xFFFFFFFFu;
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + j * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)dst, 16, (int)off, (int)sbase, 0, 0);
    };
    // ---- strip DMA (scales): piece p = the dwords of strip rows 64p .. 64p+63 (lane = row)
    auto xs_base = [&](const Tile& t, int cc) -> unsigned {
        return (unsigned)__builtin_amdgcn_readfirstlane(((t.img * d.H + t.lo) * d.W) * CB + cc * 4);
    };
    auto issue_xs_piece = [&](int p, unsigned sbase, int NS, int region) {
        const unsigned off = (64 * p + lane < NS) ? (unsigned)((64 * p + lane) * CB) : 0xFFFFFFFFu;
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + SBUF + p * 256;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsxs, (__attribute__((address_space(3))) void*)dst, 4, (int)off, (int)sbase, 0, 0);
    };
    // ---- weight tile DMA: rows n_base + lr + 64 i (XOR swizzle: rows are 16-aligned per MFMA tile), and its scale dwords
    // NISS (round 3): only waves 0 .. NISS-1 issue the K loop's DMAs (16 / NISS weight pieces, NW / NISS strip pieces per step each).  A
    // 1-KiB LDS-DMA instruction occupies the CU's one address path for ~16 cycles and a wave sits in its issue until the path takes it: with
    // all eight waves issuing their 3-4 pieces behind the step's barrier, every wave stood there ~450 cycles per step (in-kernel stamps)
    // with the matrix pipes empty.  Now the non-issuing wave of every SIMD goes straight to its fragment reads and MFMAs (-3 .. -6 % per
    // launch, A/B in one process; placing the pieces one by one behind the step's first MFMAs instead measured the same with 8 issuing
    // waves and worse with 4, and running the two waves of a SIMD half a step apart - two barriers per step - measured 4 % slower).
    static_assert(NISS == 4 || NISS == 8, "issuing waves");
    constexpr int WPI = 16 / NISS;                                                 // weight pieces per issuing wave and step
    const unsigned wvl0 = (unsigned)(lr * d.ldw + (ch ^ ((lr >> 1) & 7)) * 16);   // piece wave + NISS i: rows + 8 NISS i (same swizzle term)
    auto w_base = [&](const Tile& t, int tp, int cc) -> unsigned {
        const int te = __builtin_amdgcn_readfirstlane(d.tap[tp]);
        return (unsigned)__builtin_amdgcn_readfirstlane(t.n_base * d.ldw + (te >> 16) * Cin + cc * BK);
    };
    auto ws_base = [&](const Tile& t, int tp, int cc) -> unsigned {
        const int te = __builtin_amdgcn_readfirstlane(d.tap[tp]);
        return (unsigned)__builtin_amdgcn_readfirstlane((t.n_base * NTAPS + (te >> 16)) * CB + cc * 4);
    };
    auto issue_w1 = [&](int i, bool g2, unsigned so, int region) {
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + SBUF + XSB + (wave + NISS * i) * 1024;
        const int soi = (int)so + i * NISS * 8 * d.ldw;
        if (g2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (__attribute__((address_space(3))) void*)dst, 16, (int)wvl0, soi, 0, 0);
        else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (__attribute__((address_space(3))) void*)dst, 16, (int)wvl0, soi, 0, 0);
    };
    auto issue_ws = [&](bool g2, unsigned so, int region) {     // waves 0 and 1: the scale dwords of weight rows 64 wave + lane
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + SBUF + XSB + WSTG + wave * 256;
        const int off = (64 * wave + lane) * NTAPS * CB;
        if (g2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsws2, (__attribute__((address_space(3))) void*)dst, 4, off, (int)so, 0, 0);
        else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsws1, (__attribute__((address_space(3))) void*)dst, 4, off, (int)so, 0, 0);
    };
    auto issue_first = [&](const Tile& t, int region) {
        const unsigned sb = strip_base(t, 0), xb = xs_base(t, 0);
        for (int j = wave; 8 * j < t.NS; j += NW) issue_strip_piece(j, sb, t.NS, region);
        if (wave < SPIECES && 64 * wave < t.NS) issue_xs_piece(wave, xb, t.NS, region);
        const unsigned so = w_base(t, 0, 0);
        if (wave < NISS) {
#pragma unroll
            for (int i = 0; i < WPI; ++i) issue_w1(i, t.g2, so, region);
        }
        if (wave < 2) issue_ws(t.g2, ws_base(t, 0, 0), region);
    };

    // ---- mirror pixels of one landed strip chunk (MIRROR): block-uniform control flow, no barrier inside.  Item = (mirror pixel,
    //      16-byte chunk k = 16 channels); the two lanes k, k ^ 1 of a pixel hold one 32-channel scale block.
    const int tpi_m = tpi;
    auto mirror_fix = [&](int region, const Tile& t) {
        if constexpr (MIRROR) {
            unsigned char* sb = smem + region * REG;
            unsigned char* xsb = sb + SBUF;
            const int nrows = t.NS >> 6;
            const bool top = t.ti == 0, edge = top || t.ti == tpi_m - 1;
            const int rA = (top ? 2 : d.H - 3) - t.lo, rB = (top ? 0 : d.H - 1) - t.lo;
            const int nitem = (edge ? 12 + 66 : 12) * 8;
            auto ld = [&](int slot, int k, float (&f)[16]) {
                const u32x4_t v = *reinterpret_cast<const u32x4_t*>(sb + slot * 128 + (((k + (slot & 6)) & 7) << 4));
                const unsigned sc = (*reinterpret_cast<const unsigned*>(xsb + slot * 4) >> (8 * (k >> 1))) & 0xffu;
                const float scale = __uint_as_float(sc << 23);                         // 2^(sc - 127)
#pragma unroll
                for (int dw_ = 0; dw_ < 4; ++dw_) {
                    const auto lo2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[dw_], false), hi2 = __builtin_amdgcn_cvt_pk_f32_fp8((int)v[dw_], true);
                    f[4 * dw_] = lo2[0] * scale; f[4 * dw_ + 1] = lo2[1] * scale; f[4 * dw_ + 2] = hi2[0] * scale; f[4 * dw_ + 3] = hi2[1] * scale;
                }
            };
            for (int i = tid; i < nitem; i += 64 * NW) {
                const int px = i >> 3, k = i & 7;
                int s0, s1, s2 = -1, s3 = -1;
                bool live = true;
                if (px < 12) {
                    const int side = px >= 6 ? 1 : 0, r = px - 6 * side;
                    live = r < nrows;                                                  // the same for both lanes of a scale block
                    s0 = r * 64 + (side ? d.W - 3 : 2); s1 = r * 64 + (side ? d.W - 1 : 0);
                } else {
                    const int c = px - 12;
                    if (c < 64) { s0 = rA * 64 + c; s1 = rB * 64 + c; }
                    else {
                        const int ca = c == 64 ? 2 : d.W - 3, cb = c == 64 ? 0 : d.W - 1;
                        s0 = rA * 64 + ca; s1 = rB * 64 + ca; s2 = rA * 64 + cb; s3 = rB * 64 + cb;
                    }
                }
                if (!live) { s0 = s1 = 0; }
                float f[16], g[16];
                ld(s0, k, f); ld(s1, k, g);
#pragma unroll
                for (int e = 0; e < 16; ++e) f[e] += g[e];
                if (s2 >= 0) {
                    ld(s2, k, g);
#pragma unroll
                    for (int e = 0; e < 16; ++e) f[e] += g[e];
                    ld(s3, k, g);
#pragma unroll
                    for (int e = 0; e < 16; ++e) f[e] += g[e];
                }
                // MX re-quantisation of the 32-channel block held by lanes k (even) and k ^ 1: the rule of mx_quantize8
                float am = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) am = fmaxf(am, fabsf(f[e]));
                am = fmaxf(am, __shfl_xor(am, 1, 64));
                const int eb = (int)((__float_as_uint(am) >> 23) & 0xff);
                const int sbn = am == 0.f ? 127 : min(max(eb - 8, 0), 254);
                const float inv = __uint_as_float((unsigned)(254 - sbn) << 23);
                u32x4_t o;
#pragma unroll
                for (int dw_ = 0; dw_ < 4; ++dw_) {
                    float v4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) v4[e] = fminf(fmaxf(f[4 * dw_ + e] * inv, -448.f), 448.f);
                    int pk = __builtin_amdgcn_cvt_pk_fp8_f32(v4[0], v4[1], 0, false);
                    pk = __builtin_amdgcn_cvt_pk_fp8_f32(v4[2], v4[3], pk, true);
                    o[dw_] = (unsigned)pk;
                }
                if (live) {
                    const int so = t.NS + (px < 12 ? 8 * (px % 6) + (px >= 6 ? 5 : 2) : 48 + (px - 12 < 64 ? px - 12 : (px - 12 == 64 ? 66 : 69)));
                    *reinterpret_cast<u32x4_t*>(sb + so * 128 + (((k + (so & 6)) & 7) << 4)) = o;
                    if ((k & 1) == 0) xsb[so * 4 + (k >> 1)] = (unsigned char)sbn;
                }
            }
        }
    };

    auto now = [&]() -> unsigned long long {
        if constexpr (STAMP) {
            unsigned long long tt;
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
            __builtin_amdgcn_sched_barrier(0);
            return tt;
        } else return 0ull;
    };
    unsigned long long st_wait = 0, st_issue = 0, st_mma = 0, st_epi = 0, st_rt = 0, st_tiles = 0;
    const unsigned long long st_t0 = now();
    Tile cur = get_tile(0);
    if (!cur.valid) return;
    __syncthreads();
    issue_first(cur, 0);

    // row table: two 16-bit LDS offsets per register, unpacked at the point of use behind a compiler barrier - left to itself the
    // optimiser hoists the 36 zero-extended offsets AND the 36 scale-dword addresses derived from them out of the chunk loop (they do
    // not depend on the chunk): 72 registers, 20 of them spilled to scratch and re-loaded inside the K loop, one VMEM load per
    // tap-step whose wait drains the step's LDS-DMAs (round 3: that was the kernel's 51 % SQ_WAIT_ANY)
    unsigned rtp[NTAPS * MT / 2];
    int rt_ti = -1;
    auto rt_get = [&](int t, int b) -> unsigned {
        const int i = t * MT + b;
        unsigned v = rtp[i >> 1];
        asm volatile("" : "+v"(v));
        return (i & 1) ? (v >> 16) : (v & 0xffffu);
    };
    auto build_rt = [&](const Tile& tl) {
        const bool refl = d.pad_mode == UIG_PAD_REFLECT;
        const int ho0 = tl.p0 / d.Wo, rem0 = tl.p0 - ho0 * d.Wo;
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int pr = rem0 + wm * WM + b * 16 + l16;
            const int dho = (pr * d.wo_magic) >> 20;
            const int ho = ho0 + dho, wo = pr - dho * d.Wo;
            const bool pv = tl.p0 + wm * WM + b * 16 + l16 < HoWo;
            int hrow[3], wcol[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int hi_ = ho + (d.tap[3 * i] & 255) - 128;
                const bool ok = refl | ((unsigned)hi_ < (unsigned)d.H);
                hrow[i] = ok ? ((refl ? reflect_idx(hi_, d.H) : hi_) - tl.lo) * d.W : -65536;
                const int wi_ = wo + ((d.tap[i] >> 8) & 255) - 128;
                const bool okw = refl | ((unsigned)wi_ < (unsigned)d.W);
                wcol[i] = okw ? (refl ? reflect_idx(wi_, d.W) : wi_) : -65536;
            }
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                int s0 = hrow[t / 3] + wcol[t % 3];
                if constexpr (MIRROR) {                 // taps that read a mirror pixel instead (only where the plain tap is inside the image)
                    const int dh_ = (d.tap[3 * (t / 3)] & 255) - 128, dw_ = ((d.tap[t % 3] >> 8) & 255) - 128;
                    const bool ra = (ho == 1 && dh_ == 1) || (ho == d.H - 2 && dh_ == -1);
                    const bool cl = wo == 1 && dw_ == 1, cr = wo == d.W - 2 && dw_ == -1;
                    if (s0 >= 0) {
                        if (ra) s0 = tl.NS + 48 + (cl ? 66 : (cr ? 69 : wo + dw_));
                        else if (cl | cr) s0 = tl.NS + 8 * (ho + dh_ - tl.lo) + (cr ? 5 : 2);
                    }
                }
                const int s = (pv & (s0 >= 0)) ? s0 : CAP;
                const unsigned av = (unsigned)(s * 128 + (((q + (s & 6)) & 7) << 4)) & 0xffffu;
                const int ri = t * MT + b;
                rtp[ri >> 1] = (ri & 1) ? ((rtp[ri >> 1] & 0xffffu) | (av << 16)) : ((rtp[ri >> 1] & 0xffff0000u) | av);
            }
        }
    };

#pragma unroll
    for (int i = 0; i < NTAPS * MT / 2; ++i) rtp[i] = 0u;
    const int wswz = (l16 >> 1) & 7;
    const int co_lo = (q ^ wswz) << 4;
    const int sh = 8 * q;                                      // this lane's K block inside a scale dword
    int par = 0;
    for (int r = 0;; ++r) {
        const Tile nxt = get_tile(r + 1);
        const unsigned long long st_a = now();
        if (cur.ti != rt_ti) { build_rt(cur); rt_ti = cur.ti; }
        if constexpr (STAMP) { st_rt += now() - st_a; ++st_tiles; }
        if constexpr (MIRROR) {
            if (r == 0) {                                      // the block's first chunk: nothing ran in front of it to hide this behind
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                mirror_fix(0, cur);
            }
        }
        const float* bias = cur.g2 ? d.bias2 : bias1;

        f32x4_t acc[NT][MT];
        strip_init_acc<MT, NT, WN>(acc, bias, d.Nrows, cur.n_base, wn, lane);

        // the DMA issue that belongs to step (cc, t): the NEXT step's weights + their scales, one piece of the next chunk's strip, its scales
        // (plain scalars, no Tile references selected at run time: those put the two Tile objects into scratch memory)
        const unsigned sb_c = strip_base(cur, 0), sb_n = strip_base(nxt, 0), xb_c = xs_base(cur, 0), xb_n = xs_base(nxt, 0);
        const int NS_c = cur.NS, NS_n = nxt.NS, nb_c = cur.n_base, nb_n = nxt.n_base;
        const bool g2_c = cur.g2, g2_n = nxt.g2, nxt_ok = nxt.valid;
        constexpr int NK = WPI + 1 + NW / NISS;
        auto issue_piece = [&](int cc, int t, int k) {
            const int pc = par ^ (cc & 1);
            const bool last_cc = cc + 1 == ncc, last_t = t + 1 == NTAPS;
            const bool pre_next = last_cc && nxt_ok;
            const bool s_on = !last_cc || pre_next;
            const unsigned s_base = !last_cc ? sb_c + (unsigned)((cc + 1) * BK) : sb_n;
            const unsigned x_base = !last_cc ? xb_c + (unsigned)((cc + 1) * 4) : xb_n;
            const int s_NS = !last_cc ? NS_c : NS_n;
            const bool w_on = !(last_t && last_cc) || pre_next;
            const bool wrap = last_t && last_cc;                               // the next step is the next tile's first
            const bool w_g2 = wrap ? g2_n : g2_c;
            const int tn = last_t ? 0 : t + 1, ccn = wrap ? 0 : (last_t ? cc + 1 : cc), nbn = wrap ? nb_n : nb_c;
            const int te = __builtin_amdgcn_readfirstlane(d.tap[tn]) >> 16;
            const unsigned w_so = (unsigned)__builtin_amdgcn_readfirstlane(nbn * d.ldw + te * Cin + ccn * BK);
            const unsigned ws_so = (unsigned)__builtin_amdgcn_readfirstlane((nbn * NTAPS + te) * CB + ccn * 4);
            const int w_reg = pc ^ ((t + 1) & 1);
            // piece k of the step's NK DMA instructions of this wave: WPI weight pieces, the weight scales (waves 0, 1), NW / NISS strip pieces + scales
            if (k < WPI) { if (w_on) issue_w1(k, w_g2, w_so, w_reg); }
            else if (k == WPI) { if (w_on && wave < 2) issue_ws(w_g2, ws_so, w_reg); }
            else {
                const int kk = k - WPI - 1;
                const int slot = t * NW + wave + NISS * kk;
                if (s_on && slot < PIECES && 8 * slot < s_NS) issue_strip_piece(slot, s_base, s_NS, pc ^ 1);
                const int xp = wave + NISS * kk;
                if (t == (MIRROR ? 0 : NTAPS - 1) && s_on && xp < SPIECES && 64 * xp < s_NS) issue_xs_piece(xp, x_base, s_NS, pc ^ 1);
            }
        };
        for (int cc = 0; cc < ncc; ++cc) {
            const int pc = par ^ (cc & 1);
            const unsigned char* sx = smem + pc * REG;
            const bool last_cc = cc + 1 == ncc;
            const bool s_on = !last_cc || nxt_ok;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const unsigned long long st_0 = now();
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                const unsigned long long st_1 = now();
                const bool last_t = t + 1 == NTAPS;
                if (wave < NISS) {
#pragma unroll
                    for (int k = 0; k < NK; ++k) issue_piece(cc, t, k);
                }
                if constexpr (MIRROR) {      // next chunk's strip (steps 0-6) and scales (step 0) have landed and been published: its mirror pixels
                    if (last_t && s_on) mirror_fix(pc ^ 1, last_cc ? nxt : cur);
                }

                const unsigned long long st_2 = now();
                const unsigned char* swb = smem + (pc ^ (t & 1)) * REG + SBUF + XSB;
                const unsigned char* sw = swb + (wn * WN + l16) * 128;
                // Round 3: the step's fragments in two halves over the pixel groups (4 weight fragments + 2 x 2 strip fragments: 48 operand
                // registers live at once instead of 64).  With all 8 fragments loaded up front the kernel sat at 256 VGPRs with 20 of them
                // spilled to scratch inside the K loop (84 B per lane: scratch traffic counted on the same vmcnt the step's DMA wait drains).
                i32x8_t wf[NT];
                int wsc[NT];
#pragma unroll
                for (int a = 0; a < NT; ++a) {
                    const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co_lo), hi = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + (co_lo ^ 64));
                    wf[a] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                    wsc[a] = (int)(*reinterpret_cast<const unsigned*>(swb + WSTG + (wn * WN + a * 16 + l16) * 4) >> sh);
                }
#pragma unroll
                for (int bh = 0; bh < MT; bh += 2) {
                    i32x8_t xf[2];
                    int xsc[2];
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        const unsigned a0 = rt_get(t, bh + b);
                        const u32x4_t lo = *reinterpret_cast<const u32x4_t*>(sx + a0), hi = *reinterpret_cast<const u32x4_t*>(sx + (a0 ^ 64u));
                        xf[b] = i32x8_t{(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
                        xsc[b] = (int)(*reinterpret_cast<const unsigned*>(sx + SBUF + ((a0 >> 7) << 2)) >> sh);
                    }
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b)
                            acc[a][bh + b] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(wf[a], xf[b], acc[a][bh + b], 0, 0, 0, wsc[a], 0, xsc[b]);
                    if (bh == 0) __builtin_amdgcn_sched_barrier(0);      // keep the second half's loads behind the first half's MFMAs
                }
                // pin the accumulators here: the MFMAs are pure register operations, and without a use inside the step the
                // optimiser sinks the whole chain of all nine taps below the last tap's loads (seen: 144 MFMAs after the last
                // barrier, 550 registers of fragments spilled to scratch)
#pragma unroll
                for (int a = 0; a < NT; ++a)
#pragma unroll
                    for (int b = 0; b < MT; ++b) asm volatile("" : "+v"(acc[a][b]));
                if constexpr (STAMP) { const unsigned long long st_3 = now(); st_wait += st_1 - st_0; st_issue += st_2 - st_1; st_mma += st_3 - st_2; }
            }
        }
        const int pl = par ^ ((ncc - 1) & 1);
        const unsigned long long st_e = now();

        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        unsigned char* scratch = smem + pl * REG + wave * SCRW;
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        strip_epilogue<bf16_t, MT, NT, WM, WN, !MIRROR>(acc, scratch, d, y, cur.img, cur.p0, wm, wn, cur.n_base, lane_e);
        if constexpr (STAMP) st_epi += now() - st_e;
        if (!nxt.valid) break;
        if (wave == ZW) {                                      // restore the zero row and its scales (the scratch covered them)
            *reinterpret_cast<u32x4_t*>(smem + pl * REG + CAP * 128 + lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
            if (lane < 8) *reinterpret_cast<unsigned*>(smem + pl * REG + SBUF + (CAP + lane) * 4) = 0x7f7f7f7fu;
        }
        par = pl ^ 1;
        cur = nxt;
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned long long st_end = now();
        if (lane == 0 && d.dbg != nullptr) {
            unsigned long long* o = d.dbg + ((long)blockIdx.x * NW + wave) * 8;
            o[0] = st_end - st_t0; o[1] = st_wait; o[2] = st_issue; o[3] = st_mma; o[4] = st_epi; o[5] = st_rt; o[6] = st_tiles; o[7] = 0;
        }
    }
}


static int fp8_device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

static int strip_rows_needed256(int H, int W, int dh_min, int dh_max) {
    int worst = 0;
    for (int p0 = 0; p0 < H * W; p0 += 256) {
        const int pl = std::min(p0 + 256, H * W) - 1;
        const int lo = std::max(0, p0 / W + dh_min), hi = std::min(H - 1, pl / W + dh_max);
        worst = std::max(worst, (hi - lo + 1) * W);
    }
    return worst;
}

static unsigned long long* g_mx_dbg = nullptr;      // diagnostic: device buffer for the STAMP build (8 x u64 per wave)
extern "C" void uig_debug_set_mx_stamps(void* dev_buf) { g_mx_dbg = (unsigned long long*)dev_buf; }

static int g_mx_niss = 4;    // tuning / A-B hook: waves that issue the K loop's DMAs (8 = all: the round-2 form)
extern "C" void uig_debug_set_mx_issuers(int n) { g_mx_niss = n == 8 ? 8 : 4; }

template <bool MIRROR, bool STAMP = false, int NISS = 4>
static int launch_mx(const MxDesc& m_in, const void* xq, const void* wq, const float* bias, void* y, void* stream) {
    constexpr int CAP = 448;
    if constexpr (!STAMP && NISS == 4) {
        if (g_mx_dbg != nullptr) return g_mx_niss == 8 ? launch_mx<MIRROR, true, 8>(m_in, xq, wq, bias, y, stream) : launch_mx<MIRROR, true, 4>(m_in, xq, wq, bias, y, stream);
        if (g_mx_niss == 8) return launch_mx<MIRROR, false, 8>(m_in, xq, wq, bias, y, stream);
    }
    MxDesc m = m_in;
    if constexpr (STAMP) m.d.dbg = g_mx_dbg;
    const StripDesc& d = m.d;
    const size_t smem = 2 * ((size_t)(CAP + 8) * 128 + (CAP + 8) * 4 + 128 * 128 + 128 * 4);
    auto kern = conv_strip_fp8_kernel<CAP, MIRROR, STAMP, NISS>;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return uig_set_error((int)e, "conv_strip_fp8: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    const int ntiles = d.B * ((d.H * d.W + 255) / 256) * (d.Nrows / 128);
    hipLaunchKernelGGL(kern, dim3(std::min(ntiles, fp8_device_cus())), dim3(512), smem, (hipStream_t)stream,
                       (const unsigned char*)xq, (const unsigned char*)wq, bias, (bf16_t*)y, m);
    UIG_LAUNCH_CHECK("uig_conv3x3_mx_fp8");
    return 0;
}

extern "C" int uig_conv3x3_mx_fp8_applicable(int B, int H, int W, int Cin, int Nrows) {
    if (B <= 0 || H < 4 || W < 4 || W > 512 || Cin % 128 != 0 || Nrows % 128 != 0 || Nrows < 128) return 0;
    if ((long)B * H * W * Cin >= (1L << 32) - 64 || (long)Nrows * 9 * Cin >= (1L << 32) - 64) return 0;
    return strip_rows_needed256(H, W, -1, 1) <= 448 ? 1 : 0;
}

extern "C" int uig_conv3x3_mx_fp8(const void* xq, const void* xs, const void* wq, const void* ws, const float* bias,
                                  const void* wq2, const void* ws2, const float* bias2, int group_images,
                                  float* in_partial, const void* border_add, const void* res_add, void* y,
                                  int B, int H, int W, int Cin, int Nrows, int pad_mode, int gather_mode, int ldc,
                                  int act, float slope,
                                  const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial, void* stream) {
    UIG_CHECK_ARG(xq && xs && wq && ws && y, "uig_conv3x3_mx_fp8: null pointer");
    if (bst_partial != nullptr)
        UIG_CHECK_ARG(bst_x && bst_stats && (border_add || res_add) && (H * W) % 64 == 0, "uig_conv3x3_mx_fp8: fused InstanceNorm-backward statistics need border_add / res_add and H*W %% 64 == 0");
    UIG_CHECK_ARG(uig_conv3x3_mx_fp8_applicable(B, H, W, Cin, Nrows) == 1,
                  "uig_conv3x3_mx_fp8: unsupported shape B=%d %dx%d Cin=%d N=%d (Cin, N multiples of 128; 256-pixel strips <= 448 rows)", B, H, W, Cin, Nrows);
    UIG_CHECK_ARG(gather_mode == UIG_GATHER_DIRECT || gather_mode == UIG_GATHER_TRANSPOSED, "uig_conv3x3_mx_fp8: bad gather_mode %d", gather_mode);
    UIG_CHECK_ARG(pad_mode == UIG_PAD_ZERO || (pad_mode == UIG_PAD_REFLECT && gather_mode == UIG_GATHER_DIRECT), "uig_conv3x3_mx_fp8: reflect pad needs direct mode");
    UIG_CHECK_ARG(ldc >= Nrows && (ldc * 2) % 16 == 0, "uig_conv3x3_mx_fp8: bad ldc %d", ldc);
    if (wq2 != nullptr) UIG_CHECK_ARG(ws2 != nullptr && group_images > 0 && group_images < B, "uig_conv3x3_mx_fp8: bad pair (group_images=%d, B=%d)", group_images, B);
    if (border_add != nullptr) UIG_CHECK_ARG(H == W, "uig_conv3x3_mx_fp8: border_add needs a square map");
    MxDesc m{};
    StripDesc& d = m.d;
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Ho = H; d.Wo = W; d.pad_mode = pad_mode; d.dh_min = -1; d.dh_max = 1;
    d.Nrows = Nrows; d.ldw = 9 * Cin; d.ldc = ldc; d.Nstore = Nrows; d.act = act; d.slope = slope;
    d.x_bytes = (unsigned)((long)B * H * W * Cin); d.w_bytes = (unsigned)((long)Nrows * 9 * Cin);
    d.wp2 = wq2; d.bias2 = bias2; d.group_images = group_images; d.in_partial = in_partial; d.border_add = border_add; d.res_add = res_add;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) {
            const int t = kh * 3 + kw;      // same tap tables as uig_conv_gather: direct (dh = kh - 1) / transposed stride 1 (dh = 1 - kh)
            d.tap[t] = gather_mode == UIG_GATHER_DIRECT ? (((kh - 1) + 128) | (((kw - 1) + 128) << 8) | (t << 16))
                                                         : (((1 - kh) + 128) | (((1 - kw) + 128) << 8) | (t << 16));
        }
    d.wo_magic = ((1 << 20) + W - 1) / W;
    if (bst_partial != nullptr) { d.bst_x = bst_x; d.bst_stats = bst_stats; d.bst_partial = bst_partial; d.bst_act = bst_act; d.bst_slope = bst_slope; }
    m.xs = (const unsigned char*)xs; m.ws = (const unsigned char*)ws; m.ws2 = (const unsigned char*)ws2;
    m.xs_bytes = (unsigned)((long)B * H * W * (Cin / 32)); m.ws_bytes = (unsigned)((long)Nrows * 9 * (Cin / 32));
    return launch_mx<false>(m, xq, wq, bias, y, stream);
}

// The input gradient of a REFLECTION-padded 3x3 convolution, fp8 operands, in one launch (MIRROR variant of the kernel above): dq / ds
// = MX-quantised output gradient [B,H,W,C], wq / ws = the MX-quantised tap-major weights of the transposed gather.  Replaces
// uig_conv3x3_mx_fp8(pad zero, transposed) + the bf16 border GEMM (uig_conv_dgrad_border) of the round-2 form.
extern "C" int uig_conv3x3_mx_fp8_dgrad_mirror_applicable(int B, int H, int W, int Cin, int Nrows) {
    if (uig_conv3x3_mx_fp8_applicable(B, H, W, Cin, Nrows) != 1) return 0;
    return (W == 64 && H >= 8 && H % 4 == 0) ? 1 : 0;
}

extern "C" int uig_conv3x3_mx_fp8_dgrad_mirror(const void* dq, const void* ds, const void* wq, const void* ws, const void* wq2, const void* ws2,
                                               int group_images, const void* res_add, void* dx, int B, int H, int W, int Cin, int Nrows, int ldc,
                                               void* stream) {
    UIG_CHECK_ARG(dq && ds && wq && ws && dx, "uig_conv3x3_mx_fp8_dgrad_mirror: null pointer");
    UIG_CHECK_ARG(uig_conv3x3_mx_fp8_dgrad_mirror_applicable(B, H, W, Cin, Nrows) == 1,
                  "uig_conv3x3_mx_fp8_dgrad_mirror: unsupported shape B=%d %dx%d Cin=%d N=%d (64-wide maps, H %% 4 == 0, channels multiples of 128)", B, H, W, Cin, Nrows);
    UIG_CHECK_ARG(ldc >= Nrows && (ldc * 2) % 16 == 0, "uig_conv3x3_mx_fp8_dgrad_mirror: bad ldc %d", ldc);
    if (wq2 != nullptr) UIG_CHECK_ARG(ws2 != nullptr && group_images > 0 && group_images < B, "uig_conv3x3_mx_fp8_dgrad_mirror: bad pair (group_images=%d, B=%d)", group_images, B);
    MxDesc m{};
    StripDesc& d = m.d;
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.Ho = H; d.Wo = W; d.pad_mode = UIG_PAD_ZERO; d.dh_min = -1; d.dh_max = 1;
    d.Nrows = Nrows; d.ldw = 9 * Cin; d.ldc = ldc; d.Nstore = Nrows; d.act = UIG_ACT_NONE; d.slope = 0.f;
    d.x_bytes = (unsigned)((long)B * H * W * Cin); d.w_bytes = (unsigned)((long)Nrows * 9 * Cin);
    d.wp2 = wq2; d.group_images = group_images; d.res_add = res_add;
    for (int kh = 0; kh < 3; ++kh)
        for (int kw = 0; kw < 3; ++kw) d.tap[kh * 3 + kw] = ((1 - kh) + 128) | (((1 - kw) + 128) << 8) | ((kh * 3 + kw) << 16);
    d.wo_magic = ((1 << 20) + W - 1) / W;
    m.xs = (const unsigned char*)ds; m.ws = (const unsigned char*)ws; m.ws2 = (const unsigned char*)ws2;
    m.xs_bytes = (unsigned)((long)B * H * W * (Cin / 32)); m.ws_bytes = (unsigned)((long)Nrows * 9 * (Cin / 32));
    return launch_mx<true>(m, dq, wq, nullptr, dx, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// MX quantisation of a [P][C] bf16 / f32 matrix along C (blocks of 32): q[P][C] e4m3 bytes + s[P][C/32] E8M0 bytes.
// A thread owns 8 consecutive elements; the 4 threads of a block combine their maxima with two xor-shuffles.
template <typename T>
__global__ void mx_quantize_kernel(const T* __restrict__ x, unsigned char* __restrict__ q, unsigned char* __restrict__ s, long n8, int C) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // chunk of 8 elements
    const bool ok = i < n8;
    float f[8];
    if (ok) {
        if constexpr (sizeof(T) == 2) {
            chunk_to_f32<bf16_t>(*reinterpret_cast<const u32x4_t*>(x + 8 * i), f);
        } else {
            const f32x4_t a = *reinterpret_cast<const f32x4_t*>(x + 8 * i), b = *reinterpret_cast<const f32x4_t*>(x + 8 * i + 4);
            f[0] = a[0]; f[1] = a[1]; f[2] = a[2]; f[3] = a[3]; f[4] = b[0]; f[5] = b[1]; f[6] = b[2]; f[7] = b[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
    int sb;
    const u32x2_t w = mx_quantize8(f, sb);
    if (ok) {
        *reinterpret_cast<u32x2_t*>(q + 8 * i) = w;
        if ((i & 3) == 0) s[i >> 2] = (unsigned char)sb;
    }
}

// Every fp8 weight operand of every layer in ONE launch (the per-step refresh after Adam + repack: 72 quantiser launches of ~4 us
// each at the launch floor otherwise).  items: device array of {x (bf16), q, s, n8 (16-byte chunks of 8 elements), block_end}
// records (40 bytes), block_end = inclusive prefix sum of ceil(n8 / 256): a block of 256 threads finds its record by bisection.
struct MxQItem { const bf16_t* x; unsigned char* q; unsigned char* s; long n8; long block_end; };
__global__ __launch_bounds__(256) void mx_quantize_multi_kernel(const MxQItem* __restrict__ items, int nitems) {
    __shared__ int s_item;
    if (threadIdx.x == 0) {
        int lo = 0, hi = nitems - 1;
        while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].block_end > (long)blockIdx.x) hi = mid; else lo = mid + 1; }
        s_item = lo;
    }
    __syncthreads();
    const int ii = s_item;
    const MxQItem it = items[ii];
    const long b0 = ii ? items[ii - 1].block_end : 0;
    const long i = ((long)blockIdx.x - b0) * 256 + threadIdx.x;
    const bool ok = i < it.n8;
    float f[8];
    if (ok) chunk_to_f32<bf16_t>(*reinterpret_cast<const u32x4_t*>(it.x + 8 * i), f);
    else {
#pragma unroll
        for (int e = 0; e < 8; ++e) f[e] = 0.f;
    }
    int sb;
    const u32x2_t w = mx_quantize8(f, sb);
    if (ok) {
        *reinterpret_cast<u32x2_t*>(it.q + 8 * i) = w;
        if ((i & 3) == 0) it.s[i >> 2] = (unsigned char)sb;
    }
}
extern "C" int uig_mx_quantize_multi(const void* items_dev, int nitems, long total_blocks, void* stream) {
    UIG_CHECK_ARG(items_dev && nitems > 0 && total_blocks > 0 && total_blocks < (1L << 31), "uig_mx_quantize_multi: bad args");
    hipLaunchKernelGGL(mx_quantize_multi_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream, (const MxQItem*)items_dev, nitems);
    UIG_LAUNCH_CHECK("uig_mx_quantize_multi");
    return 0;
}

extern "C" int uig_mx_quantize(const void* x, void* q, void* scales, long P, int C, int dtype, void* stream) {
    UIG_CHECK_ARG(x && q && scales, "uig_mx_quantize: null pointer");
    UIG_CHECK_ARG(P > 0 && C > 0 && C % 32 == 0, "uig_mx_quantize: C=%d must be a positive multiple of 32 (P=%ld)", C, P);
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_mx_quantize: bad dtype %d", dtype);
    const long n8 = P * C / 8;
    const int blocks = (int)((n8 + 255) / 256);
    if (dtype == UIG_BF16) hipLaunchKernelGGL(mx_quantize_kernel<bf16_t>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)x, (unsigned char*)q, (unsigned char*)scales, n8, C);
    else hipLaunchKernelGGL(mx_quantize_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)x, (unsigned char*)q, (unsigned char*)scales, n8, C);
    UIG_LAUNCH_CHECK("uig_mx_quantize");
    return 0;
}
