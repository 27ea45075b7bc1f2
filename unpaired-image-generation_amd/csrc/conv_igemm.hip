// conv_igemm.hip — implicit-GEMM gather convolution for gfx950 (MI355X).
//
// One kernel serves Conv2d forward, ConvTranspose2d forward and both of their input gradients
// (aten::convolution / aten::convolution_backward[input]); see uig_conv_gather in include/uig.h.
//
//   D[n][m] = sum_k  Wp[n][k] * X[pix(m, tap(k))][c(k)]        k = (tap, c),  c contiguous (NHWC)
//
// Layout in HBM: activations NHWC (channels padded to 8), packed weights [N][tap][C]: both operands are
// K-contiguous, so every lane moves 16-byte chunks.  Per K-step the block streams a BM-pixel x 128-byte im2col tile
// and a BN-channel x 128-byte weight tile straight into XOR-swizzled LDS (buffer_load ... lds, double buffered, one
// barrier per K-step), and each of the 4 waves runs MFMA 16x16 tiles over its WM x WN sub-tile:
//   bf16: v_mfma_f32_16x16x32_bf16 (BK = 64),   f32: v_mfma_f32_16x16x4_f32 (BK = 32, exact f32 fmaf chain).
// The weight tile is the MFMA A operand and the pixel tile the B operand, so each lane ends up holding 4
// consecutive output channels of one pixel (8/16-byte stores into NHWC).
// Reflection padding is resolved in the gather address map; zero padding and ragged tiles by the buffer range check.
#include "uig_common.h"
#include <algorithm>


struct GatherDesc {
    int B, H, W, Cin;
    int Mh, Mw;            // per-phase iteration grid (per image)
    int si, so;            // input stride, output stride
    int pad_mode;
    int Nrows, ldw;
    int Ho, Wo, ldc, Nstore;
    int act; float slope;
    int cin_shift;
    unsigned x_bytes, w_bytes;   // buffer sizes for the hardware range check
    int nphase;
    int ph_tap0[9];
    signed char ph_oh[8], ph_ow[8];
    unsigned char ph_swap[8];   // phase iterates its grid transposed: (ii, jj) <- (jj, ii)   (border-line launches)
    int compact_out;            // output row = (image * nphase + phase) * Mh*Mw + ii*Mw + jj  instead of the (ho, wo) map
    int tap[64];           // per tap: (dh + 128) | (dw + 128) << 8 | weight-tap-index << 16  (one scalar dword load)
    const void* wp2;       // paired launch: GEMM rows >= group_rows (the second network's images) use wp2 / bias2
    const float* bias2;
    int group_rows;
    float* in_partial;     // optional: InstanceNorm partial statistics [img][Ho*Wo/64][Nstore][2] (needs Mh*Mw % 64 == 0)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
    }
};

// Tile configuration: BM pixels x BN channels per block, WAVES_M x WAVES_N waves (64 threads each), NSTAGE LDS stages.
// NSTAGE == 2: one tile in flight behind the one being consumed (drained with vmcnt(0) each K-step).
// NSTAGE >= 3: NSTAGE-1 tiles in flight; a COUNTED s_waitcnt vmcnt(N) retires only the oldest tile and a raw s_barrier
//              (no implicit drain) orders it for the other waves' ds_reads: the LDS-DMA stream never stops.
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool SMALL_CIN>
__global__ __launch_bounds__(64 * WAVES_M * WAVES_N, 2)
void igemm_kernel(const T* __restrict__ x, const T* __restrict__ wp_, const float* __restrict__ bias_, T* __restrict__ y,
                  const GatherDesc d) {
    constexpr int E = ElemTraits<T>::E;
    constexpr int BK = 8 * E;                 // 128 bytes of K per row per step
    constexpr int NWAVES = WAVES_M * WAVES_N, RPP = NWAVES * 8;     // rows staged per pass of the whole block
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int MT = WM / 16, NT = WN / 16;
    constexpr int RA = BM / RPP, RB = (BN + RPP - 1) / RPP;
    constexpr int STAGE = (BM + BN) * 128;
    static_assert(WM % 16 == 0 && WN % 16 == 0, "wave tile must be a multiple of the 16x16 MFMA tile");
    static_assert(BM % RPP == 0, "BM must be a multiple of the rows staged per pass");
    static_assert(NSTAGE == 2 || BN % RPP == 0, "counted vmcnt needs every wave to issue the same number of DMAs");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = tid >> 3, ch = tid & 7;

    // ---- tile coordinates (XCD-aware: blocks that share an XCD get a contiguous run of tiles)
    const int nwg = gridDim.x;
    int bid;
    {
        const int o = blockIdx.x, xcd = o & 7, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (o >> 3);
    }
    const int ntn = (d.Nrows + BN - 1) / BN;
    // paired launch: each network's rows get their own run of tiles (group 1 starts at row group_rows whatever its alignment),
    // so a block never straddles the two weight sets; rows past the end of a group are masked like the ragged last tile
    const int mtile = bid / ntn;
    const int gt0 = d.wp2 != nullptr ? (d.group_rows + BM - 1) / BM : 0;
    const bool g2 = d.wp2 != nullptr && mtile >= gt0;
    const int n_base = (bid % ntn) * BN, m_base = g2 ? d.group_rows + (mtile - gt0) * BM : mtile * BM;
    const T* wp = g2 ? static_cast<const T*>(d.wp2) : wp_;
    const float* bias = g2 ? d.bias2 : bias_;
    const int ph = blockIdx.y;
    const int tap0 = d.ph_tap0[ph], ntap = d.ph_tap0[ph + 1] - tap0;
    const int M = (d.wp2 != nullptr && !g2) ? d.group_rows : d.B * d.Mh * d.Mw;      // end of this block's group
    const int Cin = d.Cin;

    // ---- per-thread row bookkeeping for the gather
    // (image, row, column) of a thread's staged rows: ONE pair of integer divisions (row 0), the others by carrying - four
    // runtime divisions per row were ~100 VALU instructions each for every thread of every block, and these layers run 9-18 K-steps
    int hb[RA], wb[RA], ib[RA];
    unsigned vmask = 0;
    {
        const int m0 = m_base + lr;
        int jj0 = m0 % d.Mw, t = m0 / d.Mw, ii0 = t % d.Mh, b = t / d.Mh;
        const bool sw = d.ph_swap[blockIdx.y] != 0;
#pragma unroll
        for (int i = 0; i < RA; ++i) {
            const bool v = m0 + RPP * i < M;
            const int ii = sw ? jj0 : ii0, jj = sw ? ii0 : jj0;
            hb[i] = v ? ii * d.si : 0; wb[i] = v ? jj * d.si : 0; ib[i] = v ? b * d.H * d.W : 0;
            vmask |= (v ? 1u : 0u) << i;
            jj0 += RPP;
            while (jj0 >= d.Mw) { jj0 -= d.Mw; if (++ii0 == d.Mh) { ii0 = 0; ++b; } }
        }
    }
    int wrow[RB];
    unsigned nmask = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) {
        const int r = lr + RPP * i, n = n_base + r;
        const bool v = (r < BN) && (n < d.Nrows);
        wrow[i] = (v ? n : 0) * d.ldw;
        nmask |= (v ? 1u : 0u) << i;
    }

    const int ktot = ntap * Cin;
    const int nk = (ktot + BK - 1) / BK;

    // Buffer descriptors (wave-uniform: kernel arguments only). Out-of-range voffset (0xFFFFFFFF) makes the hardware
    // return zeros: zero padding and ragged tiles cost one select per tap instead of a predicated load per K-step.
    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp), 0, d.w_bytes, 0x00020000);
    // Tiles go HBM/L2 -> LDS directly (buffer_load ... lds, 16 B per lane, no VGPR staging, no ds_write).  An LDS-DMA
    // wave-instruction writes 1 KiB linearly = 8 rows x 128 B, lane L -> row L/8, chunk slot L%8, which is exactly this
    // thread mapping (wave w, slot i covers rows 8w+32i .. +7).  The XOR swizzle that keeps the ds_read_b128 fragment
    // reads conflict-free is applied on the SOURCE side: slot s of row r holds global chunk s ^ ((r>>1)&7).
    const int chs = ch ^ ((lr >> 1) & 7);
    const int wave_row = __builtin_amdgcn_readfirstlane(wave * 8);
    int tap_l = 0, c0 = 0;        // running (tap, channel) of the K-step being loaded (non-small mode)
    unsigned xvo[RA], wvo[RB];    // per-row byte offsets of the current tap; the channel offset rides in the SGPR soffset
    int wso = 0;
#pragma unroll
    for (int i = 0; i < RB; ++i) wvo[i] = ((nmask >> i) & 1u) ? (unsigned)((wrow[i] + chs * E) * (int)sizeof(T)) : 0xFFFFFFFFu;

    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    auto issue_tile = [&](int ks, int stage) {
        lds_ptr_t sx = (lds_ptr_t)smem + stage * STAGE + wave_row * 128;
        lds_ptr_t sw = sx + BM * 128;
        if constexpr (SMALL_CIN) {
            const int kf = ks * BK + chs * E;
            const int tl = kf >> d.cin_shift, c = kf & (Cin - 1);
            const bool kok = tl < ntap;
            const int tap = tap0 + (kok ? tl : 0);
            const int te = d.tap[tap];
            const int ddh = (te & 255) - 128, ddw = ((te >> 8) & 255) - 128, wtap = te >> 16;
            const bool refl = d.pad_mode == UIG_PAD_REFLECT;
#pragma unroll
            for (int i = 0; i < RA; ++i) {
                const int hi = hb[i] + ddh, wi = wb[i] + ddw;
                const bool inb = ((unsigned)hi < (unsigned)d.H) & ((unsigned)wi < (unsigned)d.W);
                const bool ok = kok & (((vmask >> i) & 1u) != 0) & (refl | inb);
                const int hr = refl ? reflect_idx(hi, d.H) : hi, wr = refl ? reflect_idx(wi, d.W) : wi;
                const unsigned off = ok ? (unsigned)(((ib[i] + hr * d.W + wr) * Cin + c) * (int)sizeof(T)) : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(sx + i * RPP * 128), 16, (int)off, 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if constexpr (BN % RPP != 0) { if (wave_row + RPP * i >= BN) continue; }       // wave-uniform: rows past the weight tile
                const bool ok = kok & (((nmask >> i) & 1u) != 0);
                const unsigned off = ok ? (unsigned)((wrow[i] + wtap * Cin + c) * (int)sizeof(T)) : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(sw + i * RPP * 128), 16, (int)off, 0, 0, 0);
            }
        } else {
            if (c0 == 0) {       // new tap (block-uniform branch): recompute the gather offsets once per Cin/BK K-steps
                // readfirstlane makes the uniformity provable: scalar kernarg loads, no waterfall loops around the buffer ops
                const int tap = __builtin_amdgcn_readfirstlane(tap0 + tap_l);
                const int te = __builtin_amdgcn_readfirstlane(d.tap[tap]);
                const int ddh = (te & 255) - 128, ddw = ((te >> 8) & 255) - 128;
                wso = __builtin_amdgcn_readfirstlane((te >> 16) * Cin * (int)sizeof(T));
                const bool refl = d.pad_mode == UIG_PAD_REFLECT;
#pragma unroll
                for (int i = 0; i < RA; ++i) {
                    const int hi = hb[i] + ddh, wi = wb[i] + ddw;
                    const bool inb = ((unsigned)hi < (unsigned)d.H) & ((unsigned)wi < (unsigned)d.W);
                    const bool ok = (((vmask >> i) & 1u) != 0) & (refl | inb);
                    const int hr = refl ? reflect_idx(hi, d.H) : hi, wr = refl ? reflect_idx(wi, d.W) : wi;
                    const unsigned off = (unsigned)(((ib[i] + hr * d.W + wr) * Cin + chs * E) * (int)sizeof(T));
                    xvo[i] = ok ? off : 0xFFFFFFFFu;
                }
            }
            const int so = __builtin_amdgcn_readfirstlane(c0 * (int)sizeof(T));
            const int wso2 = __builtin_amdgcn_readfirstlane(wso + so);
#pragma unroll
            for (int i = 0; i < RA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)(sx + i * RPP * 128), 16, (int)xvo[i], so, 0, 0);
#pragma unroll
            for (int i = 0; i < RB; ++i) {
                if constexpr (BN % RPP != 0) { if (wave_row + RPP * i >= BN) continue; }
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (__attribute__((address_space(3))) void*)(sw + i * RPP * 128), 16, (int)wvo[i], wso2, 0, 0);
            }
            c0 += BK; if (c0 >= Cin) { c0 = 0; ++tap_l; }
        }
    };

    f32x4_t acc[NT][MT];
#pragma unroll
    for (int a = 0; a < NT; ++a)
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int wm = wave % WAVES_M, wn = wave / WAVES_M;
    const int l16 = lane & 15, q = lane >> 4;
    const int swz = (l16 >> 1) & 7;

    // ---- main loop: NSTAGE-deep ring of LDS stages filled by LDS-DMA, one barrier per K-step
    constexpr int PER_TILE = RA + RB;                       // DMA instructions per wave per tile
#pragma unroll
    for (int st = 0; st < NSTAGE - 1; ++st)
        if (st < nk) issue_tile(st, st);
    for (int ks = 0; ks < nk; ++ks) {
        // retire tile ks: this wave's own DMAs by vmcnt, the other waves' by the barrier behind it
        if constexpr (NSTAGE == 2) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (ks + NSTAGE - 2 < nk) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NSTAGE - 2) * PER_TILE) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        // every wave is past its reads of stage (ks-1) % NSTAGE: refill it with tile ks + NSTAGE - 1
        if (ks + NSTAGE - 1 < nk) issue_tile(ks + NSTAGE - 1, (ks + NSTAGE - 1) % NSTAGE);
        const int cur = ks % NSTAGE;
        const unsigned char* sx = smem + cur * STAGE + (wm * WM + l16) * 128;
        const unsigned char* sw = smem + cur * STAGE + BM * 128 + (wn * WN + l16) * 128;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int co = ((q + 4 * t) ^ swz) << 4;
            u32x4_t xf[MT], wf[NT];
#pragma unroll
            for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(sx + b * 16 * 128 + co);
#pragma unroll
            for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co);
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int b = 0; b < MT; ++b) Mma<T>::run(wf[a], xf[b], acc[a][b]);
        }
    }

    // ---- epilogue: full-row stores through LDS when this wave's 64 channels are all stored, else direct 8/16-byte stores
    const int oh0 = d.ph_oh[ph], ow0 = d.ph_ow[ph];
    const bool vec_ok = ((d.Nstore & 3) == 0) && ((d.ldc & 3) == 0);
    const int nw0 = n_base + wn * WN;
    auto out_row = [&](int m) -> T* {                 // output pointer (channel 0) of GEMM row m, nullptr if not stored
        if (m >= M) return nullptr;
        const int jj = m % d.Mw, t = m / d.Mw, ii = t % d.Mh, bb = t / d.Mh;
        if (d.compact_out) return y + ((long)(bb * d.nphase + ph) * (d.Mh * d.Mw) + ii * d.Mw + jj) * d.ldc;
        const int ho = ii * d.so + oh0, wo = jj * d.so + ow0;
        if (ho >= d.Ho || wo >= d.Wo) return nullptr;
        return y + ((long)(bb * d.Ho + ho) * d.Wo + wo) * d.ldc;
    };
    if constexpr (MT == 4 && NT == 4) {
        __syncthreads();                                   // block-uniform: every wave is done reading the staging buffers
        if (vec_ok && nw0 + 64 <= d.Nstore && (d.ldc * (int)sizeof(T)) % 16 == 0) {      // wave-uniform choice
            float b4[NT * 4];
#pragma unroll
            for (int a = 0; a < NT; ++a)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int n = nw0 + a * 16 + 4 * q + e;
                    b4[a * 4 + e] = (bias != nullptr && n < d.Nrows) ? bias[n] : 0.f;
                }
            const int mw = m_base + wm * WM;
            float* so = nullptr;
            if (d.in_partial != nullptr && mw < M) {      // the wave's 64 rows lie in one image (Mh*Mw % 64 == 0, host-checked)
                const int per = d.Mh * d.Mw, im = mw / per, loc = mw - im * per;
                so = d.in_partial + (((long)im * (d.nphase * (per / 64)) + ph * (per / 64) + loc / 64) * d.Nstore + nw0) * 2;
            }
            // the store loop asks for rows mw + r0 + RPI * i in increasing order: coordinates carried from one row to the next
            // (out_row's four divisions per row and lane were a third of the epilogue)
            int cj = 0, ci = 0, cb = 0, cm = -1;
            auto row_walk = [&](int r) -> T* {
                const int m = mw + r;
                if (m >= M) return nullptr;
                if (cm < 0 || m < cm) { cj = m % d.Mw; const int t = m / d.Mw; ci = t % d.Mh; cb = t / d.Mh; }
                else { cj += m - cm; while (cj >= d.Mw) { cj -= d.Mw; if (++ci == d.Mh) { ci = 0; ++cb; } } }
                cm = m;
                T* pp;
                if (d.compact_out) pp = y + ((long)(cb * d.nphase + ph) * (d.Mh * d.Mw) + ci * d.Mw + cj) * d.ldc;
                else {
                    const int ho = ci * d.so + oh0, wo = cj * d.so + ow0;
                    if (ho >= d.Ho || wo >= d.Wo) return nullptr;
                    pp = y + ((long)(cb * d.Ho + ho) * d.Wo + wo) * d.ldc;
                }
                return pp + nw0;
            };
            store_tile_via_lds<T, MT, NT>(acc, smem + wave * (64 * 64 * (int)sizeof(T)), lane, b4, d.act, d.slope, row_walk, so, min(64, M - mw));
            return;
        }
    }
#pragma unroll
    for (int b = 0; b < MT; ++b) {
        T* yp = out_row(m_base + wm * WM + b * 16 + l16);
        if (yp == nullptr) continue;
#pragma unroll
        for (int a = 0; a < NT; ++a) {
            const int n = n_base + wn * WN + a * 16 + 4 * q;
            if (n >= d.Nstore) continue;
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float bv = (bias != nullptr && n + e < d.Nrows) ? bias[n + e] : 0.f;
                v[e] = apply_act(acc[a][b][e] + bv, d.act, d.slope);
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
}

// explicit instantiations (host stubs + device code for every tile configuration the dispatcher uses)
#define UIG_INST(T, BM, BN, WMV, WNV, NS) \
    template __global__ void igemm_kernel<T, BM, BN, WMV, WNV, NS, true>(const T*, const T*, const float*, T*, const GatherDesc); \
    template __global__ void igemm_kernel<T, BM, BN, WMV, WNV, NS, false>(const T*, const T*, const float*, T*, const GatherDesc);
UIG_INST(bf16_t, 256, 16, 4, 1, 2) UIG_INST(bf16_t, 128, 64, 2, 2, 2) UIG_INST(bf16_t, 128, 128, 2, 2, 2) UIG_INST(bf16_t, 128, 256, 2, 4, 3) UIG_INST(bf16_t, 128, 128, 2, 2, 3)
UIG_INST(float, 256, 16, 4, 1, 2) UIG_INST(float, 128, 64, 2, 2, 2) UIG_INST(float, 128, 128, 2, 2, 2) UIG_INST(float, 128, 256, 2, 4, 3) UIG_INST(float, 128, 128, 2, 2, 3)
#undef UIG_INST

// ------------------------------------------------------------------------------------------------ host side
template <typename T, int BM, int BN, int WAVES_M, int WAVES_N, int NSTAGE, bool SMALL>
static int launch_igemm(const void* x, const void* wp, const float* bias, void* y, const GatherDesc& d, hipStream_t s) {
    const int M = d.B * d.Mh * d.Mw;
    const int mt = d.wp2 != nullptr ? (d.group_rows + BM - 1) / BM + (M - d.group_rows + BM - 1) / BM : (M + BM - 1) / BM;
    const int nt = (d.Nrows + BN - 1) / BN;
    const size_t smem = NSTAGE * (size_t)(BM + BN) * 128;
    auto kern = igemm_kernel<T, BM, BN, WAVES_M, WAVES_N, NSTAGE, SMALL>;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(kern), (size_t)(int)smem);
        if (e != hipSuccess) return uig_set_error((int)e, "igemm: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(kern, dim3(mt * nt, d.nphase), dim3(64 * WAVES_M * WAVES_N), smem, s,
                       (const T*)x, (const T*)wp, bias, (T*)y, d);
    UIG_LAUNCH_CHECK("uig_conv_gather");
    return 0;
}

static int g_force_tile = 0;   // testing / tuning hook: 0 = auto, 128 / 256 = force that BN for wide layers
extern "C" void uig_debug_set_tile(int bn) { g_force_tile = bn; }

template <typename T>
static int dispatch_igemm(const void* x, const void* wp, const float* bias, void* y, const GatherDesc& d, hipStream_t s) {
    constexpr int BK = 8 * ElemTraits<T>::E;
    const bool small = d.Cin < BK || (d.Cin % BK) != 0;
    if (d.Nrows <= 16) {
        return small ? launch_igemm<T, 256, 16, 4, 1, 2, true>(x, wp, bias, y, d, s)
                     : launch_igemm<T, 256, 16, 4, 1, 2, false>(x, wp, bias, y, d, s);
    } else if (d.Nrows <= 64) {
        return small ? launch_igemm<T, 128, 64, 2, 2, 2, true>(x, wp, bias, y, d, s)
                     : launch_igemm<T, 128, 64, 2, 2, 2, false>(x, wp, bias, y, d, s);
    }
    // Measured on MI355X (ResBlock conv, batch 8): 128x128 / 4 waves / 2 blocks per CU = 701 TF, 128x256 / 8 waves / 3-stage
    // ring = 683 TF (both limited by the per-CU L2->LDS path), so the narrow tile stays the default.
    // Few 128x128 tiles (< 1.5 per CU) with a long reduction (the PatchGAN's 4x4 256->512 layer: 256 tiles x 128 K-steps) run one
    // latency-bound block per CU; half-width tiles double the blocks so that two co-resident blocks cover each other's
    // waits.  Not with fused InstanceNorm statistics (those need the 64x64 wave tile's LDS epilogue).
    {
        const long mt = ((long)d.B * d.Mh * d.Mw + 127) / 128;
        const long tiles = mt * ((d.Nrows + 127) / 128) * d.nphase;
        const int ksteps = (d.ph_tap0[1] - d.ph_tap0[0]) * (d.Cin / BK);
        // The reflection-border GEMM (8 phases, 12 K-steps, 128 full-width tiles at batch 16) is bound by its column phases'
        // 32-KB-strided rows, not by the K loop (3-, 4- and 6-stage rings measure the same): half-width tiles put a block on
        // every CU, 16.8 -> 11.8 us.
        const bool border = d.compact_out && d.nphase == 8 && tiles <= 256;
        if (!small && d.in_partial == nullptr &&
            (g_force_tile == 64 || (g_force_tile == 0 && ((tiles < 384 && ksteps >= 32 && d.nphase == 1) || border))))
            return launch_igemm<T, 128, 64, 2, 2, 2, false>(x, wp, bias, y, d, s);
    }
    const bool wide = g_force_tile == 256;
    if (wide)      // full-width tile: the im2col tile is staged once per pixel tile, 8 waves, 3-stage DMA ring
        return small ? launch_igemm<T, 128, 256, 2, 4, 3, true>(x, wp, bias, y, d, s)
                     : launch_igemm<T, 128, 256, 2, 4, 3, false>(x, wp, bias, y, d, s);
    // 3-stage ring with counted vmcnt for the 128x128 tile (two tiles in flight, 96 KB of LDS): measured no better than the
    // 2-stage form even on small, latency-bound grids (border GEMM 17.2 vs 16.1 us, D4 conv 54.8 vs 52.5 us): hook only.
    if (!small && g_force_tile == 3)
        return launch_igemm<T, 128, 128, 2, 2, 3, false>(x, wp, bias, y, d, s);
    return small ? launch_igemm<T, 128, 128, 2, 2, 2, true>(x, wp, bias, y, d, s)
                 : launch_igemm<T, 128, 128, 2, 2, 2, false>(x, wp, bias, y, d, s);
}

int uig_try_conv_strip(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                       float* in_partial, const void* border_add, const void* res_add, void* y, int B, int H, int W, int Cin, int Nrows,
                       int k, int pad_mode, const int* taps, int ntaps, int dh_min, int dh_max, int Ho, int Wo, int ldc, int Nstore,
                       int act, float slope, int dtype, long x_bytes, long w_bytes, hipStream_t s, int* rc_out, UigBst* bst, int mirror, UigFinReq* fin);
extern "C" int uig_instnorm_finalize(const float* partial, int nslab, float* stats, int B, int64_t HW, int C, float eps, void* stream);
int uig_instnorm_finalize_bwd(const float* partial, int nslab, float* gm, int B, int64_t HW, int C, void* stream);      // instnorm.hip

int uig_try_conv_cin8(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                      void* y, int B, int H, int W, int Cin, int Nrows, int pad_mode, const int* taps, int ntaps,
                      int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype, long w_bytes, hipStream_t s, int* rc_out);
int uig_try_conv_gemv(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                      void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad, int pad_mode,
                      int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype, hipStream_t s, int* rc_out);
int uig_try_conv_rowstrip(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                          void* y, int B, int H, int W, int Cin, int Nrows, int k,
                          int pad_mode, const int* taps, int ntaps, int Ho, int Wo, int ldc, int Nstore, int act, float slope,
                          int dtype, long x_bytes, long w_bytes, hipStream_t s, int* rc_out);

int uig_try_conv_tr2(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                     float* in_partial, void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                     const int* ph_tap0, const int* taps, int Ho, int Wo, int ldc, int Nstore, int act, float slope, int dtype,
                     long x_bytes, long w_bytes, hipStream_t s, int* rc_out);

static int floordiv(int a, int b) { return (a >= 0) ? a / b : -((-a + b - 1) / b); }

static int conv_gather_impl2(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                             float* in_partial, const void* border_add, const void* res_add, void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                             int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                             int act, float slope, int dtype, void* stream, UigBst* bst, int mirror, UigFinReq* fin);

// fin / bst->gm (round 4): the launch's statistics are wanted FINAL.  The kernel family that can finalises them inside its launch
// (arrival tickets) and says so in `done`; for the rest the finalize launch runs here, behind the convolution - same bits either way.
static int conv_gather_impl(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                            float* in_partial, const void* border_add, const void* res_add, void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                            int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                            int act, float slope, int dtype, void* stream, UigBst* bst = nullptr, int mirror = 0, UigFinReq* fin = nullptr) {
    if (fin != nullptr) { UIG_CHECK_ARG(in_partial && fin->stats, "uig_conv_gather_fin: in_partial and in_stats go together"); fin->done = 0; }
    if (bst != nullptr) bst->done = 0;
    int rc = conv_gather_impl2(x, wp, bias, wp2, bias2, group_images, in_partial, border_add, res_add, y, B, H, W, Cin, Nrows, kH, kW, stride, pad,
                               pad_mode, gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream, bst, mirror, fin);
    if (rc) return rc;
    if (fin != nullptr && !fin->done) {
        rc = uig_instnorm_finalize(in_partial, Ho * Wo / 64, fin->stats, B, (int64_t)Ho * Wo, Nstore, fin->eps, stream);
        if (rc) return rc;
    }
    if (bst != nullptr && bst->gm != nullptr && !bst->done)
        rc = uig_instnorm_finalize_bwd(bst->partial, Ho * Wo / 64, bst->gm, B, (int64_t)Ho * Wo, Nstore, stream);
    return rc;
}

static int conv_gather_impl2(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2, int group_images,
                             float* in_partial, const void* border_add, const void* res_add, void* y, int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                             int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                             int act, float slope, int dtype, void* stream, UigBst* bst, int mirror, UigFinReq* fin) {
    UIG_CHECK_ARG(x && wp && y, "uig_conv_gather: null pointer");
    if (wp2 != nullptr) UIG_CHECK_ARG(group_images > 0 && group_images < B, "uig_conv_gather_pair: group_images=%d must be in (0, B=%d)", group_images, B);
    UIG_CHECK_ARG(B > 0 && H > 0 && W > 0 && Ho > 0 && Wo > 0, "uig_conv_gather: bad shape B=%d H=%d W=%d Ho=%d Wo=%d", B, H, W, Ho, Wo);
    UIG_CHECK_ARG(Cin >= 8 && Cin % 8 == 0, "uig_conv_gather: Cin=%d must be a multiple of 8 (pad channels)", Cin);
    UIG_CHECK_ARG(kH >= 1 && kW >= 1 && kH * kW <= 64 && kH <= 8 && kW <= 8, "uig_conv_gather: unsupported kernel %dx%d", kH, kW);
    UIG_CHECK_ARG(stride == 1 || stride == 2, "uig_conv_gather: stride=%d unsupported", stride);
    UIG_CHECK_ARG(Nrows >= 1 && Nstore >= 1 && Nstore <= ldc, "uig_conv_gather: bad N (Nrows=%d Nstore=%d ldc=%d)", Nrows, Nstore, ldc);
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_conv_gather: bad dtype %d", dtype);
    const long esz = dtype == UIG_BF16 ? 2 : 4;
    UIG_CHECK_ARG((long)B * H * W * Cin * esz < (1L << 32) - 64 && (long)B * Ho * Wo * ldc < (1L << 31) && (long)Nrows * kH * kW * Cin * esz < (1L << 32) - 64,
                  "uig_conv_gather: tensor too large for 32-bit byte offsets");
    if (pad_mode == UIG_PAD_REFLECT)
        UIG_CHECK_ARG(gather_mode == UIG_GATHER_DIRECT && pad < H && pad < W, "uig_conv_gather: reflect pad needs direct mode and pad < dim");
    const int BKe = (dtype == UIG_BF16) ? 64 : 32;
    if (Cin < BKe || Cin % BKe) UIG_CHECK_ARG((Cin & (Cin - 1)) == 0, "uig_conv_gather: small Cin=%d must be a power of two", Cin);

    GatherDesc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = Cin; d.pad_mode = pad_mode;
    d.Nrows = Nrows; d.ldw = kH * kW * Cin; d.Ho = Ho; d.Wo = Wo; d.ldc = ldc; d.Nstore = Nstore;
    d.act = act; d.slope = slope;
    d.cin_shift = 0; while ((1 << d.cin_shift) < Cin) ++d.cin_shift;
    d.x_bytes = (unsigned)((long)B * H * W * Cin * esz); d.w_bytes = (unsigned)((long)Nrows * kH * kW * Cin * esz);
    if (gather_mode == UIG_GATHER_DIRECT) {
        UIG_CHECK_ARG(Ho == (H + 2 * pad - kH) / stride + 1 && Wo == (W + 2 * pad - kW) / stride + 1,
                      "uig_conv_gather: direct output %dx%d does not match input %dx%d k=%dx%d s=%d p=%d", Ho, Wo, H, W, kH, kW, stride, pad);
        d.nphase = 1; d.si = stride; d.so = 1; d.Mh = Ho; d.Mw = Wo;
        d.ph_tap0[0] = 0; d.ph_tap0[1] = kH * kW; d.ph_oh[0] = 0; d.ph_ow[0] = 0;
        for (int kh = 0; kh < kH; ++kh)
            for (int kw = 0; kw < kW; ++kw) {
                const int t = kh * kW + kw;
                d.tap[t] = ((kh - pad) + 128) | (((kw - pad) + 128) << 8) | (t << 16);
            }
    } else {
        UIG_CHECK_ARG(gather_mode == UIG_GATHER_TRANSPOSED, "uig_conv_gather: bad gather_mode %d", gather_mode);
        UIG_CHECK_ARG(Ho <= (H - 1) * stride - 2 * pad + kH + (stride - 1) && Wo <= (W - 1) * stride - 2 * pad + kW + (stride - 1),
                      "uig_conv_gather: transposed output %dx%d too large for input %dx%d k=%dx%d s=%d p=%d", Ho, Wo, H, W, kH, kW, stride, pad);
        d.nphase = stride * stride; d.si = 1; d.so = stride;
        d.Mh = (Ho + stride - 1) / stride; d.Mw = (Wo + stride - 1) / stride;
        int nt = 0;
        for (int a = 0; a < stride; ++a)
            for (int b = 0; b < stride; ++b) {
                const int p = a * stride + b;
                d.ph_tap0[p] = nt; d.ph_oh[p] = (signed char)a; d.ph_ow[p] = (signed char)b;
                for (int kh = 0; kh < kH; ++kh) {
                    if (((a + pad - kh) % stride + stride) % stride) continue;
                    for (int kw = 0; kw < kW; ++kw) {
                        if (((b + pad - kw) % stride + stride) % stride) continue;
                        d.tap[nt] = (floordiv(a + pad - kh, stride) + 128) | ((floordiv(b + pad - kw, stride) + 128) << 8) | ((kh * kW + kw) << 16);
                        ++nt;
                    }
                }
            }
        d.ph_tap0[d.nphase] = nt;
        for (int p = 0; p < d.nphase; ++p)
            UIG_CHECK_ARG(d.ph_tap0[p + 1] > d.ph_tap0[p], "uig_conv_gather: transposed phase %d has no taps (k=%dx%d s=%d p=%d)", p, kH, kW, stride, pad);
    }
    if (in_partial != nullptr) {     // fused InstanceNorm statistics need the full-row LDS epilogue on every wave
        const bool tr2 = gather_mode == UIG_GATHER_TRANSPOSED && kH == 3 && kW == 3 && stride == 2 && pad == 1 && Ho == 2 * H && Wo == 2 * W &&
                         uig_conv_tr2_applicable(B, H, W, Cin, Nrows, Nstore, ldc, dtype) == 1;      // 64-channel layers: only on the phase-fused kernel
        UIG_CHECK_ARG((Nrows > 64 || tr2) && Nrows % 64 == 0 && Nstore == Nrows && (d.Mh * d.Mw) % 64 == 0 && Ho % d.so == 0 && Wo % d.so == 0 &&
                      (ldc * (dtype == UIG_BF16 ? 2 : 4)) % 16 == 0 && g_force_tile == 0,
                      "uig_conv_gather_ex: fused IN statistics unsupported for this shape (N=%d, %dx%d)", Nrows, d.Mh, d.Mw);
        d.in_partial = in_partial;
    }
    hipStream_t s = (hipStream_t)stream;
    if (gather_mode == UIG_GATHER_TRANSPOSED && border_add == nullptr && res_add == nullptr) {   // stride-2 3x3: all four phases of a tile in one block
        int rc = 0;
        if (uig_try_conv_tr2(x, wp, bias, wp2, bias2, group_images, in_partial, y, B, H, W, Cin, Nrows, kH, kW, stride, pad, d.ph_tap0, d.tap,
                             Ho, Wo, ldc, Nstore, act, slope, dtype, (long)d.x_bytes, (long)d.w_bytes, s, &rc))
            return rc;
    }
    if (gather_mode == UIG_GATHER_DIRECT && in_partial == nullptr && border_add == nullptr && res_add == nullptr) {   // 1..4 output channels, wide input: one wave per dot product
        int rc = 0;
        if (uig_try_conv_gemv(x, wp, bias, wp2, bias2, group_images, y, B, H, W, Cin, Nrows, kH, kW, stride, pad, pad_mode, Ho, Wo, ldc, Nstore,
                              act, slope, dtype, s, &rc))
            return rc;
    }
    if (d.nphase == 1 && stride == 1 && Cin == 8 && in_partial == nullptr && border_add == nullptr && res_add == nullptr) {   // one chunk of input channels per pixel
        int rc = 0;
        if (uig_try_conv_cin8(x, wp, bias, wp2, bias2, group_images, y, B, H, W, Cin, Nrows, pad_mode, d.tap, d.ph_tap0[1], Ho, Wo, ldc, Nstore,
                              act, slope, dtype, (long)d.w_bytes, s, &rc))
            return rc;
    }
    if (d.nphase == 1 && stride == 1 && kH == kW && in_partial == nullptr && border_add == nullptr && res_add == nullptr) {   // few output channels, many taps
        int rc = 0;
        if (uig_try_conv_rowstrip(x, wp, bias, wp2, bias2, group_images, y, B, H, W, Cin, Nrows, kH, pad_mode, d.tap, kH * kW, Ho, Wo,
                                  ldc, Nstore, act, slope, dtype, (long)d.x_bytes, (long)d.w_bytes, s, &rc))
            return rc;
    }
    if (d.nphase == 1 && stride == 1 && kH == kW) {      // stride-1 k x k: LDS-resident input strip kernel (conv_strip.hip)
        int dmin = 127, dmax = -127, rc = 0;
        for (int t = 0; t < kH * kW; ++t) { const int dh = (d.tap[t] & 255) - 128; dmin = std::min(dmin, dh); dmax = std::max(dmax, dh); }
        if (uig_try_conv_strip(x, wp, bias, wp2, bias2, group_images, in_partial, border_add, res_add, y, B, H, W, Cin, Nrows, kH, pad_mode, d.tap, kH * kW, dmin, dmax,
                               Ho, Wo, ldc, Nstore, act, slope, dtype, (long)d.x_bytes, (long)d.w_bytes, s, &rc, bst, mirror, fin))
            return rc;
    }
    UIG_CHECK_ARG(mirror == 0, "uig_reflect3x3_dgrad_mirror: shape not taken by the persistent strip kernel (query uig_reflect3x3_dgrad_mirror_applicable)");
    UIG_CHECK_ARG(bst == nullptr, "uig_conv_gather_bst: the fused InstanceNorm-backward statistics need the bf16 strip kernel with border / residual terms");
    UIG_CHECK_ARG(border_add == nullptr && res_add == nullptr, "uig_conv_gather_ex: border_add / res_add need the stride-1 3x3 strip kernel (query uig_conv_strip_applicable)");
    if (wp2 != nullptr) {
        const long grows = (long)group_images * d.Mh * d.Mw;
        if (grows % 256 != 0 && in_partial != nullptr) {      // fused-statistics slabs assume 64-row alignment of both groups: two launches (same results)
            const long esz2 = dtype == UIG_BF16 ? 2 : 4;
            const long pstride = (long)(Ho * Wo / 64) * Nstore * 2;
            int rc = conv_gather_impl2(x, wp, bias, nullptr, nullptr, 0, in_partial, nullptr, nullptr, y, group_images, H, W, Cin, Nrows, kH, kW, stride, pad, pad_mode,
                                       gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream, nullptr, 0, nullptr);
            if (rc) return rc;
            return conv_gather_impl2((const char*)x + (long)group_images * H * W * Cin * esz2, wp2, bias2, nullptr, nullptr, 0,
                                     in_partial ? in_partial + group_images * pstride : nullptr, nullptr, nullptr, (char*)y + (long)group_images * Ho * Wo * ldc * esz2, B - group_images, H, W, Cin, Nrows, kH, kW,
                                     stride, pad, pad_mode, gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream, nullptr, 0, nullptr);
        }
        d.wp2 = wp2; d.bias2 = bias2; d.group_rows = (int)grows;
    }
    uig_note_conv_kernel(UIG_K_IGEMM);
    return dtype == UIG_BF16 ? dispatch_igemm<bf16_t>(x, wp, bias, y, d, s) : dispatch_igemm<float>(x, wp, bias, y, d, s);
}

extern "C" int uig_conv_gather(const void* x, const void* wp, const float* bias, void* y,
                               int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                               int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                               int act, float slope, int dtype, void* stream) {
    return conv_gather_impl(x, wp, bias, nullptr, nullptr, 0, nullptr, nullptr, nullptr, y, B, H, W, Cin, Nrows, kH, kW, stride, pad, pad_mode, gather_mode,
                            Ho, Wo, ldc, Nstore, act, slope, dtype, stream);
}

extern "C" int uig_conv_gather_pair(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                                    int group_images, void* y,
                                    int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                                    int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                                    int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(wp2 != nullptr, "uig_conv_gather_pair: null wp2");
    return conv_gather_impl(x, wp, bias, wp2, bias2, group_images, nullptr, nullptr, nullptr, y, B, H, W, Cin, Nrows, kH, kW, stride, pad, pad_mode, gather_mode,
                            Ho, Wo, ldc, Nstore, act, slope, dtype, stream);
}

extern "C" int uig_conv_gather_ex(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                                  int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                                  int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                                  int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                                  int act, float slope, int dtype, void* stream) {
    return conv_gather_impl(x, wp, bias, wp2, bias2, wp2 ? group_images : 0, in_partial, border_add, res_add, y, B, H, W, Cin, Nrows, kH, kW, stride, pad,
                            pad_mode, gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream);
}


// uig_conv_gather_ex plus the statistics of the InstanceNorm BACKWARD that consumes this launch's output as its dy (see
// StripDesc::bst_*): bf16 strip-kernel launches with border_add and / or res_add only (3x3 stride-1 input gradients).
extern "C" int uig_conv_gather_bst(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                                   int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                                   int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                                   int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                                   int act, float slope, int dtype,
                                   const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial, void* stream) {
    UIG_CHECK_ARG(bst_x && bst_stats && bst_partial, "uig_conv_gather_bst: null statistics pointer");
    UigBst b{bst_x, bst_stats, bst_partial, bst_act, bst_slope, nullptr, nullptr, 0};
    return conv_gather_impl(x, wp, bias, wp2, bias2, wp2 ? group_images : 0, in_partial, border_add, res_add, y, B, H, W, Cin, Nrows, kH, kW, stride, pad,
                            pad_mode, gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream, &b);
}

// uig_conv_gather_ex whose fused InstanceNorm statistics come out FINAL (round 4): in_stats fp32[B][Nstore][2] = (mean, rstd) with
// eps = in_eps, bit-identical to uig_instnorm_finalize on in_partial.  Where the kernel family supports it the image's last-arriving
// block finalises inside the convolution launch (arrival tickets: `tickets` = >= B zero-initialised 32-bit words, left zero); elsewhere
// the finalize launch runs behind the convolution.  tickets == NULL: always the latter.
extern "C" int uig_conv_gather_fin(const void* x, const void* wp, const float* bias, const void* wp2, const float* bias2,
                                   int group_images, float* in_partial, const void* border_add, const void* res_add, void* y,
                                   int B, int H, int W, int Cin, int Nrows, int kH, int kW, int stride, int pad,
                                   int pad_mode, int gather_mode, int Ho, int Wo, int ldc, int Nstore,
                                   int act, float slope, int dtype, float* in_stats, float in_eps, unsigned* tickets, void* stream) {
    UIG_CHECK_ARG(in_partial && in_stats, "uig_conv_gather_fin: null statistics pointer");
    UigFinReq f{in_stats, in_eps, tickets, 0};
    return conv_gather_impl(x, wp, bias, wp2, bias2, wp2 ? group_images : 0, in_partial, border_add, res_add, y, B, H, W, Cin, Nrows, kH, kW, stride, pad,
                            pad_mode, gather_mode, Ho, Wo, ldc, Nstore, act, slope, dtype, stream, nullptr, 0, &f);
}

// Input gradient of a reflection-padded (pad 1) 3x3 stride-1 convolution in ONE launch: dx (B, H, W, ldc) = the zero-padded
// transposed convolution of dy (B, H, W, C) with the mirrored-border terms folded inside the persistent strip kernel (mirror pixels,
// conv_strip_pk.hip) [+ res_add, a tensor of dx's shape: the ResBlock skip gradient].  wp / wp2 + group_images as uig_conv_gather_pair
// (the packed TRANSPOSED-gather operands).  Only where uig_reflect3x3_dgrad_mirror_applicable says 1.
extern "C" int uig_reflect3x3_dgrad_mirror(const void* dy, const void* wp, const void* wp2, int group_images, const void* res_add, void* dx,
                                           int B, int H, int W, int C, int Nrows, int ldc, int dtype, void* stream) {
    UIG_CHECK_ARG(dy && wp && dx, "uig_reflect3x3_dgrad_mirror: null pointer");
    return conv_gather_impl(dy, wp, nullptr, wp2, nullptr, wp2 ? group_images : 0, nullptr, nullptr, res_add, dx, B, H, W, C, Nrows, 3, 3, 1, 1, UIG_PAD_ZERO,
                            UIG_GATHER_TRANSPOSED, H, W, ldc, ldc, UIG_ACT_NONE, 0.f, dtype, stream, nullptr, 1);
}

// uig_reflect3x3_dgrad_mirror that also produces the statistics of the InstanceNorm BACKWARD consuming dx as its dy (round 4; the
// arguments of uig_conv_gather_bst): bst_partial fp32[B][H*W/64][ldc][2] and, finalised (inside the launch through `tickets`, or by a
// finalize launch behind it when tickets == NULL), bst_gm fp32[B][ldc][2] = (mean g, mean g*xhat) for uig_instnorm_act_bwd_colsum_t.
extern "C" int uig_reflect3x3_dgrad_mirror_bst(const void* dy, const void* wp, const void* wp2, int group_images, const void* res_add, void* dx,
                                               int B, int H, int W, int C, int Nrows, int ldc, int dtype,
                                               const void* bst_x, const float* bst_stats, int bst_act, float bst_slope, float* bst_partial, float* bst_gm,
                                               unsigned* tickets, void* stream) {
    UIG_CHECK_ARG(dy && wp && dx && bst_x && bst_stats && bst_partial && bst_gm, "uig_reflect3x3_dgrad_mirror_bst: null pointer");
    UigBst b{bst_x, bst_stats, bst_partial, bst_act, bst_slope, bst_gm, tickets, 0};
    return conv_gather_impl(dy, wp, nullptr, wp2, nullptr, wp2 ? group_images : 0, nullptr, nullptr, res_add, dx, B, H, W, C, Nrows, 3, 3, 1, 1, UIG_PAD_ZERO,
                            UIG_GATHER_TRANSPOSED, H, W, ldc, ldc, UIG_ACT_NONE, 0.f, dtype, stream, &b, 1);
}

// ---------------------------------------------------------------------------------------------------------------
// Border terms of the input gradient of a reflection-padded (pad 1) 3x3 stride-1 convolution.
//   dx = zero-pad-1 transposed conv of dy   (exact 64x64-style grid: perfect tile rounds, conv_strip.hip)
//      + terms whose padded coordinate was mirrored:
//        T: dx[1][j]   += sum_kw dy[0][j+1-kw]   W[0][kw]      B: dx[H-2][j] += sum_kw dy[H-1][j+1-kw] W[2][kw]
//        L: dx[i][1]   += sum_kh dy[i+1-kh][0]   W[kh][0]      R: dx[i][W-2] += sum_kh dy[i+1-kh][W-1] W[kh][2]
//        corners: dx[1][1] += dy[0][0] W[0][0], dx[1][W-2] += dy[0][W-1] W[0][2], dx[H-2][1] += dy[H-1][0] W[2][0],
//                 dx[H-2][W-2] += dy[H-1][W-1] W[2][2]
// This launch computes those 8 groups as 8 phases of the generic gather kernel into the compact buffer
// bord[B][8][S][ldc] (S = H = W; corner phases use position 0 only); the strip kernel's epilogue adds them
// (border_add of uig_conv_gather_ex).  Replaces the (H+2)x(W+2) padded gradient + uig_reflect_fold.
extern "C" int uig_reflect3x3_dgrad_border(const void* dy, const void* wp, const void* wp2, int group_images, void* bord,
                                           int B, int H, int W, int C, int Nrows, int ldc, int dtype, void* stream) {
    UIG_CHECK_ARG(dy && wp && bord, "uig_reflect3x3_dgrad_border: null pointer");
    UIG_CHECK_ARG(H == W && H >= 4 && H <= 128, "uig_reflect3x3_dgrad_border: needs a square map of side 4..128 (got %dx%d)", H, W);
    UIG_CHECK_ARG(C % 8 == 0 && Nrows >= 1 && Nrows <= ldc, "uig_reflect3x3_dgrad_border: bad channels C=%d Nrows=%d ldc=%d", C, Nrows, ldc);
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_reflect3x3_dgrad_border: bad dtype");
    const int BKe = dtype == UIG_BF16 ? 64 : 32;
    UIG_CHECK_ARG(C % BKe == 0, "uig_reflect3x3_dgrad_border: C=%d must be a multiple of %d", C, BKe);
    const long esz = dtype == UIG_BF16 ? 2 : 4;
    GatherDesc d{};
    d.B = B; d.H = H; d.W = W; d.Cin = C; d.pad_mode = UIG_PAD_ZERO;
    d.Nrows = Nrows; d.ldw = 9 * C; d.Ho = 1; d.Wo = 1; d.ldc = ldc; d.Nstore = ldc; d.act = UIG_ACT_NONE; d.slope = 0.f;
    d.cin_shift = 0; while ((1 << d.cin_shift) < C) ++d.cin_shift;
    d.x_bytes = (unsigned)((long)B * H * W * C * esz); d.w_bytes = (unsigned)((long)Nrows * 9 * C * esz);
    d.nphase = 8; d.si = 1; d.so = 1; d.Mh = 1; d.Mw = H; d.compact_out = 1;
    auto tap = [](int dh, int dw, int wt) { return (dh + 128) | ((dw + 128) << 8) | (wt << 16); };
    int nt = 0;
    // T, B (position = column)
    d.ph_tap0[0] = nt; for (int kw = 0; kw < 3; ++kw) d.tap[nt++] = tap(0, 1 - kw, 0 * 3 + kw);
    d.ph_tap0[1] = nt; for (int kw = 0; kw < 3; ++kw) d.tap[nt++] = tap(H - 1, 1 - kw, 2 * 3 + kw);
    // L, R (position = row: swapped iteration)
    d.ph_tap0[2] = nt; d.ph_swap[2] = 1; for (int kh = 0; kh < 3; ++kh) d.tap[nt++] = tap(1 - kh, 0, kh * 3 + 0);
    d.ph_tap0[3] = nt; d.ph_swap[3] = 1; for (int kh = 0; kh < 3; ++kh) d.tap[nt++] = tap(1 - kh, W - 1, kh * 3 + 2);
    // corners (position 0 is the value; other positions are never read)
    d.ph_tap0[4] = nt; d.tap[nt++] = tap(0, 0, 0);
    d.ph_tap0[5] = nt; d.tap[nt++] = tap(0, W - 1, 2);
    d.ph_tap0[6] = nt; d.tap[nt++] = tap(H - 1, 0, 6);
    d.ph_tap0[7] = nt; d.tap[nt++] = tap(H - 1, W - 1, 8);
    d.ph_tap0[8] = nt;
    if (wp2 != nullptr) {
        UIG_CHECK_ARG(group_images > 0 && group_images < B, "uig_reflect3x3_dgrad_border: bad group_images");
        const long grows = (long)group_images * H;
        if (grows % 256 != 0) {       // tiles could straddle the groups: two launches
            int rc = uig_reflect3x3_dgrad_border(dy, wp, nullptr, 0, bord, group_images, H, W, C, Nrows, ldc, dtype, stream);
            if (rc) return rc;
            return uig_reflect3x3_dgrad_border((const char*)dy + (long)group_images * H * W * C * esz, wp2, nullptr, 0,
                                               (char*)bord + (long)group_images * 8 * H * ldc * esz, B - group_images, H, W, C, Nrows,
                                               ldc, dtype, stream);
        }
        d.wp2 = wp2; d.bias2 = nullptr; d.group_rows = (int)grows;
    }
    hipStream_t s = (hipStream_t)stream;
    return dtype == UIG_BF16 ? dispatch_igemm<bf16_t>(dy, wp, nullptr, bord, d, s) : dispatch_igemm<float>(dy, wp, nullptr, bord, d, s);
}
