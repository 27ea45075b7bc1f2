// conv_strip_desc.h — launch descriptor and MFMA wrapper shared by the strip-convolution kernels (conv_strip.hip, conv_strip_pk.hip).
#pragma once
#include "uig_common.h"

struct StripDesc {
    int B, H, W, Cin;        // input  (B, H, W, Cin)
    int Ho, Wo;              // output (B, Ho, Wo, ldc): one GEMM row per output pixel
    int pad_mode;
    int dh_min, dh_max;
    int Nrows, ldw, ldc, Nstore;
    int act; float slope;
    unsigned x_bytes, w_bytes;
    int tap[16];             // (dh + 128) | (dw + 128) << 8 | weight-tap-index << 16
    const void* wp2;         // paired launch: images >= group_images use wp2 / bias2
    const float* bias2;
    int group_images;
    float* in_partial;       // optional: InstanceNorm partial statistics [img][HoWo/64][Nstore][2]
    const void* res_add;     // optional: tensor of y's shape added to the output in the epilogue (the ResBlock skip gradient in dgrad)
    const void* border_add;  // optional: bord[B][8][S][ldc] of uig_reflect3x3_dgrad_border, added to rows 1 / H-2 and cols 1 / W-2
    unsigned long long* dbg; // diagnostic build only (STAMP): per-wave cycle sums {wait+barrier, DMA issue, reads+MFMA, total}
    int wo_magic;            // persistent kernel: ceil(2^20 / Wo), set by uig_launch_strip_pk
    unsigned long long mg_ntn, mg_tpi, mg_wo;      // persistent kernel (round 4): ceil(2^32 / d) for d = channel tiles, tiles per image, Wo: n / d = (n * mg) >> 32 for n * d < 2^32 (get_tile's four divisions were ~1.4 k cycles of scalar reciprocal sequences in front of the block's first DMA)
    // optional: statistics of the InstanceNorm BACKWARD that consumes this launch's output as its dy (input-gradient launches):
    // bst_x = that norm's saved input (shape of y), bst_stats = its (mean, rstd) [B][ldc][2], bst_act / bst_slope its activation;
    // bst_partial [B][HoWo/64][Nstore][2] receives (sum g, sum g * xhat), g = dy * act'(xhat), per 64-pixel slab - what the
    // norm's own statistics pass (one read of dy and x) would have produced (uig_instnorm_act_bwd_colsum_pre consumes it)
    const void* bst_x;
    const float* bst_stats;
    float* bst_partial;
    int bst_act; float bst_slope;
    // persistent bf16 kernel only: 1 = this launch is the input gradient of a REFLECTION-padded 3x3 convolution on a 64-wide map and
    // the kernel folds the mirrored-border terms itself (conv_strip_pk.hip, "mirror pixels"): no border_add buffer, no border GEMM
    int mirror;
    // persistent bf16 kernel only (round 3): the input x is the RAW output of the convolution in front of an InstanceNorm(+ReLU /
    // LeakyReLU) and this launch applies that norm itself, on the strip as it sits in LDS (conv_strip_pk.hip, "norm strip"): no
    // apply pass between the two convolutions of a ResBlock.  nrm_stats = the norm's (mean, rstd) fp32[B][Cin][2] (the finalize
    // launch's output), nrm_act / nrm_slope its activation; nrm_h (optional): tensor of x's shape that receives the normalised
    // activations - what the apply pass would have written: the backward pass's weight-gradient operand.
    const float* nrm_stats;
    int nrm_act; float nrm_slope;
    void* nrm_h;
    int wt_a, wt_b;          // persistent kernel, lean phased schedule: weight-tap index of tap t = wt_a + wt_b * t (set by uig_launch_strip_pk)
    int need_rows;           // strip rows the worst 256-pixel tile needs (set by uig_try_conv_strip)
    int wide512;             // persistent bf16 kernel: 1 = the 512-row strip without zero rows (conv_strip_pk.hip, NOZ; set by uig_try_conv_strip)
    // round 4, persistent kernel: in-launch finalize (arrival tickets, uig_common.h) of the forward statistics (fin: in_partial -> (mean, rstd))
    // and of the norm-backward statistics (bfin: bst_partial -> (mean g, mean g*xhat)); tickets == NULL = off (the caller runs the finalize launch)
    UigFin fin, bfin;
};


template <typename T> struct MmaS;
template <> struct MmaS<bf16_t> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
        c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
    }
};
template <> struct MmaS<float> {
    static __device__ __forceinline__ void run(const u32x4_t& a, const u32x4_t& b, f32x4_t& c) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a[e]), __uint_as_float(b[e]), c, 0, 0, 0);
    }
};


// Accumulator initialisation of a strip tile: the bias rides in the accumulators from the start (acc = bias + sum of
// products, one fp32 rounding order for every strip kernel), so the epilogue carries no bias registers: its 16 scalar loads
// per lane were the first thing the register allocator spilled, one `s_waitcnt vmcnt(0)` each, in the persistent kernel.
template <int MT, int NT, int WN>
__device__ __forceinline__ void strip_init_acc(f32x4_t (&acc)[NT][MT], const float* __restrict__ bias, int Nrows, int n_base, int wn, int lane) {
    const int q = lane >> 4;
#pragma unroll
    for (int a = 0; a < NT; ++a) {
        const int n = n_base + wn * WN + a * 16 + 4 * q;         // Nrows is a multiple of 128 for every strip launch: n + 3 < Nrows
        const f32x4_t bv = (bias != nullptr && n < Nrows) ? *reinterpret_cast<const f32x4_t*>(bias + n) : f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int b = 0; b < MT; ++b) acc[a][b] = bv;
    }
}

// Epilogue of one BM x BN strip tile, shared by conv_strip.hip and conv_strip_pk.hip: activation, optional reflection-border /
// residual adds, optional InstanceNorm partial statistics; full-row stores through this wave's 64x64 LDS scratch when its 64
// channels are all stored, else direct 8/16-byte stores.  The bias is already in the accumulators (strip_init_acc).
// The caller has made sure (barrier) that `scratch` is free.
// BSTE: compile the norm-backward statistics (d.bst_*) in - always with the border form (BORD), an opt-in variant of the mirror-pixel kernel
template <typename T, int MT, int NT, int WM, int WN, bool BORD = true, typename EHook = NoMidHook, bool BSTE = BORD>
__device__ __forceinline__ void strip_epilogue(const f32x4_t (&acc)[NT][MT], unsigned char* scratch, const StripDesc& d, T* __restrict__ y,
                                               int img, int p0, int wm, int wn, int n_base, int lane, EHook ehook = EHook()) {
    const int HoWo = d.Ho * d.Wo;
    const int l16 = lane & 15, q = lane >> 4;
    const bool vec_ok = ((d.Nstore & 3) == 0) && ((d.ldc & 3) == 0);
    const int nw0 = n_base + wn * WN;
    // 64 x 16 wave tiles (NT == 1: the 64 x 64-tile kernel of very small grids, round 3) take the plain full-row form only: the host
    // never sends border / residual / norm-backward terms there
    if (MT == 4 && (NT == 4 || (NT == 1 && d.border_add == nullptr && d.res_add == nullptr)) && vec_ok && nw0 + WN <= d.Nstore &&
        (d.ldc * (int)sizeof(T)) % 16 == 0) {      // wave-uniform choice
        const int pw = p0 + wm * WM;
        T* ybase = y + (long)img * HoWo * d.ldc + nw0;
        float* so = nullptr;
        if (d.in_partial != nullptr && pw < HoWo)
            so = d.in_partial + (((long)img * ((HoWo + 63) / 64) + pw / 64) * d.Nstore + nw0) * 2;
        auto rowp = [&](int r) -> T* { const int p = pw + r; return p < HoWo ? ybase + (long)p * d.ldc : nullptr; };
        if (d.border_add == nullptr && d.res_add == nullptr && (NT != 4 || !BSTE || d.bst_partial == nullptr)) {
            store_tile_via_lds<T, MT, NT>(acc, scratch, lane, nullptr, d.act, d.slope, rowp, so, min(64, HoWo - pw), NoRowAdd(), ehook, d.fin.tickets != nullptr);   // ehook: diagnostic stamp
        } else if constexpr (NT == 4) {
            // reflection-pad dgrad: add the mirrored-border terms (phases T,B,L,R,TL,TR,BL,BR of the compact border buffer)
            constexpr int E = ElemTraits<T>::E;
            const int S = d.Ho;                                   // square map
            const T* bimg = static_cast<const T*>(d.border_add) + (long)img * 8 * S * d.ldc + nw0;
            if constexpr (sizeof(T) == 2) {
                // bf16: the border chunk and the residual chunk of this lane's 8 store rows are fetched NOW (range-checked buffer
                // loads: an out-of-range offset returns zeros, so no branches), and their latency hides behind the LDS
                // transposition; loading them inside the store loop cost the kernel 9-20 us.  A pixel has at most ONE border
                // term (its line's or its column's) except the four pixels (1|S-2, 1|S-2) of an image, which also take the
                // column and corner terms: those are loaded in the store loop by the lanes concerned (2 of 128 wave tiles of a
                // 64x64 map).  2 prefetched chunks per row instead of 4: 64 registers instead of 128 at the epilogue's peak.
                const bool hasb = BORD && d.border_add != nullptr, hasr = d.res_add != nullptr;      // !BORD: the caller never passes border_add
                const __amdgpu_buffer_rsrc_t rsb = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<void*>(hasb ? d.border_add : d.res_add), 0, hasb ? (unsigned)((long)d.B * 8 * S * d.ldc * (int)sizeof(T)) : 0u, 0x00020000);
                const __amdgpu_buffer_rsrc_t rsr = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<void*>(hasr ? d.res_add : d.border_add), 0, hasr ? (unsigned)((long)d.B * HoWo * d.ldc * (int)sizeof(T)) : 0u, 0x00020000);
                constexpr int NI = 8;                                  // 64 rows / 8 rows per store instruction
                // rows in flight: a ring of ND prefetch slots, refilled as rows are stored.  With border terms 2 and 4 measure the same
                // and 4 spills with the statistics path; without them (mirror-pixel kernel: one chunk per row) all 8 rows are fetched up
                // front - the ring of 2 left ~4 exposed round trips: 6.7k cycles per tile by the stamps
                constexpr int ND = BORD ? 2 : (BSTE ? 4 : 8);        // with the norm-backward statistics: a ring of 4 residual rows (registers)
                const int c = lane % 8, r0 = lane / 8;
                auto boff = [&](int phase, int pos) -> int {
                    return (int)(((((long)img * 8 + phase) * S + pos) * d.ldc + nw0 + c * E) * (long)sizeof(T));
                };
                // statistics of the InstanceNorm backward that takes this output as its dy (bst_*): the norm's saved input is
                // prefetched like the residual; (sum g, sum g * xhat) of the lane's 8 channels over its 8 rows, combined below
                const bool bst = BSTE && d.bst_partial != nullptr && pw < HoWo;         // wave-uniform (round 4: a variant of the mirror-pixel kernel carries them too)
                const __amdgpu_buffer_rsrc_t rsn = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<void*>(bst ? d.bst_x : (hasr ? d.res_add : d.border_add)), 0, bst ? (unsigned)((long)d.B * HoWo * d.ldc * (int)sizeof(T)) : 0u, 0x00020000);
                // ring of the norm's-input chunks (bst_* statistics).  Mirror-pixel kernel: all 8 rows are requested BEFORE the LDS transposition
                // (the fragment registers of the K loop are free by then, the accumulators still live): the tensor comes from HBM (the forward
                // pass wrote it long ago) and nothing else of the epilogue can hide that round trip
                constexpr int NDX = BORD ? 2 : 8;
                u32x4_t pre[ND][BORD ? 2 : 1], prex[NDX];              // [border term (line or column),] residual tensor; the norm's input
                constexpr int PR = BORD ? 1 : 0;                       // slot of the residual chunk
                unsigned both = 0;                                     // bit i: row i is one of the image's four double-border pixels
                auto fetch = [&](int i) {                              // row i -> slot i % ND (static after unrolling)
                    const int p = pw + r0 + 8 * i;
                    const int h = p / d.Wo, w = p - h * d.Wo;
                    const bool ok = p < HoWo;
                    if constexpr (BORD) {
                        const bool okb = ok & hasb;
                        const bool rt_ = okb & (h == 1), rb_ = okb & (h == S - 2), cl_ = okb & (w == 1), cr_ = okb & (w == S - 2);
                        const int o0 = (rt_ | rb_) ? boff(rt_ ? 0 : 1, w) : ((cl_ | cr_) ? boff(cl_ ? 2 : 3, h) : -1);
                        pre[i % ND][0] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsb, o0, 0, 0));
                        both |= (((rt_ | rb_) & (cl_ | cr_)) ? 1u : 0u) << i;
                    }
                    pre[i % ND][PR] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                        rsr, ok ? (int)((((long)img * HoWo + p) * d.ldc + nw0 + c * E) * (long)sizeof(T)) : -1, 0, 0));
                };
                auto fetch_x = [&](int i) {
                    const int p = pw + r0 + 8 * i;
                    prex[i % NDX] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                        rsn, (bst & (p < HoWo)) ? (int)((((long)img * HoWo + p) * d.ldc + nw0 + c * E) * (long)sizeof(T)) : -1, 0, 0));
                };
                constexpr int ND0 = 2;                                 // fetched NOW, behind the LDS transposition (accumulators still live); the rest in mid()
#pragma unroll
                for (int i = 0; i < ND0; ++i) fetch(i);
                if constexpr (!BORD && BSTE) {
                    if (bst) {
#pragma unroll
                        for (int i = 0; i < NDX; ++i) fetch_x(i);
                    }
                }
                float bmu[E], brs[E], bs1[E], bs2[E];
#pragma unroll
                for (int e = 0; e < E; ++e) { bmu[e] = 0.f; brs[e] = 0.f; bs1[e] = 0.f; bs2[e] = 0.f; }
                auto mid = [&]() {                                     // the accumulators are dead here
                    ehook();
#pragma unroll
                    for (int i = ND0; i < ND; ++i) fetch(i);
                    if (!bst) return;
                    const float* sp = d.bst_stats + ((long)img * d.ldc + nw0 + c * E) * 2;
#pragma unroll
                    for (int e = 0; e < E; e += 2) {
                        const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(sp + 2 * e);
                        bmu[e] = t4[0]; brs[e] = t4[1]; bmu[e + 1] = t4[2]; brs[e + 1] = t4[3];
                    }
                    if constexpr (BORD) {
#pragma unroll
                        for (int i = 0; i < NDX; ++i) fetch_x(i);
                    }
                };
                auto add = [&](int r, int, const u32x4_t& v, int i) -> u32x4_t {
                    float f[E], g[E];
                    chunk_to_f32<T>(v, f);
#pragma unroll
                    for (int k = 0; k < (BORD ? 2 : 1); ++k) {
                        chunk_to_f32<T>(pre[i % ND][k], g);
#pragma unroll
                        for (int e = 0; e < E; ++e) f[e] += g[e];
                    }
                    if (BORD && ((both >> i) & 1u)) {                  // rare: column term + corner term of a double-border pixel
                        const int p = pw + r;
                        const int h = p / d.Wo, w = p - h * d.Wo;
                        const u32x4_t cv = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsb, boff(w == 1 ? 2 : 3, h), 0, 0));
                        const u32x4_t kv = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(
                            rsb, boff((h == 1 ? 4 : 6) + (w == 1 ? 0 : 1), 0), 0, 0));
                        chunk_to_f32<T>(cv, g);
#pragma unroll
                        for (int e = 0; e < E; ++e) f[e] += g[e];
                        chunk_to_f32<T>(kv, g);
#pragma unroll
                        for (int e = 0; e < E; ++e) f[e] += g[e];
                    }
                    const u32x4_t outc = f32_to_chunk<T>(f);
                    if (bst) {                                         // on the values AS STORED (what the norm's own pass would read back)
                        float gv[E], xv[E];
                        chunk_to_f32<T>(outc, gv);
                        chunk_to_f32<T>(prex[i % NDX], xv);
                        const bool in = pw + r < HoWo;
#pragma unroll
                        for (int e = 0; e < E; ++e) {
                            const float xh = (xv[e] - bmu[e]) * brs[e];
                            float gg = gv[e];
                            if (d.bst_act == UIG_ACT_RELU) gg = xh > 0.f ? gg : 0.f;
                            else if (d.bst_act == UIG_ACT_LRELU) gg = xh > 0.f ? gg : gg * d.bst_slope;
                            gg = in ? gg : 0.f;
                            bs1[e] += gg; bs2[e] += gg * xh;
                        }
                    }
                    if (i + ND < NI) fetch(i + ND);                    // refill the slot just consumed
                    if (bst && i + NDX < NI) fetch_x(i + NDX);
                    return outc;
                };
                store_tile_via_lds<T, MT, NT>(acc, scratch, lane, nullptr, d.act, d.slope, rowp, so, min(64, HoWo - pw), add, mid, d.fin.tickets != nullptr);
                if (bst) {                                             // combine the 8 lanes that hold the same chunk; lanes 0-7 write
#pragma unroll
                    for (int o = 8; o < 64; o <<= 1)
#pragma unroll
                        for (int e = 0; e < E; ++e) { bs1[e] += __shfl_xor(bs1[e], o, 64); bs2[e] += __shfl_xor(bs2[e], o, 64); }
                    if (lane < 8) {
                        float* o = d.bst_partial + (((long)img * ((HoWo + 63) / 64) + pw / 64) * d.Nstore + nw0 + lane * E) * 2;
                        if (d.bfin.tickets != nullptr) {               // finalised inside this launch: write-through stores
#pragma unroll
                            for (int e = 0; e < E; ++e) uig_store8_sc1(o + 2 * e, bs1[e], bs2[e]);
                        } else {
#pragma unroll
                            for (int e = 0; e < E; e += 2) *reinterpret_cast<f32x4_t*>(o + 2 * e) = f32x4_t{bs1[e], bs2[e], bs1[e + 1], bs2[e + 1]};
                        }
                    }
                }
            } else {
            auto add = [&](int r, int c, const u32x4_t& v, int) -> u32x4_t {
                const int p = pw + r;
                if (p >= HoWo) return v;
                float f[E], g[E];
                chunk_to_f32<T>(v, f);
                if (d.res_add != nullptr) {
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(static_cast<const T*>(d.res_add) + ((long)img * HoWo + p) * d.ldc + nw0 + c * E), g);
#pragma unroll
                    for (int e = 0; e < E; ++e) f[e] += g[e];
                }
                const int h = p / d.Wo, w = p - h * d.Wo;
                const bool bd = d.border_add != nullptr;
                const bool rt_ = bd & (h == 1), rb_ = bd & (h == S - 2), cl_ = bd & (w == 1), cr_ = bd & (w == S - 2);
                auto acc_from = [&](int phase, int pos) {
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(bimg + ((long)phase * S + pos) * d.ldc + c * E), g);
#pragma unroll
                    for (int e = 0; e < E; ++e) f[e] += g[e];
                };
                if (rt_) acc_from(0, w);
                if (rb_) acc_from(1, w);
                if (cl_) acc_from(2, h);
                if (cr_) acc_from(3, h);
                if (rt_ & cl_) acc_from(4, 0);
                if (rt_ & cr_) acc_from(5, 0);
                if (rb_ & cl_) acc_from(6, 0);
                if (rb_ & cr_) acc_from(7, 0);
                return f32_to_chunk<T>(f);
            };
            store_tile_via_lds<T, MT, NT>(acc, scratch, lane, nullptr, d.act, d.slope, rowp, so, min(64, HoWo - pw), add);
            }
        }
    } else {
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int p = p0 + wm * WM + b * 16 + l16;
            if (p >= HoWo) continue;
            T* yp = y + ((long)img * HoWo + p) * d.ldc;
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                const int n = n_base + wn * WN + a * 16 + 4 * q;
                if (n >= d.Nstore) continue;
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = apply_act(acc[a][b][e], d.act, d.slope);
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
}
