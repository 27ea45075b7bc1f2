// wgrad_rows.hip — weight gradient of the stride-1 3x3 "same" convolutions (the ResBlock convs: 2/3 of the weight-gradient
// FLOPs of the step) for gfx950, bf16.
//
//   part[s][n][(kh*3+kw)*Cq + c] = sum_{image rows R in split s} sum_{j<64} dY[R][j][n] * X[R + kh-1][j + kw-1][c]
//
// The generic kernel (wgrad.hip) stages one [64 px][128 ch] tile of each operand per 128x128x64 MMA block: 64 FLOP per
// staged byte, which is exactly what the L2 -> LDS path of a CU sustains (64 B/clk against 4096 FLOP/clk) - it runs
// staging-bound at ~560 TFLOP/s.  Here one K-step is ONE IMAGE ROW (W = 64 pixels): a block stages the dY row slice
// [64 px][128 co] and the X row slice [64 px][128 ci] of input row R+kh-1 once and uses them for all THREE kw taps
// (fragment reads of the X tile shifted by one pixel row; reflection / zero padding is a per-lane row choice made once),
// i.e. 192 FLOP per staged byte.  Tiles: (Np/128) x (Cq/128) x 3 (kh); the image rows are split over the rest of the grid
// and the fp32 partial slabs go to the same deterministic wgrad_reduce kernels as before.
//
// Staging is LDS-DMA (buffer_load ... lds, 1 KiB per wave-instruction, four stages, one block barrier per K-step).  The
// DMA writes LDS linearly, so rows are 256 B with no padding; the 16-byte chunks of row r are XOR-swizzled by 2*(r&7) on
// the SOURCE address, which makes the ds_read_b64_tr_b16 fragment reads (8 pixel rows x 32 B per 32-lane group)
// conflict-free.  Both MFMA operands reduce over pixels, the strided index of NHWC: the transposing read delivers them.
//
// Round 3, HALO: image rows wider than 64 pixels (128-wide ResBlock maps of the 512x512 configuration, BASELINE configs[3]; any
// multiple of 64).  A K-step is then one 64-pixel SEGMENT of an image row; the kw = 0 / 2 taps of the segment's first / last pixel
// read the neighbouring segment's edge pixel (or, at the image border, the reflected pixel / zero).  Those two pixels are staged as
// two extra rows (64, 65) behind the X tile by ONE more DMA piece per K-step (wave 7; its counted vmcnt waits are one higher per
// stage in flight), with the reflection / zero decision made on the scalar side per step, so the per-lane fragment address table
// stays one table: pixel -1 -> row 64, pixel 64 -> row 65.  On 64-wide rows the original form (in-tile reflection) is kept.
//
// Round 3, S2: the STRIDE-2 3x3 pad-1 layers with 128-multiples of channels on both sides and 64-pixel output rows - down2
// (Conv2d 128 -> 256) and up1 (ConvTranspose2d 256 -> 128, whose weight gradient is the same contraction with the roles of the two
// maps swapped) at 256 x 256 - which ran on the generic split-K kernel at ~390 TFLOP/s.  P is the SMALL map (64-pixel rows), Q the
// large one (128-pixel rows): a K-step stages one P row [64 px][128 ch] and the Q row 2 i + kh - 1 [128 px][128 ch] (zeros for row -1)
// and the three kw taps read its pixels 2 j + kw - 1 (pixel -1 = the zero row): 131 FLOP per staged byte.  48 KB per stage, so three
// stages instead of four (the DMAs of step ks + 2 are issued at the START of step ks's second phase: one full step of flight); the Q
// tile's XOR swizzle is by ((row >> 1) & 7) so that the same-parity rows a tap reads stay conflict-free.
#include "uig_common.h"
#include <algorithm>
#include <type_traits>

struct WgRowsDesc {
    int B, H, Np, Cq, pad_mode;
    int W, S;                // image row width (a multiple of 64) and its 64-pixel segments per row; a K-step = one segment
    int Hq, Wq;              // S2: the large map's size (2 H x 128); otherwise H x W
    int ncols;               // 9 * Cq
    int rows_total;          // B * H * S K-steps ("rows" below = K-steps: image rows on 64-wide maps)
    int ntc, ntiles, splits; // ci tiles, tiles per network = (Np/128) * ntc * 3
    int group_rows;          // two networks in one launch: image rows [0, group_rows) are network 0's, the rest network 1's
                             // (0 = one network); partial slabs are laid out [network][split][Np][ncols]
    unsigned p_bytes, q_bytes;
    // The image rows of a network are up to two RUNS of whole images, each in one of two operand tensor pairs (round 2: the
    // weight gradient of BOTH generator passes of a step in one launch - pass 1's batch lives in (P, Q), pass 2's in (P2, Q2)).
    // Runs 0, 1 = network 0's, runs 2, 3 = network 1's, walked in this order; group_rows = run_rows[0] + run_rows[1].
    int run_rows[4];         // image rows (images * H) of each run; 0 = empty
    int run_img0[4];         // first image of the run inside its tensor
    int run_sel[4];          // 0: (P, Q), 1: (P2, Q2)
    unsigned p2_bytes, q2_bytes;
};

namespace {
constexpr int WR_W = 64;                       // pixels per image row = pixels per K-step
constexpr int WR_TILE = 64 * 256;              // one staged operand tile: 64 pixel rows x 128 channels bf16
constexpr int WR_NST = 4;
constexpr int wr_xtile(bool halo, bool s2 = false) { return s2 ? 128 * 256 : (halo ? 68 * 256 : WR_TILE); }   // HALO: + rows 64 (left neighbour), 65 (right), 66-67 unused; S2: a 128-pixel row
constexpr int wr_stage(bool halo, bool s2 = false) { return WR_TILE + wr_xtile(halo, s2) + 256; }        // dY tile | X tile | one zero row (zero padding of the X columns)
constexpr int wr_nst(bool s2) { return s2 ? 3 : WR_NST; }
}

template <bool HALO, bool S2 = false>
__global__ __launch_bounds__(512, 1) void wgrad_rows3_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q,
                                                               const bf16_t* __restrict__ P2, const bf16_t* __restrict__ Q2,
                                                               float* __restrict__ part, const WgRowsDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    static_assert(!(HALO && S2), "stride-2 rows are 64 pixels wide");
    constexpr int WR_STAGE = wr_stage(HALO, S2), XT0 = WR_TILE, ZROW = WR_TILE + wr_xtile(HALO, S2);      // stage-relative offsets: X tile, zero row
    constexpr int NST = wr_nst(S2);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware order: consecutive logical blocks (the tiles of one split: same image rows) share an XCD and its L2
    int bid;
    {
        const int nwg = gridDim.x, o = blockIdx.x, xcd = o & 7, qq = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (o >> 3);
    }
    const int nets = d.group_rows > 0 ? 2 : 1;
    const int tile2 = bid % (d.ntiles * nets), split = bid / (d.ntiles * nets);
    const int net = tile2 / d.ntiles, tile = tile2 % d.ntiles;
    const int kh = tile % 3, t2 = tile / 3;
    const int ci_base = (t2 % d.ntc) * 128, n_base = (t2 / d.ntc) * 128;
    const int net_row0 = net ? d.group_rows : 0;
    const int net_rows = nets == 1 ? d.rows_total : (net ? d.rows_total - d.group_rows : d.group_rows);
    const int row_begin = net_row0 + (int)((long)split * net_rows / d.splits);
    const int row_end = net_row0 + (int)((long)(split + 1) * net_rows / d.splits);
    const int nk = row_end - row_begin;

    // zero rows (one per stage)
    if (tid < NST * 16) *reinterpret_cast<u32x4_t*>(smem + (tid >> 4) * WR_STAGE + ZROW + (tid & 15) * 16) = u32x4_t{0u, 0u, 0u, 0u};

    // ---- DMA: wave w stages pieces w and w+8 (rows 4w..4w+3 and +32) of both tiles
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P), 0, d.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Q), 0, d.q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsP2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P2 ? P2 : P), 0, P2 ? d.p2_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Q2 ? Q2 : Q), 0, Q2 ? d.q2_bytes : 0u, 0x00020000);
    const int ra = 4 * wave + (lane >> 4);                            // pixel (LDS row) this lane fills in piece w
    const int chunk = (lane & 15) ^ ((ra & 7) << 1);                  // source chunk for LDS slot lane&15 (same for row ra+32)
    const int chunkq = S2 ? ((lane & 15) ^ (((ra >> 1) & 7) << 1)) : chunk;      // S2: the Q tile is swizzled by the row PAIR (same for rows ra + 32 k)
    const unsigned voffP = (unsigned)((ra * d.Np + n_base + chunk * 8) * 2);
    const unsigned voffQ = (unsigned)((ra * d.Cq + ci_base + chunkq * 8) * 2);
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    // (tensor, image, row, segment) of the next K-step to issue: the network's steps are run 2*net followed by run 2*net+1 (whole images)
    const int run_a = 2 * net, a_rows = d.run_rows[run_a];
    const int HS = d.H * d.S;                                         // K-steps per image
    const int l0 = row_begin - net_row0;                              // step inside the network
    const bool in_a = l0 < a_rows;
    const int loc0 = in_a ? l0 : l0 - a_rows;
    int sel = d.run_sel[in_a ? run_a : run_a + 1];
    int ib = d.run_img0[in_a ? run_a : run_a + 1] + loc0 / HS, ii = (loc0 % HS) / d.S, sg = (loc0 % HS) % d.S;
    int left = in_a ? a_rows - l0 : 0x7fffffff;                      // steps until the switch to the second run
    const int nx_sel = d.run_sel[run_a + 1], nx_img0 = d.run_img0[run_a + 1];
    // HALO: the extra piece (rows 64, 65 of the X tile): lanes 0-15 the left neighbour pixel, 16-31 the right one, the rest nothing
    const int hq = lane >> 4;
    const unsigned hchunk = (unsigned)(((lane & 15) ^ ((hq & 7) << 1)) * 16 + ci_base * 2);   // row 64 + hq: swizzle by 2 * ((64 + hq) & 7) = 2 * hq
    auto issue = [&](int stage) {
        const int hi = S2 ? 2 * ii + kh - 1 : ii + kh - 1;
        const bool valid = (!S2 && refl) | ((unsigned)hi < (unsigned)d.Hq);
        const int hr = (!S2 && refl) ? reflect_idx(hi, d.H) : (valid ? hi : 0);
        const int sP = __builtin_amdgcn_readfirstlane((int)(((unsigned)(ib * d.H + ii) * (unsigned)d.W + (unsigned)(sg * WR_W)) * 2u * (unsigned)d.Np));
        const int sQ = __builtin_amdgcn_readfirstlane((int)(((unsigned)(ib * d.Hq + hr) * (unsigned)d.Wq + (unsigned)(sg * WR_W)) * 2u * (unsigned)d.Cq));
        const unsigned vq = valid ? voffQ : 0xFFFFFFFFu;              // zero-padded row: out-of-range offset -> zeros
        lds_ptr_t dst = (lds_ptr_t)smem + stage * WR_STAGE + wave * 1024;
        if (__builtin_amdgcn_readfirstlane(sel) == 0) {               // block-uniform
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (__attribute__((address_space(3))) void*)dst, 16, (int)voffP, sP, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (__attribute__((address_space(3))) void*)(dst + 8192), 16, (int)voffP,
                                                     sP + 32 * d.Np * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + XT0), 16, (int)vq, sQ, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + XT0 + 8192), 16, (int)vq,
                                                     sQ + 32 * d.Cq * 2, 0, 0);
            if constexpr (S2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + XT0 + 2 * 8192), 16, (int)vq,
                                                         sQ + 64 * d.Cq * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + XT0 + 3 * 8192), 16, (int)vq,
                                                         sQ + 96 * d.Cq * 2, 0, 0);
            }
        } else {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP2, (__attribute__((address_space(3))) void*)dst, 16, (int)voffP, sP, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP2, (__attribute__((address_space(3))) void*)(dst + 8192), 16, (int)voffP,
                                                     sP + 32 * d.Np * 2, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ2, (__attribute__((address_space(3))) void*)(dst + XT0), 16, (int)vq, sQ, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ2, (__attribute__((address_space(3))) void*)(dst + XT0 + 8192), 16, (int)vq,
                                                     sQ + 32 * d.Cq * 2, 0, 0);
            if constexpr (S2) {
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ2, (__attribute__((address_space(3))) void*)(dst + XT0 + 2 * 8192), 16, (int)vq,
                                                         sQ + 64 * d.Cq * 2, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ2, (__attribute__((address_space(3))) void*)(dst + XT0 + 3 * 8192), 16, (int)vq,
                                                         sQ + 96 * d.Cq * 2, 0, 0);
            }
        }
        if constexpr (HALO) {
            if (wave == 7) {                                          // wave-uniform: one more piece per K-step from this wave (its vmcnt counts differ)
                // pixel left of / right of the segment inside input row hr; at the image border the reflected pixel, or nothing (zeros)
                int lp = sg * WR_W - 1, rp = sg * WR_W + WR_W;
                const bool lok = valid && (lp >= 0 || refl), rok = valid && (rp < d.W || refl);
                lp = lp < 0 ? 1 : lp; rp = rp >= d.W ? d.W - 2 : rp;
                const int sH = __builtin_amdgcn_readfirstlane((int)((unsigned)(ib * d.H + hr) * (unsigned)d.W * 2u * (unsigned)d.Cq));   // row base
                const bool ok = hq == 0 ? lok : (hq == 1 ? rok : false);
                const unsigned vh = ok ? hchunk + (unsigned)((hq == 0 ? lp : rp) * d.Cq * 2) : 0xFFFFFFFFu;
                lds_ptr_t hd = (lds_ptr_t)smem + stage * WR_STAGE + XT0 + 64 * 256;
                if (__builtin_amdgcn_readfirstlane(sel) == 0)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)hd, 16, (int)vh, sH, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ2, (__attribute__((address_space(3))) void*)hd, 16, (int)vh, sH, 0, 0);
            }
        }
        if (++sg == d.S) { sg = 0; if (++ii == d.H) { ii = 0; ++ib; } }
        if (--left == 0) { sel = nx_sel; ib = nx_img0; ii = 0; sg = 0; left = 0x7fffffff; }
    };

    // ---- fragment addressing (stage-relative byte offsets).  Wave (wn, wc): 64 co x 32 ci x 3 kw.
    // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses pixel row 4g+q (+16, +32), channels 4p..4p+3 of a 16-channel
    // tile, and receives channel l16 of those 4 pixel rows.
    const int wn = wave & 1, wc = wave >> 1;
    const int l16 = lane & 15, g = lane >> 4, qq = l16 >> 2, pp = l16 & 3;
    const int k0 = 4 * g + qq;                                        // pixel of this lane within a 16-pixel group
    unsigned poff[4];                                                 // dY tile: row k0, co tile b (rows +16 / +32 / +48 are immediates)
#pragma unroll
    for (int b = 0; b < 4; ++b)
        poff[b] = (unsigned)(k0 * 256 + (((wn * 8 + 2 * b + (pp >> 1)) ^ ((k0 & 7) << 1)) << 4) + (pp & 1) * 8);
    unsigned qoff[3][4][2];                                           // X tile: [kw][16-pixel group][ci tile a]
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
            const int pix = S2 ? 2 * (16 * gq + k0) + kw - 1 : 16 * gq + k0 + kw - 1;      // S2: pixel of the 128-pixel Q row (-1 only for kw = 0, j = 0)
            const bool inb = S2 ? pix >= 0 : (unsigned)pix < (unsigned)WR_W;
            // HALO: the segment's neighbours sit in rows 64 / 65 (already reflected / zeroed by the DMA that staged them)
            const int row = S2 ? (inb ? pix : 0) : (HALO ? (pix < 0 ? 64 : (pix >= WR_W ? 65 : pix)) : (refl ? reflect_idx(pix, WR_W) : (inb ? pix : 0)));
            const bool zero = S2 ? !inb : (!HALO && !refl && !inb);
            const int swz = S2 ? (((row >> 1) & 7) << 1) : ((row & 7) << 1);
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int cidx = wc * 4 + 2 * a + (pp >> 1);
                qoff[kw][gq][a] = zero ? (unsigned)(ZROW + ((cidx & 15) << 4) + (pp & 1) * 8)
                                       : (unsigned)(XT0 + row * 256 + ((cidx ^ swz) << 4) + (pp & 1) * 8);
            }
        }

    f32x4_t acc[3][2][4];
#pragma unroll
    for (int kw = 0; kw < 3; ++kw)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 4; ++b) acc[kw][a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    // The transposing reads are issued as inline asm: for the builtin form the compiler cannot tell an LDS-DMA write from a
    // hazard on these reads and drains every in-flight DMA (s_waitcnt vmcnt(0)) before each group of them, which serialises
    // the whole pipeline (measured: 3100 cycles per K-step instead of ~1700).  The price: their lgkmcnt is ours to wait for
    // (frags_ready ties the wait to the fragment registers so no MFMA can be scheduled above it).
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    auto tr_read = [&](unsigned addr, auto off) -> bf16x4_t {
        u32x2_t r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value));
        return __builtin_bit_cast(bf16x4_t, r);
    };
    // fragment reads in five chunks of four (c = 0,1: dY tiles 2c, 2c+1; c = 2..4: X tap kw = c-2, both ci tiles)
    auto read_chunk = [&](bf16x8_t (&bf)[4], bf16x8_t (&af)[3][2], unsigned st, auto kgc, int c) {
        constexpr int kg = decltype(kgc)::value;
        if (c < 2) {
#pragma unroll
            for (int b = 2 * c; b < 2 * c + 2; ++b) {
                const unsigned pb = st + poff[b];
                const bf16x4_t lo = tr_read(pb, std::integral_constant<int, kg * 32 * 256>{});
                const bf16x4_t hi = tr_read(pb, std::integral_constant<int, kg * 32 * 256 + 16 * 256>{});
                bf[b] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        } else {
            const int kw = c - 2;
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const bf16x4_t lo = tr_read(st + qoff[kw][2 * kg][a], std::integral_constant<int, 0>{});
                const bf16x4_t hi = tr_read(st + qoff[kw][2 * kg + 1][a], std::integral_constant<int, 0>{});
                af[kw][a] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
            }
        }
    };
    auto frags_ready = [&](bf16x8_t (&bf)[4], bf16x8_t (&af)[3][2]) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]), "+v"(bf[3]), "+v"(af[0][0]), "+v"(af[0][1]), "+v"(af[1][0]),
                       "+v"(af[1][1]), "+v"(af[2][0]), "+v"(af[2][1]));
    };
    auto mma_group = [&](const bf16x8_t (&bf)[4], const bf16x8_t (&af)[3][2], int gi) {      // gi = kw * 2 + a: four MFMAs
        const int kw = gi >> 1, a = gi & 1;
#pragma unroll
        for (int b = 0; b < 4; ++b)
            acc[kw][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[kw][a], bf[b], acc[kw][a][b], 0, 0, 0);
    };

    // Software pipeline, skewed by half a K-step.  A phase = the 24 MFMAs of one 32-pixel half (six groups of four); the 20
    // transposing reads of the NEXT half (into the other fragment set) and, in the second phase, the four DMAs of step ks+3
    // are issued BETWEEN the MFMA groups, so the matrix pipe is fed from the first cycle after the barrier - the two waves
    // of a SIMD leave the barrier together, and with the reads / DMA issue up front both stalled the pipe for ~700 cycles
    // per step.  Four stages: the block barrier in the middle of step ks publishes stage ks+1 (read right after it) and
    // retires stage ks-1, which the DMAs of step ks+3 overwrite - two full steps of flight time for every DMA.
    bf16x8_t bf0[4], af0[3][2], bf1[4], af1[3][2];
#ifdef UIG_X_STAMP
    const unsigned long long xt0 = __builtin_amdgcn_s_memtime(), xr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long xw_dma = 0, xw_bar = 0;
#endif
    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    if (!S2 && nk > 2) issue(2);
    if (nk > 0) {
        // stages allowed to stay in flight: 2 / 1 / 0 (4 pieces per stage from every wave; 5 from wave 7 with HALO); S2: 1 / 0 (6 pieces)
        if constexpr (S2) {
            if (nk > 1) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (HALO && wave == 7) {
            if (nk > 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            else if (nk > 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else {
            if (nk > 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else if (nk > 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int c = 0; c < 5; ++c) read_chunk(bf0, af0, lds0, std::integral_constant<int, 0>{}, c);
    }
    for (int ks = 0; ks < nk; ++ks) {
        const unsigned st = lds0 + (unsigned)((ks % NST) * WR_STAGE);
        frags_ready(bf0, af0);                             // issued half a step ago: no stall
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) {
            __builtin_amdgcn_sched_barrier(0);
            mma_group(bf0, af0, gi);
            __builtin_amdgcn_sched_barrier(0);
            if (gi < 5) read_chunk(bf1, af1, st, std::integral_constant<int, 1>{}, gi);
        }
        __builtin_amdgcn_sched_barrier(0);
        frags_ready(bf1, af1);
        const bool more = ks + 1 < nk;                     // block-uniform
        if (more) {
#ifdef UIG_X_STAMP
            const unsigned long long w0 = __builtin_amdgcn_s_memtime();
#endif
            if (!S2 && ks + 2 < nk) {                                               // stage ks+1 landed; stage ks+2 may still fly
                if (HALO && wave == 7) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // S2: stage ks+2 is only issued behind this barrier
#ifdef UIG_X_STAMP
            const unsigned long long w1 = __builtin_amdgcn_s_memtime();
#endif
            __builtin_amdgcn_s_barrier();
#ifdef UIG_X_STAMP
            const unsigned long long w2 = __builtin_amdgcn_s_memtime();
            xw_dma += w1 - w0; xw_bar += w2 - w1;
#endif
        }
        const unsigned stn = lds0 + (unsigned)(((ks + 1) % NST) * WR_STAGE);
        if constexpr (S2) {                                    // three stages: stage (ks + 2) % 3 = the one step ks - 1 read, retired by the barrier above
            if (ks + 2 < nk) issue((ks + 2) % NST);
        }
#pragma unroll
        for (int gi = 0; gi < 6; ++gi) {
            __builtin_amdgcn_sched_barrier(0);
            mma_group(bf1, af1, gi);
            __builtin_amdgcn_sched_barrier(0);
            if (more && gi < 5) read_chunk(bf0, af0, stn, std::integral_constant<int, 0>{}, gi);
#ifndef UIG_X_NODMA
            if (!S2 && gi == 4 && ks + 3 < nk) issue((ks + 3) % WR_NST);
#endif
        }
        __builtin_amdgcn_sched_barrier(0);
    }

#ifdef UIG_X_STAMP
    const unsigned long long xt1 = __builtin_amdgcn_s_memtime(), xr1 = __builtin_amdgcn_s_memrealtime();
#endif
    // D[col][n]: lane holds n = l16 (B-operand column), cols 4g..4g+3 -> one float4 per tile into part[split][n][col]
    float* out = part + ((long)net * d.splits + split) * d.Np * d.ncols;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        const int n = n_base + wn * 64 + b * 16 + l16;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw)
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                const int col = (kh * 3 + kw) * d.Cq + ci_base + wc * 32 + a * 16 + 4 * g;
                *reinterpret_cast<f32x4_t*>(out + (long)n * d.ncols + col) = acc[kw][a][b];
            }
    }
#ifdef UIG_X_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long xt2 = __builtin_amdgcn_s_memtime(), xr2 = __builtin_amdgcn_s_memrealtime();
    if (tile == 0 && tid == 0) {
        float* o = part + ((long)net * d.splits + split) * d.Np * d.ncols;
        o[0] = (float)(xt1 - xt0); o[1] = (float)(xr1 - xr0); o[2] = (float)(xt2 - xt1); o[3] = (float)(xr2 - xr1); o[4] = (float)xw_dma; o[5] = (float)xw_bar;
    }
#endif
}

static int g_wgrad_rows = 1;    // 1 = use this kernel where it applies, 0 = never (A/B and parity hook)
extern "C" void uig_debug_set_wgrad_rows(int on) { g_wgrad_rows = on; }

// 1 if uig_wgrad_partial runs this launch on the row kernel
static int g_wgrad_rows_s2 = 1; // 1 = the stride-2 form too (round 3), 0 = stride-2 layers stay on the generic kernel (A/B and parity hook)
extern "C" void uig_debug_set_wgrad_rows_s2(int on) { g_wgrad_rows_s2 = on; }
bool uig_wgrad_rows_applicable(int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int dtype, int pad_mode) {
    if (!(g_wgrad_rows && dtype == UIG_BF16 && kH == 3 && kW == 3 && pad == 1 && Np % 128 == 0 && Cq % 128 == 0 && Hq >= 2)) return false;
    if (stride == 2)            // S2: 64-pixel rows of the small map against 128-pixel rows of the large one, zero padding
        return g_wgrad_rows_s2 && pad_mode == UIG_PAD_ZERO && Mw == WR_W && Hq == 2 * Mh && Wq == 2 * Mw;
    return stride == 1 && Mh == Hq && Mw == Wq && Mw % WR_W == 0 && Mw <= 1024 && (g_wgrad_rows != 2 || Mw == WR_W);      // hook value 2: 64-wide rows only (A/B)
}

int uig_wgrad_rows_tiles(int Np, int Cq) { return (Np / 128) * (Cq / 128) * 3; }

// General form: up to two runs of whole images per network, each in (P, Q) [sel 0] or (P2, Q2) [sel 1]; imgs[r] images starting
// at image img0[r] of its tensor; runs 0, 1 are network 0's, runs 2, 3 network 1's (one network: runs 2, 3 empty).
// B1 / B2 = images in (P, Q) / (P2, Q2) (for the range checks of the buffer descriptors).
int uig_launch_wgrad_rows_runs(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, int B1, int B2, int H, int W, int Np, int Cq,
                               int pad_mode, int splits, const int* imgs, const int* img0, const int* sel, hipStream_t s, int stride) {
    WgRowsDesc d{};
    int total = 0;
    const int S = W / WR_W;
    const bool s2 = stride == 2;                      // H x W = the SMALL map (P); Q is 2 H x 2 W
    if (s2 && (W != WR_W || pad_mode != UIG_PAD_ZERO)) return uig_set_error(-1, "wgrad(rows, stride 2): 64-pixel rows and zero padding only");
    d.W = W; d.S = S; d.Hq = s2 ? 2 * H : H; d.Wq = s2 ? 2 * W : W;
    for (int r = 0; r < 4; ++r) { d.run_rows[r] = imgs[r] * H * S; d.run_img0[r] = img0[r]; d.run_sel[r] = sel[r]; total += imgs[r]; }
    const bool two = imgs[2] + imgs[3] > 0;
    d.group_rows = two ? d.run_rows[0] + d.run_rows[1] : 0;
    d.B = total; d.H = H; d.Np = Np; d.Cq = Cq; d.pad_mode = pad_mode; d.ncols = 9 * Cq; d.rows_total = total * H * S;
    d.ntc = Cq / 128; d.ntiles = uig_wgrad_rows_tiles(Np, Cq); d.splits = splits;
    const long qpix = (long)d.Hq * d.Wq;
    d.p_bytes = (unsigned)((long)B1 * H * W * Np * 2); d.q_bytes = (unsigned)((long)B1 * qpix * Cq * 2);
    d.p2_bytes = (unsigned)((long)B2 * H * W * Np * 2); d.q2_bytes = (unsigned)((long)B2 * qpix * Cq * 2);
    if ((long)std::max(B1, B2) * std::max((long)H * W * Np, qpix * Cq) * 2 >= (1L << 32) - 64) return uig_set_error(-1, "wgrad(rows): operand larger than 4 GiB");
    const bool halo = S > 1;
    const size_t smem = (size_t)wr_nst(s2) * wr_stage(halo, s2);
    static SmemAttrOnce attr0, attr1, attr2;
    {
        hipError_t e = s2 ? attr2.ensure(reinterpret_cast<const void*>(wgrad_rows3_kernel<false, true>), smem)
                     : halo ? attr1.ensure(reinterpret_cast<const void*>(wgrad_rows3_kernel<true>), smem)
                            : attr0.ensure(reinterpret_cast<const void*>(wgrad_rows3_kernel<false>), smem);
        if (e != hipSuccess) return uig_set_error((int)e, "wgrad(rows): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    const dim3 grid(d.ntiles * splits * (two ? 2 : 1));
    if (s2) hipLaunchKernelGGL((wgrad_rows3_kernel<false, true>), grid, dim3(512), smem, s, (const bf16_t*)P, (const bf16_t*)Q, (const bf16_t*)P2, (const bf16_t*)Q2, ws, d);
    else if (halo) hipLaunchKernelGGL(wgrad_rows3_kernel<true>, grid, dim3(512), smem, s, (const bf16_t*)P, (const bf16_t*)Q, (const bf16_t*)P2, (const bf16_t*)Q2, ws, d);
    else hipLaunchKernelGGL(wgrad_rows3_kernel<false>, grid, dim3(512), smem, s, (const bf16_t*)P, (const bf16_t*)Q, (const bf16_t*)P2, (const bf16_t*)Q2, ws, d);
    UIG_LAUNCH_CHECK("uig_wgrad_partial(rows)");
    return 0;
}

int uig_launch_wgrad_rows(const void* P, const void* Q, float* ws, int B, int H, int W, int Np, int Cq, int pad_mode, int splits,
                          int group_images, hipStream_t s, int stride) {
    const int imgs[4] = {group_images > 0 ? group_images : B, 0, group_images > 0 ? B - group_images : 0, 0};
    const int img0[4] = {0, 0, group_images, 0}, sel[4] = {0, 0, 0, 0};
    return uig_launch_wgrad_rows_runs(P, Q, nullptr, nullptr, ws, B, 0, H, W, Np, Cq, pad_mode, splits, imgs, img0, sel, s, stride);
}
