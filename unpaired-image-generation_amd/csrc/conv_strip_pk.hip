// conv_strip_pk.hip — persistent form of the LDS-resident-strip convolution (conv_strip.hip) for its dominant shape: 256 output
// pixels x 128 channels per tile, 8 waves, stride-1 3x3, forward and input gradient of the ResBlock convs.
//
// What the one-tile-per-block kernel leaves on the table (in-kernel s_memtime stamps): the paired ResBlock launch is
// 512 tiles = exactly two rounds of one 149.5-KB block per CU, and in each round the tile's prologue (row table, first strip
// chunk, first weight tile: ~6 us), its epilogue (~5.6 us) and the block hand-over are fully exposed: 34 % of the launch.
// Here the grid is one block per CU and a block WALKS its tiles (tile = round * gridDim + XCD-remapped block id):
//   * the next tile's first strip chunk and first weight tile are streamed in (LDS-DMA) behind the current tile's LAST chunk
//     of MFMAs, exactly like a chunk switch inside a tile, so a tile's K loop starts the moment the previous epilogue ends;
//   * the epilogue's LDS scratch lives in the region the last K-step has just finished with ([strip s][weights s], contiguous),
//     the prefetch targets the other region: no extra LDS;
//   * the per-lane row table depends only on the tile's position inside its image, and the tiles a block walks are a whole
//     number of images apart in the common shapes: it is computed once per block, not once per tile;
//   * LDS swizzle: slot = (chunk + (row & 6)) & 7 — a rotation instead of the XOR of conv_igemm.hip.  The XOR form is
//     conflict-free for the 16-row-aligned fragment reads of a GEMM tile but 2-way conflicted for the +-1-pixel SHIFTED reads
//     that generate the dw = +-1 taps (6 of 9 taps: SQ_LDS_BANK_CONFLICT was 26 % of the LDS cycles of the round-1 kernel);
//     the rotation is conflict-free at every shift (bank analysis in DESIGN.md §3.2).
//   * DMA issue is spread between the MFMA groups of a K-step (DM = 1) instead of all 8 waves issuing their 3 pieces at the
//     top of the step while the matrix pipe idles.
// Round 4: the K loop of the bf16 kernel runs a PHASED schedule by default where the shape allows (DM = 9: two wave groups one barrier
// apart, three weight stages, counted LDS-DMA waits - the comment at the head of that loop); round 3's loop (DM = 5, four
// issuing waves, next step's strip fragments read behind this step's MFMAs) serves the NORM / BST / 512-row / odd-chunk-count launches
// and stays selectable for A/B (uig_debug_set_strip_pk(5, 0)).
#include "conv_strip_desc.h"
#include <algorithm>
#include <type_traits>

// SWZ: strip swizzle (1 = rotation, 0 = the XOR of conv_igemm.hip); LGK: wait for this wave's own fragment reads before the
// K-step barrier (strictly orders them before the DMA that overwrites the weight stage; 0 = rely on the DMA's latency as the
// one-tile-per-block kernel does); STAMP: diagnostic build writing s_memtime stamps per wave to d.dbg
//
// MIRROR (bf16, 64-pixel-wide maps): input gradient of a REFLECTION-padded 3x3 convolution without a border GEMM.  The adjoint of
// pad-1 reflection folds padded line -1 onto line 1 and line H onto line H-2 (columns alike), i.e.
//     dx[i][j] = sum_{dh,dw} S_{dh,dw}[i+dh][j+dw] * W'[dh][dw],     S_{dh,dw} = dy except:
//     dh = +1: line 2 := line 2 + line 0      dh = -1: line H-3 := line H-3 + line H-1      (and the same in columns with dw)
// - a ZERO-padded input gradient whose taps read a few "mirror pixels" (sums of two, at the four corners of four, pixels of dy)
// in place of the pixel itself: output line 1 with dh = +1, line H-2 with dh = -1, column 1 with dw = +1, column W-2 with dw = -1.
// The row table addresses pixels individually, so the mirror pixels are just extra strip slots behind the tile's NS pixels:
//     NS + 8r + 2, NS + 8r + 5     column mirrors of strip line r: dy[r][2] + dy[r][0], dy[r][W-3] + dy[r][W-1]       (r < 6)
//     NS + 48 + c  (c < 64)        line mirror (tile 0: dy[2][c] + dy[0][c]; last tile: dy[H-3][c] + dy[H-1][c])
//     NS + 48 + 66, NS + 48 + 69   the line mirror's own column mirrors (four-pixel sums: the corner terms)
// (a mirror pixel's slot has the low 3 bits of the pixel it stands in for - W is 64 - which is what the LDS bank group of a slot
// depends on: the fragment reads keep the conflict-free pattern of 16 consecutive pixels)
// summed in fp32 from the landed strip and rounded once to bf16, by all threads during the LAST tap-step of the previous chunk
// (every strip piece of the chunk has landed and been published by then: pieces are issued in steps 0-6), visible to all after
// the next step's barrier.  12 (+66) pixels x 8 chunks of work per K chunk against 36 x 512 MFMAs: nothing, and the 22-us
// eight-phase border GEMM in front of every input-gradient launch (uig_reflect3x3_dgrad_border) and the epilogue's border loads go.
//
// NORM (bf16; round 3): the input x is the raw output of the convolution in front of an InstanceNorm(+ReLU) and THIS launch applies
// the norm - conv2 of a ResBlock consumes conv1's output directly, the apply pass between them (one read + one write of the whole
// tensor per ResBlock and pass: 18 launches of a train step) is gone.  The strip arrives in LDS by DMA as before and is rewritten IN
// PLACE piece by piece: in tap-step t a wave normalises the 1-KiB piece it DMA'd itself in step t - 1 (landed by its own vmcnt wait at
// the top of the step; nobody reads the chunk before the next chunk's first barrier), one 16-byte chunk per lane: ds_read_b128, 8 x
// (sub, mul, max), pack, ds_write_b128 behind the first half of the step's 32 MFMAs.  (The first form rewrote the whole chunk at the
// top of the chunk's last step - a burst of 6 such items per thread with all 8 waves on the VALU and the LDS write path at once:
// +27 us per launch, slower than the apply pass it replaced.)  (mean, rstd) of the (at most NTB)
// images whose tiles this block walks are copied from the norm's statistics tensor into an LDS table behind the two regions at the
// head of the launch.  Optionally the normalised activations of the tile's OWN pixels are also stored to d.nrm_h by the blocks of
// the first channel tile (the backward pass needs them as the weight-gradient operand): a write without the apply pass's read,
// hidden behind the MFMAs.  Same arithmetic as in_apply_fwd_kernel ((x - mean) * rstd, act, one rounding): bitwise the two-launch
// result.
//
// NOZ (round 3): no zero rows behind the strip - for launches that never read one (reflection padding, whole tiles): the strip may then
// be CAP = 512 rows (2 x 80 KB of LDS exactly, 16-bit row table up to 65,520), i.e. a 256-pixel tile of a 128-pixel-wide map (two
// image rows + two halo rows): the ResBlock forward convolutions of the 512x512 configuration on this kernel instead of the generic one.
template <typename T, int CAP, int DM, int SWZ = 1, bool LGK = true, bool STAMP = false, bool MIRROR = false, bool NORM = false, bool NOZ = false, int NISS = 8, bool PKRT = false, bool BST = false>
__global__ __launch_bounds__(512, 2)
void conv_strip_pk_kernel(const T* __restrict__ x, const T* __restrict__ wp1, const float* __restrict__ bias1, T* __restrict__ y,
                          const StripDesc d) {
    constexpr int E = ElemTraits<T>::E;
    constexpr int BK = 8 * E;
    constexpr int BM = 256, BN = 128, NW = 8, NTAPS = 9, WM = 64, WN = 64, MT = 4, NT = 4;
    constexpr int PIECES = CAP / 8;
    // PH (DM == 9, round 4): the phased schedule - three weight stages, LDS [strip 0][W0][W1][strip 1][W2] (see the K loop)
    constexpr bool PH = DM == 9;
    constexpr int SBUF = (CAP + (NOZ ? 0 : 8)) * 128, WSTG = BN * 128, REG = PH ? SBUF + 2 * WSTG : SBUF + WSTG;     // LDS: [strip 0][weights 0][strip 1][weights 1]
    constexpr int SCRW = 64 * 64 * (int)sizeof(T);                                  // one wave's epilogue scratch
    constexpr bool XPREF = NW * SCRW <= SBUF + WSTG;      // the scratch fits the region the last K-step used: prefetch the next tile behind the last chunk
    static_assert(!PH || (sizeof(T) == 2 && SWZ == 1 && XPREF && !NORM && !NOZ && NISS == 8 && 2 * SBUF + 3 * WSTG <= 163840), "phased schedule: bf16, eight issuing waves, 160 KB");
    constexpr int ZW = CAP * 128 / SCRW;          // the wave whose scratch covers the region's zero row
    static_assert(CAP % 8 == 0 && PIECES <= NTAPS * NW, "one strip piece per wave per K-step");
    static_assert(NOZ || !XPREF || (CAP * 128) % SCRW + 1024 <= SCRW, "zero row must lie inside one wave's scratch");
    static_assert(!NOZ || (XPREF && !MIRROR && !NORM && (CAP - 1) * 128 + 127 < 65536), "no-zero-row variant: plain forward launches, 16-bit row table");
    static_assert(!MIRROR || (sizeof(T) == 2 && SWZ == 1 && XPREF && 6 * 64 + 48 <= CAP && 5 * 64 + 48 + 72 <= CAP), "mirror pixels: bf16, 64-wide maps");
    static_assert(!NORM || (sizeof(T) == 2 && SWZ == 1 && XPREF && !MIRROR), "norm strip: bf16 forward launches");
    constexpr int NTB = 4;                        // NORM: images (= tiles) per block whose (mean, rstd) fit the LDS table (host-checked)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = tid >> 3, ch = tid & 7;
    const int wm = wave & 3, wn = wave >> 2;
    const int l16 = lane & 15, q = lane >> 4, l8 = lane >> 3, ls = lane & 7;
    const int Cin = d.Cin, HoWo = d.Ho * d.Wo;
    const int ncc = Cin / BK;
    const int ntn = d.Nrows / BN, tpi = (HoWo + BM - 1) / BM;
    const int ntiles = d.B * tpi * ntn, G = gridDim.x;

    struct Tile { int img, p0, lo, NS, n_base, ti; bool g2, valid; };
    auto get_tile = [&](int r) -> Tile {
        Tile t{};
        const int base = r * G, nwg = min(G, ntiles - base), o = blockIdx.x;
        t.valid = nwg > 0 && o < nwg;
        if (!t.valid) return t;
        const int xcd = o & 7, qq = nwg >> 3, rr = nwg & 7;      // bijective XCD remap inside the round (T1)
        const int bid = base + (xcd < rr ? xcd * (qq + 1) : rr * (qq + 1) + (xcd - rr) * qq) + (o >> 3);
        auto mdiv = [](int n, unsigned long long mg) -> int { return (int)(((unsigned long long)(unsigned)n * mg) >> 32); };      // n / d by the host's magic (conv_strip_desc.h)
        const int mtile = mdiv(bid, d.mg_ntn);
        t.n_base = (bid - mtile * ntn) * BN;
        t.img = mdiv(mtile, d.mg_tpi); t.ti = mtile - t.img * tpi; t.p0 = t.ti * BM;
        const int p_last = min(t.p0 + BM, HoWo) - 1;
        t.lo = max(0, mdiv(t.p0, d.mg_wo) + d.dh_min);
        const int hi = min(d.H - 1, mdiv(p_last, d.mg_wo) + d.dh_max);
        t.NS = (hi - t.lo + 1) * d.W;
        t.g2 = d.wp2 != nullptr && t.img >= d.group_images;
        return t;
    };

    const __amdgpu_buffer_rsrc_t rsx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x), 0, d.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(wp1), 0, d.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsw2 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(d.wp2 != nullptr ? d.wp2 : static_cast<const void*>(wp1)), 0, d.w_bytes, 0x00020000);

    auto zero_rows = [&]() {                                  // rows CAP .. CAP+7 of both strip buffers
        if constexpr (!NOZ)
        for (int i = tid; i < 2 * 64; i += 64 * NW)
            *reinterpret_cast<u32x4_t*>(smem + (i >> 6) * REG + CAP * 128 + (i & 63) * 16) = u32x4_t{0u, 0u, 0u, 0u};
    };
    zero_rows();
    [[maybe_unused]] float* ntab = reinterpret_cast<float*>(smem + 2 * REG);      // NORM: [NTB][Cin][2] (mean, rstd)

    // ---- strip DMA: piece j = strip rows 8j..8j+7; lane L -> row 8j + L/8, physical slot L%8 holding chunk (L%8 - (row & 6)) & 7
    const unsigned svl = (unsigned)((l8 * Cin + (SWZ ? ((ls - (l8 & 6)) & 7) : (ls ^ (l8 >> 1))) * E) * (int)sizeof(T));
    auto strip_base = [&](const Tile& t, int cc) -> unsigned {                 // scalar byte offset of (image, first strip row, chunk)
        return (unsigned)__builtin_amdgcn_readfirstlane((((t.img * d.H + t.lo) * d.W) * Cin + cc * BK) * (int)sizeof(T));
    };
    const unsigned svl_odd = SWZ ? svl : (unsigned)((l8 * Cin + (ls ^ ((l8 >> 1) | 4)) * E) * (int)sizeof(T));   // XOR form: odd pieces flip bit 2
    auto issue_strip_piece = [&](int j, unsigned sbase, int NS, int region) {  // all arguments wave-uniform
        // the lane offset is made opaque here: left visible, the optimiser hoists the 9 (18 with four issuing waves) per-tap offsets
        // svl + 8 j Cin sizeof(T) out of the chunk loop - they do not depend on the chunk - and carries them through the K loop in as
        // many registers (the four-issuing-wave variant then spilled; round 3)
        unsigned sv = (j & 1) ? svl_odd : svl;
        asm volatile("" : "+v"(sv));
        const unsigned off = (8 * j + l8 < NS) ? sv + (unsigned)(8 * j * Cin * (int)sizeof(T)) : 0xFFFFFFFFu;
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + j * 1024;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)dst, 16, (int)off, (int)sbase, 0, 0);
    };
    // ---- weight tile DMA: rows n_base + lr + 64 i, 16-byte chunk ch of physical slot ch ^ ((row >> 1) & 7) (rows are 16-aligned
    //      per MFMA tile, the XOR form is conflict-free there)
    // NISS (round 3): only waves 0 .. NISS-1 issue the K loop's DMAs (16 / NISS weight pieces + NW / NISS strip pieces per step each).  A
    // 1-KiB LDS-DMA instruction occupies the CU's one address path for ~16-20 cycles and a wave sits in its issue until the path takes it:
    // with all eight waves issuing their three pieces behind the step's barrier, every wave stood there for the whole burst with the matrix
    // pipes empty (in-kernel stamps of the fp8 kernel: ~450 cycles per step).  With NISS = 4 the other wave of every SIMD goes straight to
    // its fragment reads and MFMAs.  NORM keeps 8: its waves normalise the strip pieces they issued themselves.  Measured on this kernel
    // (scripts/bench_strip_pk.py, A/B in one process, bitwise equal): with the row table packed to make room (variants 20 / 21) the
    // packing costs what the split gains (forward 69.6 -> 67.9 us, mirror-pixel input gradient 75.3 -> 77.4 us); with the PLAIN row
    // table the variant first spilled 2-13 registers around the tile loop (the few bytes of scratch made the STEP 0.14 ms slower although
    // every launch was faster) until the strip pieces' per-tap lane offsets were kept from being hoisted out of the chunk loop
    // (issue_strip_piece): 249 / 254 registers, no scratch.  Forward 71.7 -> 68.6 us (16 images) / 39.0 -> 37.7 (8), mirror-pixel input
    // gradient 78.1 -> 75.3 / 42.8 -> 42.2; same box, bench.py: dominant launch 70.8 -> 68.4 us (0.437 -> 0.452 of peak), strip family in
    // the step 0.406 -> 0.417, generator forward 2.380 -> 2.349 ms; the step itself 13.48 -> 13.47 ms (a tie).  Default since.
    static_assert((NISS == 4 || NISS == 8) && (!NORM || NISS == 8), "issuing waves");
    constexpr int WPI = 16 / NISS;                             // weight pieces per issuing wave and step: piece wave + NISS i = rows + 8 NISS i (same swizzle term)
    const unsigned wvl0 = (unsigned)((lr * d.ldw + (ch ^ ((lr >> 1) & 7)) * E) * (int)sizeof(T));
    auto w_base = [&](const Tile& t, int tp, int cc) -> unsigned {
        const int te = __builtin_amdgcn_readfirstlane(d.tap[tp]);
        return (unsigned)__builtin_amdgcn_readfirstlane((t.n_base * d.ldw + (te >> 16) * Cin + cc * BK) * (int)sizeof(T));
    };
    auto issue_w1 = [&](int i, bool g2, unsigned so, int region) {
        lds_ptr_t dst = (lds_ptr_t)smem + region * REG + SBUF + (wave + NISS * i) * 1024;
        const int soi = (int)so + i * NISS * 8 * d.ldw * (int)sizeof(T);
        if (g2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (__attribute__((address_space(3))) void*)dst, 16, (int)wvl0, soi, 0, 0);
        else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (__attribute__((address_space(3))) void*)dst, 16, (int)wvl0, soi, 0, 0);
    };
    // PH: weight stage k of three (k = tap % 3: a tile has 36 = 0 mod 3 steps) and its two DMA pieces per wave; `on` false = a zero fill
    // (out-of-range offset: no memory access) so that every step issues the same NUMBER of DMAs - the K loop's waits are counted
    [[maybe_unused]] auto wst_off = [&](int k) -> int { return k == 2 ? REG + SBUF : SBUF + k * WSTG; };
    [[maybe_unused]] auto issue_wph = [&](int i, bool g2, unsigned so, int stage, bool on) {
        lds_ptr_t dst = (lds_ptr_t)smem + wst_off(stage) + (wave + NW * i) * 1024;
        const int soi = (int)so + i * NW * 8 * d.ldw * (int)sizeof(T);
        const unsigned vo = on ? wvl0 : 0xFFFFFFFFu;
        if (g2) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw2, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, soi, 0, 0);
        else    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw1, (__attribute__((address_space(3))) void*)dst, 16, (int)vo, soi, 0, 0);
    };
    auto issue_first = [&](const Tile& t, int region) {       // a tile's chunk-0 strip and first weight tile, all at once
        const unsigned sb = strip_base(t, 0);
        for (int j = wave; 8 * j < t.NS; j += NW) issue_strip_piece(j, sb, t.NS, region);
        if constexpr (PH) {                                    // taps 0 and 1 -> stages 0 and 1 
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const unsigned so = w_base(t, k, 0);
#pragma unroll
                for (int i = 0; i < 2; ++i) issue_wph(i, t.g2, so, k, true);
            }
        } else {
        const unsigned so = w_base(t, 0, 0);
        if (wave < NISS) {
#pragma unroll
            for (int i = 0; i < WPI; ++i) issue_w1(i, t.g2, so, region);
        }
        }
    };

    // ---- mirror pixels of one strip chunk (MIRROR): see the kernel comment.  Block-uniform control flow, no barrier inside.
    auto mirror_fix = [&](int region, const Tile& t) {
        if constexpr (MIRROR) {
            unsigned char* sb = smem + region * REG;
            const int nrows = t.NS >> 6;
            const bool top = t.ti == 0, edge = top || t.ti == tpi - 1;
            const int rA = (top ? 2 : d.H - 3) - t.lo, rB = (top ? 0 : d.H - 1) - t.lo;      // strip lines summed into the line mirror
            const int nitem = (edge ? 12 + 66 : 12) * 8;
            auto ld = [&](int slot, int k, float (&f)[8]) {
                chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + slot * 128 + (((k + (slot & 6)) & 7) << 4)), f);
            };
            for (int i = tid; i < nitem; i += 64 * NW) {
                const int px = i >> 3, k = i & 7;
                int s0, s1, s2 = -1, s3 = -1;
                if (px < 12) {
                    const int side = px >= 6 ? 1 : 0, r = px - 6 * side;
                    if (r >= nrows) continue;
                    s0 = r * 64 + (side ? d.W - 3 : 2); s1 = r * 64 + (side ? d.W - 1 : 0);
                } else {
                    const int c = px - 12;
                    if (c < 64) { s0 = rA * 64 + c; s1 = rB * 64 + c; }
                    else {
                        const int ca = c == 64 ? 2 : d.W - 3, cb = c == 64 ? 0 : d.W - 1;
                        s0 = rA * 64 + ca; s1 = rB * 64 + ca; s2 = rA * 64 + cb; s3 = rB * 64 + cb;
                    }
                }
                float f[8], g[8];
                ld(s0, k, f); ld(s1, k, g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g[e];
                if (s2 >= 0) {
                    ld(s2, k, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    ld(s3, k, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                }
                // slot: the low 3 bits of the pixel it stands in for (column 2 / W-3 of a line), so the fragment reads keep their bank pattern
                const int so = t.NS + (px < 12 ? 8 * (px % 6) + (px >= 6 ? 5 : 2) : 48 + (px - 12 < 64 ? px - 12 : (px - 12 == 64 ? 66 : 69)));
                *reinterpret_cast<u32x4_t*>(sb + so * 128 + (((k + (so & 6)) & 7) << 4)) = f32_to_chunk<T>(f);
            }
        }
    };

    // The same mirror pixels for the phased schedule, where they sit in the R phase of a chunk's last step and every instruction of
    // that phase counts (mirror_fix above is ~320 instructions of index arithmetic per wave; this form ~45): W is 64 (host-checked), so a
    // thread's item - pixel and 16-byte chunk - and its swizzled byte offsets inside a strip line are functions of the thread id alone:
    //   threads 0-95: column mirror (line r = px % 6, side = px / 6, chunk k) of the strip's six lines;
    //   edge tiles, all 512 threads: line-mirror pixel c = tid / 8, chunk k: both source lines and the destination share the offset
    //   c * 128 + swizzle(c, k) (the slots NS + 48 + c, rA * 64 + c, rB * 64 + c all have c's low three bits);
    //   edge tiles, threads 0-15: the line mirror's own two column mirrors (four-pixel sums) - the general form, 2 pixels per tile.
    // Same sums in the same order, one rounding: bitwise the result of mirror_fix.
    // (in the K loop the column mirrors' two source chunks are requested BEFORE the step's 16 fragment reads and summed behind them -
    // `pre` / `mid`: the LDS round trip of the sources would otherwise stand alone in front of the reads)
    [[maybe_unused]] auto mirror_fix_lean = [&](int region, const Tile& t, auto&& between) {
        if constexpr (MIRROR) {
            unsigned char* sb = smem + region * REG;
            const int nrows = t.NS >> 6;
            const bool top = t.ti == 0, edge = top || t.ti == tpi - 1;
            auto sum2 = [&](int a0, int a1, int ao) {
                float f[8], g[8];
                chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + a0), f);
                chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + a1), g);
#pragma unroll
                for (int e = 0; e < 8; ++e) f[e] += g[e];
                *reinterpret_cast<u32x4_t*>(sb + ao) = f32_to_chunk<T>(f);
            };
            int tid_m = tid;
            asm volatile("" : "+v"(tid_m));                     // opaque: nothing derived from it may be hoisted out of the K loop (registers)
            const int k = tid_m & 7, px = tid_m >> 3;
            {
                const bool side = px >= 6;
                const int r = side ? px - 6 : px;
                const bool mine = tid_m < 96 && r < nrows;
                const int c0 = side ? d.W - 3 : 2, c1 = side ? d.W - 1 : 0, sc = 8 * r + (side ? 5 : 2);
                // every thread reads (threads without an item: chunk 0 of the region - harmless) so that the reads are not under a branch
                const int a0 = mine ? (r * 64 + c0) * 128 + (((k + (c0 & 6)) & 7) << 4) : 0;
                const int a1 = mine ? (r * 64 + c1) * 128 + (((k + (c1 & 6)) & 7) << 4) : 0;
                const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(sb + a0), v1 = *reinterpret_cast<const u32x4_t*>(sb + a1);
                between();
                if (mine) {
                    float f[8], g[8];
                    chunk_to_f32<T>(v0, f); chunk_to_f32<T>(v1, g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    *reinterpret_cast<u32x4_t*>(sb + (t.NS + sc) * 128 + (((k + (sc & 6)) & 7) << 4)) = f32_to_chunk<T>(f);
                }
            }
            if (edge) {                                                             // block-uniform
                const int rA = (top ? 2 : d.H - 3) - t.lo, rB = (top ? 0 : d.H - 1) - t.lo;
                const int base = px * 128 + (((k + (px & 6)) & 7) << 4);            // px = c < 64
                sum2(rA * 8192 + base, rB * 8192 + base, (t.NS + 48) * 128 + base);
                if (tid_m < 16) {                                                   // c = 64, 65: columns 2 + 0 / W-3 + W-1 of both lines
                    const int cc2 = tid_m >> 3;
                    const int ca = cc2 == 0 ? 2 : d.W - 3, cb = cc2 == 0 ? 0 : d.W - 1, x = cc2 == 0 ? 66 : 69;
                    auto ad = [&](int slot) -> int { return slot * 128 + (((k + (slot & 6)) & 7) << 4); };
                    float f[8], g[8];
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + ad(rA * 64 + ca)), f);
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + ad(rB * 64 + ca)), g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + ad(rA * 64 + cb)), g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(sb + ad(rB * 64 + cb)), g);
#pragma unroll
                    for (int e = 0; e < 8; ++e) f[e] += g[e];
                    *reinterpret_cast<u32x4_t*>(sb + ad(t.NS + 48 + x)) = f32_to_chunk<T>(f);
                }
            }
        }
    };

    // ---- NORM: (x - mean) * rstd, activation, one rounding - on ONE strip piece (8 rows x 128 B = one 16-byte chunk per lane), in place,
    //      by the wave that DMA'd it (its own vmcnt(0) wait orders the landed piece before these reads: no barrier needed).  Lane L holds
    //      row 8j + L/8, physical slot L%8 = channel chunk (L%8 - (row & 6)) & 7 of the 64-channel K chunk cc: the same chunk for every
    //      piece, so a lane's 8 (mean, rstd) pairs are four 16-byte reads of the LDS table.
    auto fix_piece = [&](int region, int j, const Tile& t, int tslot, int cc) {
        if constexpr (NORM) {
            const int r = 8 * j + l8;
            const int k = (ls - (l8 & 6)) & 7;
            u32x4_t* p = reinterpret_cast<u32x4_t*>(smem + region * REG + j * 1024 + lane * 16);
            const float* tb = ntab + ((long)tslot * Cin + cc * BK + k * 8) * 2;
            float f[8];
            chunk_to_f32<T>(*p, f);
            const int act = d.nrm_act; const float slope = d.nrm_slope;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const f32x4_t t4 = *reinterpret_cast<const f32x4_t*>(tb + 2 * e);
                const float v0 = (f[e] - t4[0]) * t4[1], v1 = (f[e + 1] - t4[2]) * t4[3];
                f[e] = act == UIG_ACT_RELU ? (v0 > 0.f ? v0 : 0.f) : (act == UIG_ACT_LRELU ? (v0 > 0.f ? v0 : v0 * slope) : v0);
                f[e + 1] = act == UIG_ACT_RELU ? (v1 > 0.f ? v1 : 0.f) : (act == UIG_ACT_LRELU ? (v1 > 0.f ? v1 : v1 * slope) : v1);
            }
            const u32x4_t o = f32_to_chunk<T>(f);
            if (r < t.NS) *p = o;                                                     // rows past the strip stay the DMA's zeros
            const int pix = t.lo * d.W + r;                                            // the tile's own pixels also go to h (first channel tile only)
            if (d.nrm_h != nullptr && t.n_base == 0 && r < t.NS && pix >= t.p0 && pix < min(t.p0 + BM, HoWo))
                *reinterpret_cast<u32x4_t*>(static_cast<T*>(d.nrm_h) + ((long)t.img * d.H * d.W + pix) * Cin + cc * BK + k * 8) = o;
        }
    };
    if constexpr (NORM) {
        // (mean, rstd) of every image this block will touch -> LDS (plain loads now, before any DMA is in flight: a VGPR-destination
        // load inside the K loop would make the compiler drain the DMAs in front of its use)
        for (int rr = 0; rr < NTB; ++rr) {
            const Tile tt = get_tile(rr);
            if (!tt.valid) break;                                      // block-uniform
            const float* sp = d.nrm_stats + (long)tt.img * Cin * 2;
            for (int c = tid; c < Cin * 2; c += 64 * NW) ntab[(long)rr * Cin * 2 + c] = sp[c];
        }
    }

    unsigned long long tstamp[8];
    int nst = 0;
    [[maybe_unused]] unsigned long long wsum = 0, bsum = 0;      // STAMP: per-wave sums over all K-steps: waitcnt, barrier
    [[maybe_unused]] int nstep = 0;
    auto stamp = [&]() {
        if constexpr (STAMP) {
            if (nst < 8) {
                unsigned long long tt;
                __builtin_amdgcn_sched_barrier(0);
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory");
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < 8; ++i) if (i == nst) tstamp[i] = tt;
            }
            ++nst;
        }
    };
    stamp();                                                   // 0: entry
    Tile cur = get_tile(0);
    if (!cur.valid) return;                                    // block-uniform (never taken: the grid is <= the tile count)
    __syncthreads();                                           // zero rows written (no DMA in flight yet)
    issue_first(cur, 0);
    [[maybe_unused]] unsigned long long pst0 = 0;              // STAMP, phased schedule: the prologue in parts (entry -> DMAs issued -> row table -> accumulators)
    if constexpr (STAMP && PH) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pst0) :: "memory"); wsum = pst0 - tstamp[0]; __builtin_amdgcn_sched_barrier(0); }

    // ---- per-lane row table: LDS byte address (within a strip buffer, K half 0) of the B-operand row of output pixel
    //      (tile row wm*64 + b*16 + l16) displaced by tap t; half 1 of the 128-byte row is address ^ 64.
    //      Taps form a 3x3 grid: dh depends on t / 3 only, dw on t % 3 only (checked on the host).
    // PKRT: two 16-bit entries per register, unpacked at the point of use behind a compiler barrier (18 registers instead of 36: what the
    // four-issuing-wave variants need to stay clear of spills)
    unsigned rtw[PKRT ? NTAPS * MT / 2 : NTAPS * MT];
    auto rt_get = [&](int t, int b) -> unsigned {
        const int i = t * MT + b;
        if constexpr (PKRT) {
            unsigned v = rtw[i >> 1];
            asm volatile("" : "+v"(v));
            return (i & 1) ? (v >> 16) : (v & 0xffffu);
        } else return rtw[i];
    };
    auto rt_set = [&](int t, int b, unsigned av) {
        const int i = t * MT + b;
        av &= 0xffffu;
        if constexpr (PKRT) rtw[i >> 1] = (i & 1) ? ((rtw[i >> 1] & 0xffffu) | (av << 16)) : ((rtw[i >> 1] & 0xffff0000u) | av);
        else rtw[i] = av;
    };
    if constexpr (PKRT) {
#pragma unroll
        for (int i = 0; i < NTAPS * MT / 2; ++i) rtw[i] = 0u;
    }
    int rt_ti = -1;
    // Every lane builds its own table (8 waves x 2 per SIMD on the VALU: ~8 cycles per instruction at the head of the launch), so
    // the address is kept SEPARABLE: W is a multiple of 16 (host-checked), hence the swizzle term of slot hv + wv depends on wv
    // only and address(i, j) = A_i + B_j: 3 + 3 values per pixel group, one add and one select per table entry (the first form
    // swizzled each of the 36 entries: ~1000 VALU instructions, 10k cycles of prologue by the stamps; 23k with the mirror cases).
    auto build_rt = [&](const Tile& tl, int half = -1) {        // half = 0 / 1: only the pixel groups b = 2 half, 2 half + 1 (the SIMD partner builds the others)
        const bool refl = d.pad_mode == UIG_PAD_REFLECT;
        const int ho0 = tl.p0 / d.Wo, rem0 = tl.p0 - ho0 * d.Wo;              // scalar
        auto swz = [&](int x) -> int { return (SWZ ? ((q + (x & 6)) & 7) : (q ^ ((x >> 1) & 7))) << 4; };   // x: slot index mod 16
        int dhs[3], dws[3];                                                    // scalar
#pragma unroll
        for (int i = 0; i < 3; ++i) { dhs[i] = (d.tap[3 * i] & 255) - 128; dws[i] = ((d.tap[i] >> 8) & 255) - 128; }
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            if (half >= 0 && (b >> 1) != half) continue;                        // wave-uniform
            const int pr = rem0 + wm * WM + b * 16 + l16;                       // < Wo + 256
            const int dho = (pr * d.wo_magic) >> 20;                            // pr / Wo (exact: host-checked range)
            const int ho = ho0 + dho, wo = pr - dho * d.Wo;
            const bool pv = tl.p0 + wm * WM + b * 16 + l16 < HoWo;
            // source pixel of tap (i, j) = strip slot hv_i + wv_j; out of range (zero padding) -> one of the eight zero slots
            // CAP .. CAP+7, the one with the column's low 3 bits: a slot's bank group is a function of (slot & 7), so a fragment read
            // keeps the conflict-free bank pattern of 16 consecutive pixels whether or not some of its lanes are padding
            int A[3], Bj[3], Z[3];
            bool hok[3], wok[3];
            [[maybe_unused]] int CA[3], TA[3];
            [[maybe_unused]] bool ra[3], ca[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const int hi_ = ho + dhs[i];
                hok[i] = pv & (refl | ((unsigned)hi_ < (unsigned)d.H));
                A[i] = ((refl ? reflect_idx(hi_, d.H) : hi_) - tl.lo) * d.W * 128;
                const int wi_ = wo + dws[i];
                wok[i] = refl | ((unsigned)wi_ < (unsigned)d.W);
                const int wvi = refl ? reflect_idx(wi_, d.W) : wi_;
                const int sw = swz(wvi);
                Bj[i] = wvi * 128 + sw;
                Z[i] = (CAP + (wvi & 7)) * 128 + sw;
                if constexpr (MIRROR) {                                         // taps that read a mirror pixel instead (kernel comment)
                    ra[i] = (ho == 1 && dhs[i] == 1) || (ho == d.H - 2 && dhs[i] == -1);
                    CA[i] = (tl.NS + 8 * (hi_ - tl.lo)) * 128;                  // column mirrors of the tap's source line
                    const bool cl = wo == 1 && dws[i] == 1, cr = wo == d.W - 2 && dws[i] == -1;
                    ca[i] = cl | cr;
                    const int c = cr ? 5 : 2;
                    if (ca[i]) Bj[i] = c * 128 + swz(c);                        // behind CA_i instead of A_i
                    const int x = cl ? 66 : (cr ? 69 : wvi);
                    TA[i] = (tl.NS + 48 + x) * 128 + swz(x);                    // line mirror (and its own column mirrors)
                }
            }
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const int i = t / 3, j = t % 3;
                int a;
                if constexpr (MIRROR) a = ra[i] ? TA[j] : (ca[j] ? CA[i] : A[i]) + Bj[j];
                else a = A[i] + Bj[j];
                rt_set(t, b, (unsigned)((NOZ || (hok[i] & wok[j])) ? a : Z[j]));      // NOZ: host-checked that no tap leaves the image
            }
        }
    };

    const int wswz = (l16 >> 1) & 7;
    int par = 0;                                               // region holding chunk 0 of the current tile
    for (int r = 0;; ++r) {
        const Tile nxt = get_tile(r + 1);
        if (cur.ti != rt_ti) {
            if constexpr (PH && !PKRT) {
                // Waves w and w + 4 (SIMD partners, the same 64 pixel rows) hold the SAME table, and building it is ~1000 dependent
                // instructions that one wave executes at ~5 cycles apiece whether or not its partner does the same beside it (stamps:
                // 5.5 k cycles of the block's prologue).  Each wave builds HALF of it (two of the four 16-pixel groups) and hands that
                // half to its partner through the LDS: two 16-bit entries per dword, 9 x 256 B per half, in wave w's own epilogue-scratch
                // area of strip 1 (free here: no DMA of the tile's first chunk or weight tiles goes there, the waves' previous
                // epilogues are over).
                unsigned char* xb = smem + REG + wm * SCRW;
                build_rt(cur, wn);
                {
                    unsigned pk[NTAPS];
#pragma unroll
                    for (int t = 0; t < NTAPS; ++t) pk[t] = wn == 0 ? (rtw[4 * t] | (rtw[4 * t + 1] << 16)) : (rtw[4 * t + 2] | (rtw[4 * t + 3] << 16));
                    unsigned char* mine = xb + wn * 2560;
                    *reinterpret_cast<u32x4_t*>(mine + lane * 16) = u32x4_t{pk[0], pk[1], pk[2], pk[3]};
                    *reinterpret_cast<u32x4_t*>(mine + 1024 + lane * 16) = u32x4_t{pk[4], pk[5], pk[6], pk[7]};
                    *reinterpret_cast<unsigned*>(mine + 2048 + lane * 4) = pk[8];
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                {
                    const unsigned char* theirs = xb + (wn ^ 1) * 2560;
                    const u32x4_t v0 = *reinterpret_cast<const u32x4_t*>(theirs + lane * 16), v1 = *reinterpret_cast<const u32x4_t*>(theirs + 1024 + lane * 16);
                    const unsigned v2 = *reinterpret_cast<const unsigned*>(theirs + 2048 + lane * 4);
#pragma unroll
                    for (int t = 0; t < NTAPS; ++t) {
                        const unsigned v = t < 4 ? v0[t] : (t < 8 ? v1[t - 4] : v2);
                        if (wn == 0) { rtw[4 * t + 2] = v & 0xffffu; rtw[4 * t + 3] = v >> 16; }
                        else { rtw[4 * t] = v & 0xffffu; rtw[4 * t + 1] = v >> 16; }
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the area is this wave pair's until the next epilogue)
                }
            } else build_rt(cur);
            rt_ti = cur.ti;
        }
        if constexpr (STAMP && PH) { if (r == 0) { __builtin_amdgcn_sched_barrier(0); unsigned long long tt; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); bsum = tt - tstamp[0]; __builtin_amdgcn_sched_barrier(0); } }
        if constexpr (MIRROR) {
            if (r == 0 && !(d.mirror & 4)) {                                       // the block's first chunk: nothing ran in front of it to hide this behind
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                if constexpr (PH) mirror_fix_lean(0, cur, [] {}); else mirror_fix(0, cur);
            }
        }
        if constexpr (NORM) {
            if (r == 0) {                                                          // the block's first chunk: every wave normalises the pieces it issued itself
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                for (int j = wave; 8 * j < cur.NS; j += NW) fix_piece(0, j, cur, 0, 0);
            }
        }
        const float* bias = cur.g2 ? d.bias2 : bias1;

        f32x4_t acc[NT][MT];
        strip_init_acc<MT, NT, WN>(acc, bias, d.Nrows, cur.n_base, wn, lane);

        stamp();                                               // 1 / 4: K loop starts
        if constexpr (STAMP && PH) {                            // (+ the wait for the block's first DMAs, which the product build does behind this point)
            if (r == 0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); unsigned long long tt; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt) :: "memory"); nstep = (int)(tt - tstamp[0]); __builtin_amdgcn_sched_barrier(0); }
        }
        [[maybe_unused]] u32x4_t xn[MT];                       // DM == 5: the NEXT step's strip fragments of K half 0, read behind this step's MFMAs
        if constexpr (DM == 9) {
        // ---- the phased schedule (round 4; after the guide's 256^2 eight-phase GEMM template).  A K-step (one tap, 64 channels) of a wave is
        //   R: 16 fragment reads, the step's three DMAs (strip piece, two weight pieces), the counted DMA wait, lgkmcnt(0) | barrier |
        //   M: 32 MFMAs | barrier.
        // Waves 4-7 (group B, the SIMD partners of waves 0-3 = group A) run ONE BARRIER BEHIND group A, so in every interval between two
        // barriers one wave of each SIMD reads fragments and issues DMAs while the other owns the matrix pipe.  What the probes say about this
        // skeleton on gfx950 (scripts/probes/mfma_coissue.hip): with R = 16 ds_read_b128 (+ 16 VALU adds, + 3 LDS-DMA pieces) a step takes
        // 1025 cycles for its 1024 of MFMA - barriers, the partner's reads and DMA issue cost the matrix pipe nothing - but 20 SCALAR
        // instructions in R make it 1385, the DMAs between the MFMAs of the M phase 1185, all eight waves in step (R | barrier | M) 1357.
        // Hence the LEAN R phase: running offsets advanced by one instruction each (explicit asm: left to itself the optimiser turns a
        // running offset of an unrolled loop back into nine recomputations from the tap table), the wave's second weight piece through its
        // own lane-offset register instead of a second scalar offset, the strip piece's validity by a per-lane running row index (VALU),
        // the last chunk of a tile as its own instance of the step body so that everything depending on "last chunk" is decided at compile
        // time.  Measured (in-kernel stamps, scripts/stamp_lean.py): 1182-1190 cycles per K-step against ~1650 for round 3's loop; the
        // launch is faster by 6-7 % only - the chip runs this kernel AT ITS POWER LIMIT (rocm-smi: 1350-1360 W of 1400 under either schedule)
        // and lowers the clock as the matrix pipe fills (2.24 -> 2.07 GHz): DESIGN.md 5.000.
        // (Tried on the way, all bitwise equal: phases of one K half - 16 MFMAs between barriers like the template - the same time as
        // whole steps; the next half's fragments read between the wave's own MFMAs, one barrier per step: no gain; the DMA issue moved
        // into the M phase: slower.)
        // Hazards (intervals I_k between barriers; step s: A reads in I_2s, B in I_2s+1):
        //   * weights of step s+2 go to stage (s+2) % 3 = the stage of step s-1 (a tile has 36 = 0 mod 3 steps: the stage of a tap is
        //     tap % 3), last read by B in I_2s-1 and RETIRED there (lgkmcnt(0) in front of the barrier that ends the interval); issued by
        //     A in I_2s, by B in I_2s+1.
        //   * a wave's DMAs of step s are waited for in step s+1's R phase, AFTER that step's own DMAs are issued, by vmcnt(3 | 2) = their
        //     number (the same in every wave: zero fills stand in for pieces that do not exist): A in I_2s+2, B in I_2s+3; the barrier that
        //     ends I_2s+3 publishes them; first read by A in I_2s+4.
        //   * strip pieces of chunk c+1 (taps 0-6 of chunk c; piece 55 = a zero fill of the strip's zero rows) go to the other strip
        //     region, last read in chunk c-1; landed and published by the barrier behind tap 7's R phases; the mirror pixels are written
        //     in tap 8's R phase and retired there.
        //   * end of tile: group B skips the closing barrier of the last step (nobody waits for those MFMAs) - the groups are level
        //     again (equal barrier counts), the epilogues start without a wait; the scratch is [strip 1][W2]: the last chunk's strip and
        //     the last tap's stage, neither a prefetch target (the next tile's chunk 0 and taps 0, 1 go to strip 0, W0, W1).  The number
        //     of chunks is even (host-checked), so a tile's last chunk is always in strip 1.
        if (r == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the block's first strip chunk and weight tiles
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                 // ... and every wave's epilogue scratch reads / zero-row / mirror-pixel writes
        __builtin_amdgcn_s_barrier();
        if (wn == 1) __builtin_amdgcn_s_barrier();                          // group B: one barrier behind
        const int inc_tap = __builtin_amdgcn_readfirstlane(d.wt_b * Cin * (int)sizeof(T));
        const int inc_wrap = __builtin_amdgcn_readfirstlane(BK * (int)sizeof(T) - 8 * inc_tap);            // tap 8 of chunk c -> tap 0 of chunk c + 1
        const int k64 = __builtin_amdgcn_readfirstlane(64 * Cin * (int)sizeof(T));                          // 8 strip pieces further
        int w_run = __builtin_amdgcn_readfirstlane((cur.n_base * d.ldw + (d.wt_a + 2 * d.wt_b) * Cin) * (int)sizeof(T));      // tap 2, chunk 0
        const int w_next0 = __builtin_amdgcn_readfirstlane((nxt.n_base * d.ldw + d.wt_a * Cin) * (int)sizeof(T));             // the next tile's tap 0, chunk 0
        const __amdgpu_buffer_rsrc_t rs_cur = cur.g2 ? rsw2 : rsw1, rs_nxt = nxt.g2 ? rsw2 : rsw1;
        const unsigned wvl1 = wvl0 + (unsigned)(NW * 8 * d.ldw * (int)sizeof(T));                           // the wave's second piece: 64 rows further
        auto chunk = [&](auto lastc, int cc) {
            constexpr bool LASTC = decltype(lastc)::value;
            const int pc = par ^ (cc & 1);
            const unsigned char* sx = smem + pc * REG;
            const bool pre_next = LASTC && nxt.valid;
            const bool s_on = !LASTC || pre_next;
            const unsigned s_base = !LASTC ? strip_base(cur, cc + 1) : strip_base(nxt, 0);
            const int s_NS = s_on ? (!LASTC ? cur.NS : nxt.NS) : 0;            // 0: zero fills
            unsigned svrun = svl + (unsigned)(wave * 8 * Cin * (int)sizeof(T));  // lane offset of this wave's piece of the step (slot t * 8 + wave)
            int rowrun = wave * 8 + l8;                                         // ... and its strip row
            asm volatile("" : "+v"(svrun), "+v"(rowrun));
            const int sdst0 = __builtin_amdgcn_readfirstlane((pc ^ 1) * REG + wave * 1024);
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {
                const unsigned char* sw = smem + wst_off(t % 3) + (wn * WN + l16) * 128;
                // ---- R
                u32x4_t xf[2][MT], wf[2][NT];
                // the step's 16 fragment reads in two parts (14 + 2): the mirror-pixel step keeps its two source chunks in the registers of the
                // last two fragments while the first 14 are in flight (no registers of its own: the kernel sits at 251 of 256)
                auto frag_reads = [&](int part) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int co = ((q + 4 * h) ^ wswz) << 4;
#pragma unroll
                        for (int a = 0; a < NT; ++a)
                            if ((h == 1 && a == NT - 1) == (part == 1)) wf[h][a] = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co);
#pragma unroll
                        for (int b = 0; b < MT; ++b)
                            if ((h == 1 && b == MT - 1) == (part == 1)) xf[h][b] = *reinterpret_cast<const u32x4_t*>(sx + (rt_get(t, b) ^ (unsigned)(h << 6)));
                    }
                    __builtin_amdgcn_sched_barrier(0);
                };
                bool fixed = false;
                if constexpr (MIRROR) {
                    if (t == NTAPS - 1 && s_on && !(d.mirror & 2)) { mirror_fix_lean(pc ^ 1, LASTC ? nxt : cur, [&] { frag_reads(0); }); fixed = true; }
                }
                if (!fixed) frag_reads(0);
                frag_reads(1);
                __builtin_amdgcn_sched_barrier(0);
                if (t <= NTAPS - 3) {
                    const unsigned off = rowrun < s_NS ? svrun : 0xFFFFFFFFu;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsx, (__attribute__((address_space(3))) void*)((lds_ptr_t)smem + sdst0 + t * 8192), 16, (int)off, (int)s_base, 0, 0);
                    asm volatile("v_add_u32 %0, %2, %0\n\tv_add_u32 %1, 64, %1" : "+v"(svrun), "+v"(rowrun) : "s"(k64));
                }
                {
                    const bool NEXT_TILE = LASTC && t >= NTAPS - 2;            // taps 0, 1 of the next tile (constant after unrolling)
                    const int st = (t + 2) % 3;
                    const unsigned vo0 = (NEXT_TILE && !pre_next) ? 0xFFFFFFFFu : wvl0, vo1 = (NEXT_TILE && !pre_next) ? 0xFFFFFFFFu : wvl1;
                    lds_ptr_t dst = (lds_ptr_t)smem + wst_off(st) + wave * 1024;
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT_TILE ? rs_nxt : rs_cur, (__attribute__((address_space(3))) void*)dst, 16, (int)vo0, w_run, 0, 0);
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(NEXT_TILE ? rs_nxt : rs_cur, (__attribute__((address_space(3))) void*)(dst + NW * 1024), 16, (int)vo1, w_run, 0, 0);
                    if (t == NTAPS - 3) {
                        if constexpr (LASTC) w_run = w_next0;
                        else asm volatile("s_add_u32 %0, %0, %1" : "+s"(w_run) : "s"(inc_wrap) : "scc");
                    } else asm volatile("s_add_u32 %0, %0, %1" : "+s"(w_run) : "s"(inc_tap) : "scc");
                }
                if (t <= NTAPS - 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                // ---- M
                __builtin_amdgcn_s_setprio(1);
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int a = 0; a < NT; ++a)
#pragma unroll
                        for (int b = 0; b < MT; ++b) MmaS<T>::run(wf[h][a], xf[h][b], acc[a][b]);
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(LASTC && t == NTAPS - 1 && wn == 1)) __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int cc = 0; cc + 1 < ncc; ++cc) chunk(std::false_type{}, cc);
        chunk(std::true_type{}, ncc - 1);
        } else
        for (int cc = 0; cc < ncc; ++cc) {
            const int pc = par ^ (cc & 1);                     // region of this chunk's strip; weight stage of step t: pc ^ (t & 1)
            const unsigned char* sx = smem + pc * REG;
            const bool last_cc = cc + 1 == ncc;
            const bool pre_next = XPREF && last_cc && nxt.valid;
            // next strip chunk streamed into the other region behind this chunk's MFMAs: chunk cc+1 of this tile, or chunk 0 of the next
            const bool s_on = !last_cc || pre_next;
            const unsigned s_base = !last_cc ? strip_base(cur, cc + 1) : strip_base(nxt, 0);
            const int s_NS = !last_cc ? cur.NS : nxt.NS;
#pragma unroll
            for (int t = 0; t < NTAPS; ++t) {                   // fully unrolled: the row table is statically indexed
                // this wave's DMAs (weight tile of this step, strip pieces) have landed and its fragment reads of the previous
                // step are complete (the DMAs issued below overwrite that step's weight stage) ...
                [[maybe_unused]] unsigned long long tw0 = 0, tw1 = 0, tw2 = 0;
                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tw0) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
                if constexpr (LGK) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if constexpr (STAMP) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tw1) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
                __builtin_amdgcn_s_barrier();                   // ... and everyone else's
                if constexpr (STAMP) {                          // diagnostic: cycles this wave spent in the step's waitcnt / in its barrier
                    __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tw2) :: "memory"); __builtin_amdgcn_sched_barrier(0);
                    wsum += tw1 - tw0; bsum += tw2 - tw1; ++nstep;
                }
                const bool last_t = t + 1 == NTAPS;
                const bool w_on = !(last_t && last_cc) || pre_next;
                const bool w_g2 = (last_t && last_cc) ? nxt.g2 : cur.g2;
                const unsigned w_so = !last_t ? w_base(cur, t + 1, cc) : (!last_cc ? w_base(cur, 0, cc + 1) : w_base(nxt, 0, 0));
                const int w_reg = pc ^ ((t + 1) & 1);
                constexpr int NK = WPI + NW / NISS;             // DMA instructions of an issuing wave per step
                auto dma_piece = [&](int k) {                   // k < WPI: weight piece k of the next step; else strip piece k - WPI of the next chunk
                    if (k < WPI) { if (w_on) issue_w1(k, w_g2, w_so, w_reg); }
                    else {
                        const int slot = t * NW + wave + NISS * (k - WPI);
                        if (s_on && slot < PIECES && 8 * slot < s_NS) issue_strip_piece(slot, s_base, s_NS, pc ^ 1);
                    }
                };
                auto dma = [&](int which) {                     // 0 / 1: the two halves of the next weight tile, 2: this wave's strip pieces
                    if (wave < NISS) {
                        if (which < 2) {
#pragma unroll
                            for (int i = 0; i < WPI / 2; ++i) dma_piece(which * (WPI / 2) + i);
                        } else {
#pragma unroll
                            for (int k = WPI; k < NK; ++k) dma_piece(k);
                        }
                    }
                };
                if constexpr (DM == 0 || DM == 4 || DM == 5) { dma(0); dma(1); dma(2); }
                if constexpr (MIRROR) {
                    // last step of the chunk: the next chunk's strip (issued in steps 0-6) is complete and published by this step's
                    // barrier; its mirror pixels are written HERE, at the top of the step, while the matrix pipe still works off the
                    // previous step's MFMAs (at the end of the step the round trip through the LDS sat exposed in front of the next
                    // barrier: +2.5k cycles per tile by the stamps), and are published by the next step's barrier
                    if (last_t && s_on && !(d.mirror & 2)) mirror_fix(pc ^ 1, last_cc ? nxt : cur);
                }

                const unsigned char* sw = smem + (pc ^ (t & 1)) * REG + SBUF + (wn * WN + l16) * 128;
                if constexpr (DM == 5) {
                    // the strip chunk is resident for all nine steps: only the WEIGHT fragments need this step's barrier.  The strip
                    // fragments of half 0 were read at the end of the previous step, behind its MFMAs, so the first MFMA of the step
                    // waits for four reads instead of eight (and the pipe is not empty while the others arrive)
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        u32x4_t xf[MT], wf[NT];
                        const int co = ((q + 4 * h) ^ wswz) << 4;
#pragma unroll
                        for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co);
                        if (h == 0 && !(t == 0 && (MIRROR || NORM || cc == 0))) {
#pragma unroll
                            for (int b = 0; b < MT; ++b) xf[b] = xn[b];
                        } else {
#pragma unroll
                            for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(sx + (rt_get(t, b) ^ (unsigned)(h << 6)));
                        }
#pragma unroll
                        for (int a = 0; a < NT; ++a)
#pragma unroll
                            for (int b = 0; b < MT; ++b) MmaS<T>::run(wf[a], xf[b], acc[a][b]);
                        if constexpr (NORM) {
                            // behind the first half's MFMAs: this wave normalises the strip piece it DMA'd in the PREVIOUS step (landed: the
                            // wait at the top of this step) - one 16-byte chunk per lane and step, spread over the chunk's steps 1..7
                            if (h == 0 && t >= 1) {
                                const int pslot = (t - 1) * NW + wave;
                                if (s_on && pslot < PIECES && 8 * pslot < s_NS)
                                    fix_piece(pc ^ 1, pslot, last_cc ? nxt : cur, last_cc ? r + 1 : r, last_cc ? 0 : cc + 1);
                            }
                        }
                    }
                    if (!last_t) {
#pragma unroll
                        for (int b = 0; b < MT; ++b) xn[b] = *reinterpret_cast<const u32x4_t*>(sx + rt_get((t + 1) % NTAPS, b));
                    } else if (!MIRROR && !NORM && !last_cc) { // the next chunk's strip is complete since step 7's barrier (mirror / norm kernel: its
                                                               // mirror pixels are only published by the NEXT barrier - fresh reads there)
                        const unsigned char* sxn = smem + (pc ^ 1) * REG;
#pragma unroll
                        for (int b = 0; b < MT; ++b) xn[b] = *reinterpret_cast<const u32x4_t*>(sxn + rt_get(0, b));
                    }
                } else if constexpr (DM == 3) {                 // all 16 fragment reads of the step first, then the DMAs, then 32 MFMAs
                    u32x4_t xf[2][MT], wf[2][NT];
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
#pragma unroll
                        for (int b = 0; b < MT; ++b) xf[h][b] = *reinterpret_cast<const u32x4_t*>(sx + (rt_get(t, b) ^ (unsigned)(h << 6)));
                        const int co = ((q + 4 * h) ^ wswz) << 4;
#pragma unroll
                        for (int a = 0; a < NT; ++a) wf[h][a] = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    dma(0); dma(1); dma(2);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int h = 0; h < 2; ++h)
#pragma unroll
                        for (int a = 0; a < NT; ++a)
#pragma unroll
                            for (int b = 0; b < MT; ++b) MmaS<T>::run(wf[h][a], xf[h][b], acc[a][b]);
                } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    u32x4_t xf[MT], wf[NT];
#pragma unroll
                    for (int b = 0; b < MT; ++b) xf[b] = *reinterpret_cast<const u32x4_t*>(sx + (rt_get(t, b) ^ (unsigned)(h << 6)));
                    const int co = ((q + 4 * h) ^ wswz) << 4;
#pragma unroll
                    for (int a = 0; a < NT; ++a) wf[a] = *reinterpret_cast<const u32x4_t*>(sw + a * 16 * 128 + co);
                    if constexpr (DM == 2) { if (h == 0) { __builtin_amdgcn_sched_barrier(0); dma(0); dma(1); dma(2); __builtin_amdgcn_sched_barrier(0); } }
                    if constexpr (DM == 4) __builtin_amdgcn_s_setprio(1);
#pragma unroll
                    for (int a = 0; a < NT; ++a) {
#pragma unroll
                        for (int b = 0; b < MT; ++b) MmaS<T>::run(wf[a], xf[b], acc[a][b]);
                        if constexpr (DM == 1) {                // one DMA piece behind every 4th MFMA group of the step
                            if (h == 0 && a == 0) { __builtin_amdgcn_sched_barrier(0); dma(0); __builtin_amdgcn_sched_barrier(0); }
                            if (h == 0 && a == 2) { __builtin_amdgcn_sched_barrier(0); dma(1); __builtin_amdgcn_sched_barrier(0); }
                            if (h == 1 && a == 0) { __builtin_amdgcn_sched_barrier(0); dma(2); __builtin_amdgcn_sched_barrier(0); }
                        }
                    }
                    if constexpr (DM == 4) __builtin_amdgcn_s_setprio(0);
                }
                }
            }
        }
        stamp();                                               // 2 / 5: K loop done
        const int pl = par ^ ((ncc - 1) & 1);                  // region of the last chunk == weight stage of the last step (9 taps: odd)

        // ---- epilogue: scratch = the region the last K-step has just finished with (the prefetch went to the other one)
        if constexpr (!PH) {                                    // (phased schedule: every read of region pl was retired in front of the last phase's first barrier)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                           // every wave is done reading region pl (raw: the prefetch DMAs stay in flight)
        }
        unsigned char* scratch = smem + (XPREF ? pl * REG : 0) + wave * SCRW;
        // The epilogue's per-lane constants (LDS scratch addresses, row offsets, ...) all derive from the lane id.  Opaque to the
        // optimiser here, or it hoists them out of the tile loop and carries ~60 registers through the K loop (seen: 73 VGPRs
        // spilled to scratch, reloaded one `s_waitcnt vmcnt(0)` at a time behind the epilogue's own global stores).
        int lane_e = lane;
        asm volatile("" : "+v"(lane_e));
        if constexpr (STAMP) {                                  // + after the barrier / after the tile's LDS writes
            stamp();
            strip_epilogue<T, MT, NT, WM, WN, !MIRROR>(acc, scratch, d, y, cur.img, cur.p0, wm, wn, cur.n_base, lane_e, stamp);
        } else
        strip_epilogue<T, MT, NT, WM, WN, !MIRROR, NoMidHook, !MIRROR || BST>(acc, scratch, d, y, cur.img, cur.p0, wm, wn, cur.n_base, lane_e);
        stamp();                                               // 3 / 6: epilogue issued
        if (!nxt.valid) break;
        if constexpr (XPREF) {
            // the scratch covered region pl's zero row: its owner restores it (read again from the next tile's chunk 1 on,
            // many barriers from here)
            if constexpr (!NOZ)
            if (wave == ZW) *reinterpret_cast<u32x4_t*>(smem + pl * REG + CAP * 128 + lane * 16) = u32x4_t{0u, 0u, 0u, 0u};
            par = pl ^ 1;
        } else {
            // fp32 tiles: the scratch spans both regions, so the next tile starts like the first one
            __syncthreads();
            zero_rows();
            par = 0;
            issue_first(nxt, 0);
        }
        cur = nxt;
    }
    // round 4: no finalize launch behind this one - one arrival ticket per tile and image once ALL of this block's tiles are out (nothing
    // is added to a tile's epilogue; the block's stores must drain before it exits anyway); the block that draws an image's last ticket
    // reduces that image's statistics slabs (uig_common.h, UigFin).  All LDS is free here: word 0 carries the "I am last" flag.
    if (d.fin.tickets != nullptr || d.bfin.tickets != nullptr) {          // launch-uniform
        int tid_e = threadIdx.x;                                           // opaque: nothing derived from it here may be hoisted above the tile loop
        asm volatile("" : "+v"(tid_e));
        for (int r = 0;; ++r) {
            const Tile t = get_tile(r);
            if (!t.valid) break;                                           // block-uniform
            if (d.fin.tickets != nullptr) uig_fin_arrive<64 * NW>(d.fin, t.img, 1u, reinterpret_cast<unsigned*>(smem), tid_e);
            if (d.bfin.tickets != nullptr) uig_fin_arrive<64 * NW>(d.bfin, t.img, 1u, reinterpret_cast<unsigned*>(smem), tid_e);
        }
    }
    if constexpr (STAMP) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamp();                                               // last: stores drained
        if (lane == 0 && d.dbg != nullptr) {
            unsigned long long* o = d.dbg + ((long)blockIdx.x * NW + wave) * 8;
            for (int i = 0; i < 8; ++i) o[i] = i < nst ? tstamp[i] : 0ull;
            unsigned long long* o2 = d.dbg + (long)gridDim.x * NW * 8 + ((long)blockIdx.x * NW + wave) * 4;      // second table behind the stamps
            o2[0] = wsum; o2[1] = bsum; o2[2] = (unsigned long long)nstep; o2[3] = 0ull;
        }
    }
}

static int g_pk_dm = 0;        // tuning hook: variant of the bf16 kernel (see the switch in uig_launch_strip_pk)
static int g_pk_grid = 0;      // tuning hook: persistent grid size (0 = one block per CU)
extern "C" void uig_debug_set_strip_pk(int dm, int grid) { g_pk_dm = dm; g_pk_grid = grid; }
static long g_pk_phased = 0;   // launches that took the phased schedule (tests assert the variant they mean to cover)
extern "C" long uig_debug_strip_pk_phased_count(void) { return g_pk_phased; }

static int device_cus() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 256;
        return v;
    }();
    return n;
}

template <typename T, int CAP, int DM, int SWZ = 1, bool LGK = true, bool STAMP = false, bool MIRROR = false, bool NORM = false, bool NOZ = false, int NISS = 8, bool PKRT = false, bool BST = false>
static int launch_pk(const void* x, const void* wp, const float* bias, void* y, const StripDesc& d, int ntiles, hipStream_t s) {
    const size_t smem = 2 * ((size_t)(CAP + (NOZ ? 0 : 8)) * 128 + 128 * 128) + (NORM ? (size_t)4 * 256 * 8 : 0) + (DM >= 6 && DM <= 9 ? 128 * 128 : 0);      // NORM: (mean, rstd) of 4 images x <= 256 channels; DM 6: a third weight stage
    auto kern = conv_strip_pk_kernel<T, CAP, DM, SWZ, LGK, STAMP, MIRROR, NORM, NOZ, NISS, PKRT, BST>;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(kern), smem);
        if (e != hipSuccess) return uig_set_error((int)e, "conv_strip_pk: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    const int grid = std::min(ntiles, g_pk_grid > 0 ? g_pk_grid : device_cus());
    hipLaunchKernelGGL(kern, dim3(grid), dim3(512), smem, s, (const T*)x, (const T*)wp, bias, (T*)y, d);
    UIG_LAUNCH_CHECK("uig_conv_gather(strip, persistent)");
    return 0;
}

// 1 if the persistent kernel can take this 256x128-tile launch (d filled by uig_try_conv_strip): taps must form a 3x3 grid
// (dh a function of t / 3, dw of t % 3) and the in-tile division by Wo must be exact in 20-bit fixed point.
// need256 in 449..512: only the no-zero-row variant (uig_strip_pk_wide_ok) can take it
bool uig_strip_pk_ok(const StripDesc& d, int need256) {
    if (need256 > 512 || d.Wo > 512 || d.W % 16 != 0) return false;      // W % 16: the row table's separable swizzle (build_rt)
    for (int t = 0; t < 9; ++t)
        if ((d.tap[t] & 255) != (d.tap[3 * (t / 3)] & 255) || ((d.tap[t] >> 8) & 255) != ((d.tap[t % 3] >> 8) & 255)) return false;
    return true;
}

int uig_launch_strip_pk(const void* x, const void* wp, const float* bias, void* y, StripDesc d, int dtype, hipStream_t s) {
    d.wo_magic = ((1 << 20) + d.Wo - 1) / d.Wo;             // (pr * magic) >> 20 == pr / Wo for pr < Wo + 256 <= 768 (pr * (magic * Wo - 2^20) < 2^20)
    const int tpi = (d.Ho * d.Wo + 255) / 256;
    const int ntiles = d.B * tpi * (d.Nrows / 128);
    {   // (n * ceil(2^32 / dv)) >> 32 == n / dv while n * dv < 2^32: n < tiles (bid, mtile) or pixels of a map (p0, p_last) - checked here
        auto mg = [](unsigned dv) -> unsigned long long { return ((1ull << 32) + dv - 1) / dv; };
        const unsigned ntn_ = (unsigned)(d.Nrows / 128);
        if ((unsigned long long)ntiles * std::max(ntn_, (unsigned)tpi) >= (1ull << 32) || (unsigned long long)d.Ho * d.Wo * d.Wo >= (1ull << 32))
            return uig_set_error(-1, "conv_strip_pk: launch too large for the tile index arithmetic (%d tiles, %d x %d map)", ntiles, d.Ho, d.Wo);
        d.mg_ntn = mg(ntn_); d.mg_tpi = mg((unsigned)tpi); d.mg_wo = mg((unsigned)d.Wo);
    }
    if (d.wide512) {                  // 449..512 strip rows: the no-zero-row variant (host-checked: reflection padding, whole tiles, plain forward)
        if (dtype != UIG_BF16) return uig_set_error(-1, "conv_strip_pk: the 512-row strip is a bf16 path");
        return launch_pk<bf16_t, 512, 5, 1, true, false, false, false, true>(x, wp, bias, y, d, ntiles, s);
    }
    if (d.nrm_stats != nullptr) {     // the launch applies the InstanceNorm in front of it to its own input strip (uig_strip_pk_norm_ok)
        if (dtype != UIG_BF16 || d.mirror) return uig_set_error(-1, "conv_strip_pk: the norm strip is a bf16 forward path");
        return launch_pk<bf16_t, 448, 5, 1, true, false, false, true>(x, wp, bias, y, d, ntiles, s);
    }
    if (dtype == UIG_BF16) {
        // round 4: the phased schedule (DM 9, the default where it applies; g_pk_dm = 5 selects round 3's schedule for A/B): an even number of
        // 64-channel chunks, at most 440 strip rows (the third weight stage's LDS), weight-tap index affine in the tap (running offset)
        d.wt_a = d.tap[0] >> 16; d.wt_b = (d.tap[1] >> 16) - (d.tap[0] >> 16);
        bool wt_affine = true;
        for (int t = 0; t < 9; ++t) wt_affine = wt_affine && (d.tap[t] >> 16) == d.wt_a + d.wt_b * t;
        const bool ph_ok = (d.Cin / 64) % 2 == 0 && d.need_rows <= 440 && d.bst_partial == nullptr && wt_affine;
        if (ph_ok && (g_pk_dm == 0 || g_pk_dm == 9) && (d.dbg == nullptr || !d.mirror)) ++g_pk_phased;
        if (ph_ok && (g_pk_dm == 0 || g_pk_dm == 9)) {
            if (d.dbg != nullptr && !d.mirror) return launch_pk<bf16_t, 440, 9, 1, true, true, false, false, false, 8, false>(x, wp, bias, y, d, ntiles, s);   // coarse stamps (scripts/stamp_lean.py)
            if (d.dbg == nullptr) {
                if (d.mirror) return launch_pk<bf16_t, 440, 9, 1, true, false, true, false, false, 8, false>(x, wp, bias, y, d, ntiles, s);
                return launch_pk<bf16_t, 440, 9, 1, true, false, false, false, false, 8, false>(x, wp, bias, y, d, ntiles, s);
            }
        }
        if (d.mirror) {
            if (d.dbg != nullptr) return launch_pk<bf16_t, 448, 0, 1, true, true, true>(x, wp, bias, y, d, ntiles, s);
            // round 4: the variant whose epilogue also emits the statistics of the InstanceNorm backward that consumes dx
            if (d.bst_partial != nullptr) {
                if (g_pk_dm == 31) return launch_pk<bf16_t, 448, 5, 1, true, false, true, false, false, 4, false, true>(x, wp, bias, y, d, ntiles, s);   // plain row table (A/B: spills a few registers around the epilogue)
                return launch_pk<bf16_t, 448, 5, 1, true, false, true, false, false, 4, true, true>(x, wp, bias, y, d, ntiles, s);                        // packed row table: room for the epilogue's prefetches
            }
            switch (g_pk_dm) {
                case 12: return launch_pk<bf16_t, 448, 0, 1, true, false, true>(x, wp, bias, y, d, ntiles, s);
                case 20: return launch_pk<bf16_t, 448, 5, 1, true, false, true, false, false, 4, true>(x, wp, bias, y, d, ntiles, s);
                case 21: return launch_pk<bf16_t, 448, 5, 1, true, false, true, false, false, 8, true>(x, wp, bias, y, d, ntiles, s);
                case 23: return launch_pk<bf16_t, 448, 5, 1, true, false, true>(x, wp, bias, y, d, ntiles, s);                             // eight issuing waves (round 2's form)
                default: return launch_pk<bf16_t, 448, 5, 1, true, false, true, false, false, 4, false>(x, wp, bias, y, d, ntiles, s);  // four issuing waves
            }
        }
        if (d.dbg != nullptr) return launch_pk<bf16_t, 448, 5, 1, true, true, false, false, false, 4, false>(x, wp, bias, y, d, ntiles, s);      // the default schedule, stamped
        switch (g_pk_dm) {      // tuning variants (A/B in one process: scripts/bench_strip_pk.py)
            case 20: return launch_pk<bf16_t, 448, 5, 1, true, false, false, false, false, 4, true>(x, wp, bias, y, d, ntiles, s);   // four issuing waves, packed row table
            case 21: return launch_pk<bf16_t, 448, 5, 1, true, false, false, false, false, 8, true>(x, wp, bias, y, d, ntiles, s);   // eight issuing waves, packed row table
            case 23: return launch_pk<bf16_t, 448, 5, 1, true>(x, wp, bias, y, d, ntiles, s);                                           // eight issuing waves (round 2's default)
            case 2: return launch_pk<bf16_t, 448, 0, 0, true>(x, wp, bias, y, d, ntiles, s);      // XOR swizzle (for the bank-conflict counters)
            case 4: return launch_pk<bf16_t, 448, 0, 1, false>(x, wp, bias, y, d, ntiles, s);     // no lgkmcnt wait before the barrier
            case 8: return launch_pk<bf16_t, 448, 2, 1, false>(x, wp, bias, y, d, ntiles, s);     // reads of half 0, then the DMAs
            case 9: return launch_pk<bf16_t, 448, 3, 1, false>(x, wp, bias, y, d, ntiles, s);     // all reads up front, then the DMAs
            case 10: return launch_pk<bf16_t, 448, 4, 1, false>(x, wp, bias, y, d, ntiles, s);    // s_setprio around the MFMA clusters
            case 12: return launch_pk<bf16_t, 448, 0, 1, true>(x, wp, bias, y, d, ntiles, s);     // the form before the cross-barrier prefetch (A/B; also switches the mirror kernel back)
            // default: next step's strip fragments read behind this step's MFMAs (round 2) + four issuing waves (round 3)
            default: return launch_pk<bf16_t, 448, 5, 1, true, false, false, false, false, 4, false>(x, wp, bias, y, d, ntiles, s);
        }
    }
    if (d.mirror) return uig_set_error(-1, "conv_strip_pk: mirror pixels are a bf16 path");
    return launch_pk<float, 448, 0>(x, wp, bias, y, d, ntiles, s);
}

// 1 if a launch that uig_strip_pk_ok accepts can also fold the mirrored-border terms of a reflection-padded convolution's input
// gradient in the kernel (StripDesc::mirror): bf16, zero-padded transposed gather on a 64-pixel-wide map of >= 8 lines in whole
// 4-line tiles, taps at (-1, 0, +1)^2, all channels stored, no border buffer.
bool uig_strip_pk_mirror_ok(const StripDesc& d, int dtype) {
    if (dtype != UIG_BF16 || d.pad_mode != UIG_PAD_ZERO || d.W != 64 || d.Wo != 64 || d.H != d.Ho || d.H < 8 || d.H % 4 != 0) return false;
    if (d.border_add != nullptr || d.Nstore != d.Nrows || d.dh_min != -1 || d.dh_max != 1) return false;
    bool seen[3][3] = {};
    for (int t = 0; t < 9; ++t) {
        const int dh = (d.tap[t] & 255) - 128, dw = ((d.tap[t] >> 8) & 255) - 128;
        if (dh < -1 || dh > 1 || dw < -1 || dw > 1) return false;
        seen[dh + 1][dw + 1] = true;
    }
    for (auto& r : seen) for (bool b : r) if (!b) return false;
    return true;
}

// 1 if a launch that uig_strip_pk_ok accepts can also apply the InstanceNorm in front of it to its own input strip (StripDesc::nrm_*):
// bf16, input and output maps of the same size, at most 256 input channels and at most 4 tiles per persistent block (the LDS table of
// (mean, rstd) holds 4 images), no border / residual terms.
bool uig_strip_pk_norm_ok(const StripDesc& d, int dtype) {
    if (dtype != UIG_BF16 || d.H != d.Ho || d.W != d.Wo || d.Cin > 256 || d.border_add != nullptr || d.res_add != nullptr || d.mirror) return false;
    const int tpi = (d.Ho * d.Wo + 255) / 256;
    const long ntiles = (long)d.B * tpi * (d.Nrows / 128);
    const int grid = (int)std::min<long>(ntiles, g_pk_grid > 0 ? g_pk_grid : device_cus());
    return (ntiles + grid - 1) / grid <= 4;
}

// 1 if a launch whose 256-pixel tiles need 449..512 strip rows (128-pixel-wide maps: two image rows + two halo rows) can run on the
// no-zero-row variant: bf16, reflection padding with every reflected row inside the strip, whole tiles (no pixel past the image), all
// channels stored, no border / residual / mirror / norm terms.
bool uig_strip_pk_wide_ok(const StripDesc& d, int dtype, int need256) {
    if (dtype != UIG_BF16 || need256 <= 448 || need256 > 512 || d.pad_mode != UIG_PAD_REFLECT) return false;
    if ((d.Ho * d.Wo) % 256 != 0 || d.H != d.Ho || d.W != d.Wo || d.dh_min != -1 || d.dh_max != 1) return false;
    if (d.border_add != nullptr || d.res_add != nullptr || d.bst_partial != nullptr || d.mirror || d.nrm_stats != nullptr) return false;
    return uig_strip_pk_ok(d, need256);
}
