// wgrad_head.hip — weight gradient of the generator's output convolution (7x7, stride 1, pad 3, 64 -> 3 channels) for
// gfx950, bf16.
//
//   part[s][co][(kh*7+kw)*64 + ci] = sum_{image rows i in split s} sum_j dY[i][j][co] * Xpad[i + kh][j + kw][ci]
//
// On the generic split-K kernel (wgrad.hip) this layer is the worst case: a 16-row MFMA tile holds 3 real output channels
// and the gathered operand is staged once per tap, so the 8.4 MB/image input is pulled through LDS 49 times (320 us per
// launch at 4 images, 30 TFLOP/s).  Here the roles are turned around:
//   * one K-step is ONE IMAGE ROW; a block owns one kernel row kh and stages the input row i+kh-3 once (W + 6 pixels x 64
//     channels; the 3+3 mirrored / zero halo pixels are extra LDS rows filled by the same DMA with reflected source
//     addresses, so the products stay exact);
//   * the 7 kw taps become COLUMNS of the small operand: S[r][kw*8 + co] = dY[i][r - kw][co] (r = padded pixel index, all 8
//     stored channels), built in LDS from the dY row once per K-step by seven 16-byte copies per row;
//   * D[ci][kw*8 + co] accumulates in registers over the block's rows; fp32 partial slabs as in wgrad.hip, same reduce.
// Both operands reduce over pixels (the strided NHWC index): fragments come from ds_read_b64_tr_b16, issued as inline asm
// for the reason given in wgrad_rows.hip (the compiler drains all LDS-DMA before a builtin transposing read).
#include "uig_common.h"
#include <algorithm>
#include <type_traits>

struct WgHeadDesc {
    int B, H, W, Np, pad_mode;
    int ncols;               // 49 * 64
    int rows_total, splits;
    int group_rows;          // two networks in one launch (see wgrad_rows.hip); 0 = one
    unsigned p_bytes, q_bytes;
    int imgs_max, chunks;    // all-kernel-rows form (wgrad_head7all_kernel): splits = imgs_max * chunks, split = image-in-network * chunks + chunk
    // images of network n: n1[n] images of (P, Q) from image img1[n], then n2[n] images of a second tensor pair (P2, Q2) from image
    // img2[n] (both generator passes of a step in one launch; n2 = 0: one pair)
    int n1[2], img1[2], n2[2], img2[2];
    unsigned p2_bytes, q2_bytes;
};

namespace {
constexpr int WH_CI = 64;                      // input channels (one 128-byte LDS row per pixel)
constexpr int WH_MAXW = 256;
constexpr int WH_ROWS = 288;                   // padded pixels per K-step: W + 6 <= 262, rounded up to 9 k-groups of 32
constexpr int WH_XT = WH_ROWS * 128;           // input-row tile
constexpr int WH_DY = WH_MAXW * 16;            // raw dY row: W pixels x 8 channels
constexpr int WH_STAGE = WH_XT + WH_DY;
constexpr int WH_NST = 3;
constexpr int WH_S = WH_ROWS * 128;            // shifted-dY operand: 288 rows x 64 columns (kw * 8 + channel; kw = 7 is zero)
}

__global__ __launch_bounds__(256, 1) void wgrad_head7_kernel(const bf16_t* __restrict__ P, const bf16_t* __restrict__ Q,
                                                               float* __restrict__ part, const WgHeadDesc d) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [stage 0..2: XT | dY][S]
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nets = d.group_rows > 0 ? 2 : 1;
    // XCD-aware order: the 7 kernel-row tiles of one split read the same input rows (shifted by kh); consecutive logical
    // blocks share an XCD, so its L2 serves 6 of the 7 reads (round-robin placement pulled every row from HBM 7 times:
    // 119 us instead of the ~35 us the traffic allows).
    int bid;
    {
        const int nwg = gridDim.x, o = blockIdx.x, xcd = o & 7, qq = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (qq + 1) : r * (qq + 1) + (xcd - r) * qq) + (o >> 3);
    }
    const int tile2 = bid % (7 * nets), split = bid / (7 * nets);
    const int net = tile2 / 7, kh = tile2 % 7;
    const int net_row0 = net ? d.group_rows : 0;
    const int net_rows = nets == 1 ? d.rows_total : (net ? d.rows_total - d.group_rows : d.group_rows);
    const int row_begin = net_row0 + (int)((long)split * net_rows / d.splits);
    const int row_end = net_row0 + (int)((long)(split + 1) * net_rows / d.splits);
    const int nk = row_end - row_begin;
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    const int W = d.W;

    // ---- DMA: the input-row tile is 36 pieces of 8 LDS rows (9 per wave), the dY row W/64 pieces
    const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P), 0, d.p_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsQ = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(Q), 0, d.q_bytes, 0x00020000);
    unsigned xoff[9];                                  // per piece: byte offset of this lane's source chunk inside the input row
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int r = 8 * (wave + 4 * i) + (lane >> 3);                // LDS row = padded pixel index; pixel column = r - 3
        const int c = r - 3;
        const bool inb = (unsigned)c < (unsigned)W;
        const bool ok = r < W + 6 && (refl || inb);
        const int cs = refl ? reflect_idx(c, W) : c;
        const int chunk = (lane & 7) ^ (((r >> 1) & 3) << 1);         // 16-byte chunk swizzle (conflict-free transposing reads)
        xoff[i] = ok ? (unsigned)((cs * WH_CI + chunk * 8) * 2) : 0xFFFFFFFFu;
    }
    const unsigned yoff = (wave * 64 + lane) < W ? (unsigned)((wave * 64 + lane) * d.Np * 2) : 0xFFFFFFFFu;   // Np == 8: 16 B per pixel
    int ib = row_begin / d.H, ii = row_begin % d.H, Rn = row_begin;
    auto issue = [&](int stage) {
        const int hi = ii + kh - 3;
        const bool valid = refl | ((unsigned)hi < (unsigned)d.H);
        const int hr = refl ? reflect_idx(hi, d.H) : (valid ? hi : 0);
        const int sQ = __builtin_amdgcn_readfirstlane((int)((unsigned)(ib * d.H + hr) * (unsigned)W * (unsigned)(WH_CI * 2)));
        const int sP = __builtin_amdgcn_readfirstlane((int)((unsigned)Rn * (unsigned)W * (unsigned)(d.Np * 2)));
        lds_ptr_t dst = (lds_ptr_t)smem + stage * WH_STAGE;
#pragma unroll
        for (int i = 0; i < 9; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsQ, (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16,
                                                     (int)(valid ? xoff[i] : 0xFFFFFFFFu), sQ, 0, 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsP, (__attribute__((address_space(3))) void*)(dst + WH_XT + wave * 1024), 16, (int)yoff, sP, 0, 0);
        ++Rn;
        if (++ii == d.H) { ii = 0; ++ib; }
    };

    // ---- fragment addressing: wave w owns input channels 16w..16w+15; four 16-column tiles of S (tile t = taps kw 2t, 2t+1)
    const int l16 = lane & 15, g = lane >> 4, qq = l16 >> 2, pp = l16 & 3;
    const int k0 = 4 * g + qq;                                        // (k0 >> 1) & 3 is the same for k0 + 16 h + 32 kk
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned swz = (unsigned)(((k0 >> 1) & 3) << 1);
    const unsigned aoff = (unsigned)(k0 * 128 + (((wave * 2 + (pp >> 1)) ^ swz) << 4) + (pp & 1) * 8);
    unsigned boff[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) boff[t] = lds0 + (unsigned)(WH_NST * WH_STAGE + k0 * 128 + (((t * 2 + (pp >> 1)) ^ swz) << 4) + (pp & 1) * 8);
    auto tr_read = [&](unsigned addr, auto off) -> bf16x4_t {
        u32x2_t r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value));
        return __builtin_bit_cast(bf16x4_t, r);
    };
    struct Frags { bf16x8_t a, b[4]; };
    auto read_frags = [&](Frags& f, unsigned sa, auto kkc) {          // 10 transposing reads: one 32-pixel k-group
        constexpr int o = decltype(kkc)::value * 32 * 128;
        const bf16x4_t alo = tr_read(sa, std::integral_constant<int, o>{}), ahi = tr_read(sa, std::integral_constant<int, o + 16 * 128>{});
        f.a = bf16x8_t{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bf16x4_t lo = tr_read(boff[t], std::integral_constant<int, o>{}), hi = tr_read(boff[t], std::integral_constant<int, o + 16 * 128>{});
            f.b[t] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        }
    };

    f32x4_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) issue(0);
    if (nk > 1) issue(1);
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // stage ks landed; the 10 DMAs of stage ks+1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // stage ks complete; everyone is done with S and stage ks-1
        if (ks + 2 < nk) issue((ks + 2) % WH_NST);
        const unsigned char* st = smem + (ks % WH_NST) * WH_STAGE;
        // ---- S[r][kw*8 + c] = dY[r - kw][c]  (r: padded pixel index; zero outside the row and for r >= W + 6): chunk kw of row r
        for (int r = tid; r < WH_ROWS; r += 256) {
            unsigned char* srow = smem + WH_NST * WH_STAGE + r * 128;
            const int sw = ((r >> 1) & 3) << 1;
#pragma unroll
            for (int kw = 0; kw < 8; ++kw) {
                const int j = r - kw;
                u32x4_t v = u32x4_t{0u, 0u, 0u, 0u};
                if (kw < 7 && (unsigned)j < (unsigned)W && r < W + 6) v = *reinterpret_cast<const u32x4_t*>(st + WH_XT + j * 16);
                *reinterpret_cast<u32x4_t*>(srow + ((kw ^ sw) << 4)) = v;
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // S rows written (not __syncthreads: its fence would also drain the DMAs in flight)
        __builtin_amdgcn_s_barrier();
        const unsigned sa = lds0 + (unsigned)((ks % WH_NST) * WH_STAGE) + aoff;
        Frags f0, f1;
        auto mma = [&](Frags& f) {
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f.a, f.b[t], acc[t], 0, 0, 0);
        };
        // two fragment sets: the reads of k-group kk+1 are in flight while the MFMAs of kk run (lgkmcnt is in-order: waiting
        // for "all but the 10 newest" is exactly "set kk has arrived")
        auto ready10 = [&](Frags& f) { asm volatile("s_waitcnt lgkmcnt(10)" : "+v"(f.a), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])); };
        auto ready0 = [&](Frags& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.a), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])); };
        read_frags(f0, sa, std::integral_constant<int, 0>{});
        read_frags(f1, sa, std::integral_constant<int, 1>{}); ready10(f0); mma(f0);
        read_frags(f0, sa, std::integral_constant<int, 2>{}); ready10(f1); mma(f1);
        read_frags(f1, sa, std::integral_constant<int, 3>{}); ready10(f0); mma(f0);
        read_frags(f0, sa, std::integral_constant<int, 4>{}); ready10(f1); mma(f1);
        read_frags(f1, sa, std::integral_constant<int, 5>{}); ready10(f0); mma(f0);
        read_frags(f0, sa, std::integral_constant<int, 6>{}); ready10(f1); mma(f1);
        read_frags(f1, sa, std::integral_constant<int, 7>{}); ready10(f0); mma(f0);
        read_frags(f0, sa, std::integral_constant<int, 8>{}); ready10(f1); mma(f1);
        ready0(f0); mma(f0);
    }

    // D[ci][col]: lane holds column l16 of tile t (col = 16 t + l16 = kw * 8 + co), rows ci = 16 wave + 4g .. +3
    float* out = part + ((long)net * d.splits + split) * d.Np * d.ncols;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = 16 * t + l16, kw = col >> 3, co = col & 7;
        if (kw < 7) *reinterpret_cast<f32x4_t*>(out + (long)co * d.ncols + (kh * 7 + kw) * WH_CI + wave * 16 + 4 * g) = acc[t];
    }
}


// ---------------------------------------------------------------------------------------------------------------
// Round 2: ALL SEVEN kernel rows in one block.  The form above gives every kernel row kh its own block, so an input row is
// pulled through LDS seven times (L2 serves six of them, but the L2 -> LDS fill is what the kernel is bound by: 940 MB per
// 16-image launch, 225 us).  Here a block walks a chunk of the PADDED rows Rp of one image, stages input row reflect(Rp) ONCE
// and uses it for the seven (output row, kh) pairs it belongs to, i = Rp + 3 - kh: the dY rows live in an LDS ring
// (nine slots: seven live rows + the two in flight; one new 4-KB row per step), and the shifted-dY operand S[r][kw*8 + co] = dY[i][r - kw][co] is not materialised any more -
// the transposing read takes a per-lane address, so lane (pixel r, column group of tap kw) reads dY pixel r - kw directly
// (16-byte pixels, zero margins of 8 pixels on both sides of a ring row; tap 7 and invalid output rows read a zero row).
// 28 accumulator tiles per wave (7 kh x 4 column tiles), 252 MFMAs and 522 transposing reads per wave and step against one
// 36-KB row of DMA: the fill traffic drops from 7x to (rows + 6) / rows of the input.
// Blocks: (network, image of the network, chunk of its H + 6 padded rows); partial slabs [network][image * chunks + chunk].
namespace {
constexpr int WA_RING = 9;                             // 7 live rows i in [Rp - 3, Rp + 3] + the rows of the next two steps in flight
// STEM = false: the head (64 -> 3): wide operand = the input x (64 channels, W + 6 haloed pixels per row, 9 k-groups), narrow
//               operand = dY (8 channels), pixel r of the wide row meets dY pixel r - kw of output row Rp + 3 - kh.
// STEM = true:  the stem (3 -> 64): the roles are swapped - wide operand = dY (64 channels, W pixels, 8 k-groups), narrow operand =
//               the input x (8 channels) with its reflected halo in the ring row (W + 6 padded pixels), pixel j of dY row i meets
//               padded x pixel j + kw of padded x row i + kh.  Output dW[co][kh][kw][ci] instead of dW[co][kh][kw][ci = wide].
template <bool STEM> struct WA {
    static constexpr int ROWS = STEM ? 256 : WH_ROWS;                  // wide-operand pixels (LDS rows of 128 B) per step
    static constexpr int NKG = ROWS / 32;
    static constexpr int XT = ROWS * 128;
    static constexpr int NPX = STEM ? 8 + 320 + 8 : 8 + WH_ROWS + 8;   // narrow ring row: 8 zero pixels | pixel slots | >= 8 zero pixels
    static constexpr int DYB = NPX * 16;
    static constexpr int DY0 = WH_NST * XT, ZERO = DY0 + WA_RING * DYB, SMEM = ZERO + DYB;
    static constexpr int NBIG = ROWS / 32;                             // wide-row DMA pieces per wave (4 waves x 8 rows per piece)
    static constexpr int NSMALL = STEM ? 2 : 1;                        // narrow-row pieces per wave (the fifth piece of the stem's 262 pixels is issued by every wave)
};
}

template <bool STEM>
__global__ __launch_bounds__(256, 1) void wgrad_head7all_kernel(const bf16_t* __restrict__ P1, const bf16_t* __restrict__ Q1,
                                                                  const bf16_t* __restrict__ P2, const bf16_t* __restrict__ Q2,
                                                                  float* __restrict__ part, const WgHeadDesc d) {
    using C = WA<STEM>;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];       // [wide stage 0..2][narrow ring 0..8][zero row]
    typedef __attribute__((address_space(3))) unsigned char* lds_ptr_t;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int bid = blockIdx.x;
    const int chunk = bid % d.chunks, t0 = bid / d.chunks, img_l = t0 % d.imgs_max, net = t0 / d.imgs_max;
    const int n_img = d.n1[net] + d.n2[net];
    const bool second = img_l >= d.n1[net];                         // block-uniform: this image lives in the second tensor pair
    const int img = second ? d.img2[net] + img_l - d.n1[net] : d.img1[net] + img_l;
    const bf16_t* P = second ? P2 : P1;
    const bf16_t* Q = second ? Q2 : Q1;
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    const int W = d.W, H = d.H;
    // rows of the WIDE operand this block walks: head: padded input rows (reflection [-3, H + 3), zero padding [0, H): rows outside
    // contribute nothing); stem: the H rows of dY
    const int lo = (!STEM && refl) ? -3 : 0, tp = (!STEM && refl) ? H + 6 : H;
    const int rp0 = lo + (int)((long)chunk * tp / d.chunks), rp1 = lo + (int)((long)(chunk + 1) * tp / d.chunks);
    const int nk = img_l < n_img ? rp1 - rp0 : 0;

    // zero margins of the ring rows and the zero row (the DMAs only ever write the pixel slots from 8 on)
    for (int i = tid; i < (WA_RING + 1) * C::NPX; i += 256) {
        const int row = i / C::NPX, c = i % C::NPX;
        if (row == WA_RING || c < 8 || c >= 8 + (STEM ? 320 : 256)) *reinterpret_cast<u32x4_t*>(smem + C::DY0 + row * C::DYB + c * 16) = u32x4_t{0u, 0u, 0u, 0u};
    }
    __syncthreads();

    // wide = 64-channel tensor (head: Q = x; stem: P = dY), narrow = 8-channel tensor (head: P = dY; stem: Q = x)
    const __amdgpu_buffer_rsrc_t rsWide = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(STEM ? P : Q), 0,
                                                                             STEM ? (second ? d.p2_bytes : d.p_bytes) : (second ? d.q2_bytes : d.q_bytes), 0x00020000);
    const __amdgpu_buffer_rsrc_t rsNarrow = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(STEM ? Q : P), 0,
                                                                               STEM ? (second ? d.q2_bytes : d.q_bytes) : (second ? d.p2_bytes : d.p_bytes), 0x00020000);
    unsigned xoff[C::NBIG];                            // per piece: byte offset of this lane's source chunk inside the wide row
#pragma unroll
    for (int i = 0; i < C::NBIG; ++i) {
        const int r = 8 * (wave + 4 * i) + (lane >> 3);                // LDS row: head = padded pixel index (column r - 3), stem = pixel column
        const int c = STEM ? r : r - 3;
        const bool inb = (unsigned)c < (unsigned)W;
        const bool ok = STEM ? inb : (r < W + 6 && (refl || inb));
        const int cs = (!STEM && refl) ? reflect_idx(c, W) : c;
        const int chk = (lane & 7) ^ (((r >> 1) & 3) << 1);           // 16-byte chunk swizzle (conflict-free transposing reads)
        xoff[i] = ok ? (unsigned)((cs * WH_CI + chk * 8) * 2) : 0xFFFFFFFFu;
    }
    unsigned yoff[C::NSMALL];                          // narrow row: 16 B per pixel; head: pixel = slot; stem: slot q = padded column, pixel reflect(q - 3)
#pragma unroll
    for (int i = 0; i < C::NSMALL; ++i) {
        const int q = 64 * (i == 0 ? wave : 4) + lane;
        if constexpr (STEM) {
            const int c = q - 3;
            const bool inb = (unsigned)c < (unsigned)W;
            const bool ok = q < W + 6 && (refl || inb);
            yoff[i] = ok ? (unsigned)((refl ? reflect_idx(c, W) : c) * 16) : 0xFFFFFFFFu;
        } else {
            yoff[i] = q < W ? (unsigned)(q * 16) : 0xFFFFFFFFu;
        }
    }
    auto issue_narrow = [&](int i) {                                   // narrow row with index i -> ring slot i mod 9 (zeros if it does not exist)
        // head: dY row i of the image; stem: PADDED input row i (input row reflect(i), or nothing outside a zero-padded image)
        const bool valid = (STEM && refl) ? true : (unsigned)i < (unsigned)H;
        const int row = (STEM && refl) ? reflect_idx(i, H) : (valid ? i : 0);
        const int sN = __builtin_amdgcn_readfirstlane((int)((unsigned)(img * H + row) * (unsigned)W * 16u));
        lds_ptr_t base = (lds_ptr_t)smem + C::DY0 + ((i + 9 * 64) % WA_RING) * C::DYB + 8 * 16;
#pragma unroll
        for (int k = 0; k < C::NSMALL; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsNarrow, (__attribute__((address_space(3))) void*)(base + 64 * (k == 0 ? wave : 4) * 16), 16,
                                                     (int)(valid ? yoff[k] : 0xFFFFFFFFu), sN, 0, 0);
    };
    auto issue = [&](int k) {                                          // step k: wide row rp0 + k -> stage k % 3; narrow row rp0 + k + 3
        const int rp = rp0 + k;
        const int hr = (!STEM && refl) ? reflect_idx(rp, H) : rp;
        const int sW = __builtin_amdgcn_readfirstlane((int)((unsigned)(img * H + hr) * (unsigned)W * (unsigned)(WH_CI * 2)));
        lds_ptr_t dst = (lds_ptr_t)smem + (k % WH_NST) * C::XT;
#pragma unroll
        for (int i = 0; i < C::NBIG; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsWide, (__attribute__((address_space(3))) void*)(dst + (wave + 4 * i) * 1024), 16, (int)xoff[i], sW, 0, 0);
        issue_narrow(rp + 3);
    };
    constexpr int PER_STEP = C::NBIG + C::NSMALL;                       // DMAs per wave and step (10 in both forms)
    static_assert(PER_STEP == 10, "the counted vmcnt below");

    // ---- fragment addressing: wave w owns wide channels 16w..16w+15; four 16-column tiles (tile t = taps kw 2t, 2t+1 x 8 narrow channels)
    const int l16 = lane & 15, g = lane >> 4, qq = l16 >> 2, pp = l16 & 3;
    const int k0 = 4 * g + qq;
    const unsigned lds0 = (unsigned)(size_t)(lds_ptr_t)smem;
    const unsigned swz = (unsigned)(((k0 >> 1) & 3) << 1);
    const unsigned aoff = (unsigned)(k0 * 128 + (((wave * 2 + (pp >> 1)) ^ swz) << 4) + (pp & 1) * 8);
    unsigned boff[4];                                                  // inside a ring row: head: dY pixel r - kw; stem: padded x pixel j + kw (r, j = k0 + 16 h + 32 kk)
    bool bz[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int kw = 2 * t + (pp >> 1);
        bz[t] = kw >= 7;
        boff[t] = (unsigned)((8 + k0 + (bz[t] ? 0 : (STEM ? kw : -kw))) * 16 + (pp & 1) * 8);
    }
    auto tr_read = [&](unsigned addr, auto off) -> bf16x4_t {
        u32x2_t r;
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(decltype(off)::value));
        return __builtin_bit_cast(bf16x4_t, r);
    };
    struct BF { bf16x8_t b[4]; };
    auto read_b = [&](BF& f, unsigned row_base, auto kkc) {            // 8 transposing reads: the four column tiles of one (kh, k-group)
        constexpr int o = decltype(kkc)::value * 32 * 16;
        const unsigned zb = lds0 + (unsigned)C::ZERO;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const unsigned a = (bz[t] ? zb : row_base) + boff[t];
            const bf16x4_t lo_ = tr_read(a, std::integral_constant<int, o>{}), hi_ = tr_read(a, std::integral_constant<int, o + 16 * 16>{});
            f.b[t] = bf16x8_t{lo_[0], lo_[1], lo_[2], lo_[3], hi_[0], hi_[1], hi_[2], hi_[3]};
        }
    };

    f32x4_t acc[7][4];
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[kh][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    if (nk > 0) {
#pragma unroll
        for (int i = -3; i < 3; ++i) issue_narrow(rp0 + i);            // the six rows the first step needs besides its own new one
        issue(0);
        if (nk > 1) issue(1);
    }
    for (int ks = 0; ks < nk; ++ks) {
        if (ks + 1 < nk) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");     // step ks landed; the 10 DMAs of step ks+1 may still fly
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // step ks complete for every wave; everyone is done with step ks-1's stage and ring slot
        if (ks + 2 < nk) issue(ks + 2);
        const int rp = rp0 + ks;
        unsigned rowb[7];                                   // ring row (or the zero row) of tap row kh: head: dY row rp + 3 - kh; stem: padded x row rp - 3 + kh
#pragma unroll
        for (int kh = 0; kh < 7; ++kh) {
            const int i = STEM ? rp - 3 + kh : rp + 3 - kh;
            const bool valid = (STEM && refl) ? true : (unsigned)i < (unsigned)H;
            rowb[kh] = lds0 + (unsigned)(valid ? C::DY0 + ((i + 9 * 64) % WA_RING) * C::DYB : C::ZERO);
        }
        const unsigned sa = lds0 + (unsigned)((ks % WH_NST) * C::XT) + aoff;
        auto kgroup = [&](auto kkc) {
            constexpr int o = decltype(kkc)::value * 32 * 128;
            const bf16x4_t alo = tr_read(sa, std::integral_constant<int, o>{}), ahi = tr_read(sa, std::integral_constant<int, o + 16 * 128>{});
            bf16x8_t a = bf16x8_t{alo[0], alo[1], alo[2], alo[3], ahi[0], ahi[1], ahi[2], ahi[3]};
            BF f0, f1;
            auto ready8 = [&](BF& f) { asm volatile("s_waitcnt lgkmcnt(8)" : "+v"(a), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])); };
            auto ready0 = [&](BF& f) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(f.b[0]), "+v"(f.b[1]), "+v"(f.b[2]), "+v"(f.b[3])); };
            auto mma = [&](BF& f, int kh) {
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[kh][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, f.b[t], acc[kh][t], 0, 0, 0);
            };
            read_b(f0, rowb[0], kkc);
            read_b(f1, rowb[1], kkc); ready8(f0); mma(f0, 0);
            read_b(f0, rowb[2], kkc); ready8(f1); mma(f1, 1);
            read_b(f1, rowb[3], kkc); ready8(f0); mma(f0, 2);
            read_b(f0, rowb[4], kkc); ready8(f1); mma(f1, 3);
            read_b(f1, rowb[5], kkc); ready8(f0); mma(f0, 4);
            read_b(f0, rowb[6], kkc); ready8(f1); mma(f1, 5);
            ready0(f0); mma(f0, 6);
        };
        kgroup(std::integral_constant<int, 0>{}); kgroup(std::integral_constant<int, 1>{}); kgroup(std::integral_constant<int, 2>{});
        kgroup(std::integral_constant<int, 3>{}); kgroup(std::integral_constant<int, 4>{}); kgroup(std::integral_constant<int, 5>{});
        kgroup(std::integral_constant<int, 6>{}); kgroup(std::integral_constant<int, 7>{});
        if constexpr (C::NKG > 8) kgroup(std::integral_constant<int, 8>{});
    }

    // accumulator tile (kh, t): lane holds column l16 (= kw * 8 + narrow channel) of wide channels 16 wave + 4g .. + 3
    float* out = part + ((long)net * d.splits + (long)img_l * d.chunks + chunk) * d.Np * d.ncols;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int col = 16 * t + l16, kw = col >> 3, cn = col & 7;
            if (kw >= 7) continue;
            if constexpr (STEM) {        // part[n = co (wide)][(kh*7 + kw) * 8 + ci (narrow)]
#pragma unroll
                for (int e = 0; e < 4; ++e) out[(long)(wave * 16 + 4 * g + e) * d.ncols + (kh * 7 + kw) * 8 + cn] = acc[kh][t][e];
            } else {                     // part[n = co (narrow)][(kh*7 + kw) * 64 + ci (wide)]
                *reinterpret_cast<f32x4_t*>(out + (long)cn * d.ncols + (kh * 7 + kw) * WH_CI + wave * 16 + 4 * g) = acc[kh][t];
            }
        }
}

static int g_wgrad_head = 1;    // A/B and parity hook: 1 = all seven kernel rows per block (round 2), 2 = one kernel row per block (round 1), 0 = generic kernel
extern "C" void uig_debug_set_wgrad_head(int on) { g_wgrad_head = on; }

// splits (= partial slabs per network) the head kernel in force wants: all-rows form: (images of the larger network) x chunks of
// padded rows, the grid sized for ~256 blocks; per-row form: 256 / 7 (or 14 when paired)
static int head_all_splits(int nets, int imgs_max, int H) {
    const int chunks = std::max(1, std::min(256 / (nets * imgs_max), (H + 6) / 4));
    return imgs_max * chunks;
}
int uig_wgrad_head_splits(int B, int group_images, int H) {
    const int nets = group_images > 0 ? 2 : 1;
    const int imgs_max = group_images > 0 ? std::max(group_images, B - group_images) : B;
    const int gmin = group_images > 0 ? std::min(group_images, B - group_images) : B;
    if (g_wgrad_head == 2) return (int)std::max<long>(1, std::min<long>(256 / (7 * nets), (long)gmin * H));
    return head_all_splits(nets, imgs_max, H);
}
// two tensor pairs (uig_wgrad_partial_pair2): 0 if the kernel in force cannot take them
int uig_wgrad_head_splits2(int B1, int g1, int B2, int g2, int swap2, int H) {
    if (g_wgrad_head != 1) return 0;
    const int a = g1 + (swap2 ? B2 - g2 : g2), b = (B1 - g1) + (swap2 ? g2 : B2 - g2);
    return head_all_splits(2, std::max(a, b), H);
}

// the head (dense operand dY of 8 padded channels, gathered operand x of 64) and, on the all-rows kernel, the stem (dY 64, x 8)
bool uig_wgrad_head_applicable(int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int dtype) {
    const bool shape = g_wgrad_head && dtype == UIG_BF16 && kH == 7 && kW == 7 && stride == 1 && pad == 3 && Mh == Hq && Mw == Wq &&
                       Mw >= 8 && Mw <= WH_MAXW && Hq >= 4;
    return shape && ((Np == 8 && Cq == WH_CI) || (g_wgrad_head == 1 && Np == WH_CI && Cq == 8));
}

template <bool STEM>
static int launch_head_all_t(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, const WgHeadDesc& d, int nets, hipStream_t s) {
    auto kern = wgrad_head7all_kernel<STEM>;
    static SmemAttrOnce attr_all;
    hipError_t e = attr_all.ensure(reinterpret_cast<const void*>(kern), (size_t)WA<STEM>::SMEM);
    if (e != hipSuccess) return uig_set_error((int)e, "wgrad(head): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    hipLaunchKernelGGL(kern, dim3(nets * d.splits), dim3(256), (size_t)WA<STEM>::SMEM, s, (const bf16_t*)P, (const bf16_t*)Q,
                       (const bf16_t*)(P2 ? P2 : P), (const bf16_t*)(Q2 ? Q2 : Q), ws, d);
    UIG_LAUNCH_CHECK("uig_wgrad_partial(7x7, all rows)");
    return 0;
}
static int launch_head_all(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, WgHeadDesc d, int splits, hipStream_t s, bool stem = false) {
    const int nets = d.n1[1] + d.n2[1] > 0 ? 2 : 1;
    const int imgs_max = std::max(d.n1[0] + d.n2[0], d.n1[1] + d.n2[1]);
    if (imgs_max <= 0 || splits % imgs_max != 0)
        return uig_set_error(-1, "wgrad(head): splits %d is not images %d x chunks (use uig_wgrad_splits / uig_wgrad_pair_splits / uig_wgrad_pair2_splits)", splits, imgs_max);
    d.imgs_max = imgs_max; d.chunks = splits / imgs_max; d.splits = splits;
    return stem ? launch_head_all_t<true>(P, Q, P2, Q2, ws, d, nets, s) : launch_head_all_t<false>(P, Q, P2, Q2, ws, d, nets, s);
}

// both generator passes in one launch: network 0 = images [0, g1) of (P, Q) + its share of (P2, Q2), network 1 the rest (see wgrad.hip)
int uig_launch_wgrad_head_runs(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, int B1, int g1, int B2, int g2, int swap2,
                               int H, int W, int Np, int pad_mode, int splits, hipStream_t s) {
    const bool stem = Np == WH_CI;                    // 64 dense channels: the stem (narrow operand = the 8-channel input)
    const int Cq = stem ? 8 : WH_CI;
    WgHeadDesc d{};
    d.B = B1; d.H = H; d.W = W; d.Np = Np; d.pad_mode = pad_mode; d.ncols = 49 * Cq; d.rows_total = B1 * H; d.group_rows = g1 * H;
    d.p_bytes = (unsigned)((long)B1 * H * W * Np * 2); d.q_bytes = (unsigned)((long)B1 * H * W * Cq * 2);
    d.p2_bytes = (unsigned)((long)B2 * H * W * Np * 2); d.q2_bytes = (unsigned)((long)B2 * H * W * Cq * 2);
    d.n1[0] = g1; d.img1[0] = 0; d.n1[1] = B1 - g1; d.img1[1] = g1;
    d.n2[0] = swap2 ? B2 - g2 : g2; d.img2[0] = swap2 ? g2 : 0;
    d.n2[1] = swap2 ? g2 : B2 - g2; d.img2[1] = swap2 ? 0 : g2;
    return launch_head_all(P, Q, P2, Q2, ws, d, splits, s, stem);
}

int uig_launch_wgrad_head(const void* P, const void* Q, float* ws, int B, int H, int W, int Np, int pad_mode, int splits,
                          int group_images, hipStream_t s) {
    const bool stem = Np == WH_CI;                    // 64 dense channels: the stem (all-rows kernel only)
    const int Cq = stem ? 8 : WH_CI;
    WgHeadDesc d{};
    d.B = B; d.H = H; d.W = W; d.Np = Np; d.pad_mode = pad_mode; d.ncols = 49 * Cq; d.rows_total = B * H;
    d.splits = splits; d.group_rows = group_images * H;
    d.p_bytes = (unsigned)((long)B * H * W * Np * 2); d.q_bytes = (unsigned)((long)B * H * W * Cq * 2);
    if (g_wgrad_head != 2) {      // all seven kernel rows per block
        if (group_images > 0) { d.n1[0] = group_images; d.img1[0] = 0; d.n1[1] = B - group_images; d.img1[1] = group_images; }
        else { d.n1[0] = B; d.img1[0] = 0; }
        return launch_head_all(P, Q, nullptr, nullptr, ws, d, splits, s, stem);
    }
    if (stem) return uig_set_error(-1, "wgrad(7x7): the stem shape needs the all-rows kernel");
    const size_t smem = (size_t)WH_NST * WH_STAGE + WH_S;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(wgrad_head7_kernel), (size_t)(int)smem);
        if (e != hipSuccess) return uig_set_error((int)e, "wgrad(head): hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(wgrad_head7_kernel, dim3(7 * splits * (group_images > 0 ? 2 : 1)), dim3(256), smem, s, (const bf16_t*)P,
                       (const bf16_t*)Q, ws, d);
    UIG_LAUNCH_CHECK("uig_wgrad_partial(head)");
    return 0;
}
