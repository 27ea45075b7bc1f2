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

static int g_wgrad_head = 1;    // A/B and parity hook
extern "C" void uig_debug_set_wgrad_head(int on) { g_wgrad_head = on; }

bool uig_wgrad_head_applicable(int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int dtype) {
    return g_wgrad_head && dtype == UIG_BF16 && kH == 7 && kW == 7 && stride == 1 && pad == 3 && Mh == Hq && Mw == Wq &&
           Mw >= 8 && Mw <= WH_MAXW && Hq >= 4 && Np == 8 && Cq == WH_CI;
}

int uig_launch_wgrad_head(const void* P, const void* Q, float* ws, int B, int H, int W, int Np, int pad_mode, int splits,
                          int group_images, hipStream_t s) {
    WgHeadDesc d{};
    d.B = B; d.H = H; d.W = W; d.Np = Np; d.pad_mode = pad_mode; d.ncols = 49 * WH_CI; d.rows_total = B * H;
    d.splits = splits; d.group_rows = group_images * H;
    d.p_bytes = (unsigned)((long)B * H * W * Np * 2); d.q_bytes = (unsigned)((long)B * H * W * WH_CI * 2);
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
