// wgrad.hip — weight-gradient GEMM (aten::convolution_backward, weight grad) for gfx950.
//
//   part[s][n][col] = sum_{pixels m in split s} P[m][n] * Q[pix(m, tap(col))][c(col)],   col = tap*Cq + c
//
// The reduction runs over PIXELS, which is the slow (strided) index of both NHWC operands, so both MFMA operands are
// "transposed" reads: tiles are staged as [pixel][channel] rows in LDS (coalesced 16-byte chunks from HBM) and
//   bf16: fragments come from ds_read_b64_tr_b16 (hardware transpose read, 4 pixel rows x 16 channels per 16 lanes)
//   f32 : v_mfma_f32_16x16x4_f32 takes one value per lane, read with conflict-free ds_read_b32.
// Output tiles are few (e.g. 2 x 18 for a 256->256 3x3), so the pixel range is split over gridDim.y and the fp32
// partial slabs are summed by wgrad_reduce_kernel (deterministic; no float atomics), which also transposes
// [n][tap][c] -> torch's [d0][d1][kH][kW] and drops the padded channels.
#include "uig_common.h"
#include <algorithm>

struct WgradDesc {
    int B, Mh, Mw, Np;       // dense operand P: (B, Mh, Mw, Np)
    int Hq, Wq, Cq;          // gathered operand Q: (B, Hq, Wq, Cq)
    int kW, taps, stride, pad, pad_mode;
    int ncols;               // taps * Cq
    int M, Mper;             // pixels total, pixels per split (multiple of 32)
    int group_M, splits;     // two networks in one launch (group_M > 0): pixels [0, group_M) are network 0's; gridDim.y = 2 * splits and
                             // the partial slabs are laid out [network][split]
    unsigned p_bytes, q_bytes;
    // two RUNS of pixels per network (round 2: both generator passes of a step in one launch): splits [0, splits0) of a network
    // walk its pixels of (P, Q) as above, splits [splits0, splits) its pixels of a second tensor pair (P2, Q2) of M2 pixels, of
    // which [0, group_M2) are network (swap2 ? 1 : 0)'s.  A split never leaves its run, so the kernel body sees one tensor pair.
    // splits0 == 0: one run (P2 / Q2 unused).
    int splits0, M2, group_M2, swap2, Mper2;
    unsigned p2_bytes, q2_bytes;
};

template <typename T> struct WgTraits;
template <> struct WgTraits<bf16_t> { static constexpr int PAD = 32; };
template <> struct WgTraits<float> { static constexpr int PAD = 64; };

// FASTROW: Mw % BKP == 0, so the BKP pixels of a K-step lie in ONE image row: the (image, row) part of the gather address
// is computed once per K-step from block-uniform running coordinates and only the column part per staged row.
// BN = 256 runs 8 waves (512 threads): the gathered Q tile is staged once for all 256 rows of the dense operand instead of
// once per 128-row tile (48 KB instead of 64 KB of staging per 2.1 MMAC).
template <typename T, int BN, bool FASTROW>
__global__ __launch_bounds__((BN >= 256 ? 512 : 256), 2) void wgrad_kernel(const T* __restrict__ P1, const T* __restrict__ Q1,
                                                        const T* __restrict__ P2, const T* __restrict__ Q2,
                                                        float* __restrict__ part, const WgradDesc d) {
    constexpr int E = ElemTraits<T>::E;
    constexpr int BC = 128;
    constexpr int BKP = (sizeof(T) == 2) ? 64 : 32;   // pixels per K-step: bf16 64 (two MFMA k-groups per barrier), f32 32
    constexpr int NTHR = (BN >= 256) ? 512 : 256, NWAVE = NTHR / 64;
    constexpr int WAVES_N = (BN >= 256) ? 4 : (BN >= 128) ? 2 : 1, WAVES_C = NWAVE / WAVES_N;
    constexpr int WN = BN / WAVES_N, WC = BC / WAVES_C, NT = WN / 16, CT = WC / 16;
    constexpr int PROW = BN * (int)sizeof(T) + WgTraits<T>::PAD;      // LDS row strides (bytes)
    constexpr int QROW = BC * (int)sizeof(T) + WgTraits<T>::PAD;
    constexpr int STAGE = BKP * (PROW + QROW);
    constexpr int PCH = BN / E, QCH = BC / E;                          // 16-byte chunks per row
    constexpr int PI = (BKP * PCH + NTHR - 1) / NTHR, QI = (BKP * QCH + NTHR - 1) / NTHR;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ntn = (d.Np + BN - 1) / BN;
    const int n_base = (blockIdx.x % ntn) * BN, col_base = (blockIdx.x / ntn) * BC;
    const int net = d.group_M > 0 ? (int)blockIdx.y / d.splits : 0;
    const int split = d.group_M > 0 ? (int)blockIdx.y % d.splits : (int)blockIdx.y;
    const bool run1 = d.splits0 > 0 && split >= d.splits0;            // block-uniform: this split walks the second tensor pair
    const int net2 = d.swap2 ? 1 - net : net;
    const int net_m0 = run1 ? (net2 ? d.group_M2 : 0) : (net ? d.group_M : 0);
    const int net_m1 = run1 ? (net2 ? d.M2 : d.group_M2) : ((d.group_M > 0 && !net) ? d.group_M : d.M);
    const int mper = run1 ? d.Mper2 : d.Mper;
    const int m_begin = net_m0 + (run1 ? split - d.splits0 : split) * mper;
    const int m_end = min(net_m1, m_begin + mper);
    const T* P = run1 ? P2 : P1;
    const T* Q = run1 ? Q2 : Q1;
    const int nk = (m_end > m_begin) ? (m_end - m_begin + BKP - 1) / BKP : 0;

    // fixed per-thread chunk columns
    int q_c[QI], q_dh[QI], q_dw[QI], q_row[QI]; bool q_ok[QI];
#pragma unroll
    for (int i = 0; i < QI; ++i) {
        const int id = tid + NTHR * i;
        q_row[i] = id / QCH;
        const int col = col_base + (id % QCH) * E;
        q_ok[i] = (id < BKP * QCH) && (col < d.ncols);
        const int tap = q_ok[i] ? col / d.Cq : 0;
        q_c[i] = q_ok[i] ? col % d.Cq : 0;
        q_dh[i] = tap / d.kW - d.pad; q_dw[i] = tap % d.kW - d.pad;
    }
    int p_n[PI], p_row[PI]; bool p_ok[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) {
        const int id = tid + NTHR * i;
        p_row[i] = id / PCH;
        p_n[i] = n_base + (id % PCH) * E;
        p_ok[i] = (id < BKP * PCH) && (p_n[i] < d.Np);
    }

    u32x4_t rq[QI], rp[PI];
    // buffer descriptors: invalid rows / zero padding use an out-of-range offset (hardware returns zeros)
    const __amdgpu_buffer_rsrc_t rsq = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(Q), 0, run1 ? d.q2_bytes : d.q_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsp = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(P), 0, run1 ? d.p2_bytes : d.p_bytes, 0x00020000);
    // running pixel coordinates of the K-step being loaded: one division at the start, then incremental updates (per-K-step
    // divisions made this loop VALU-bound: ~250 VALU per 16 MFMAs).  FASTROW keeps ONE block-uniform (b0, i0, j0).
    int qj[QI], qi[QI], qb[QI];
    int j0 = 0, i0 = 0, b0 = 0;
    if constexpr (FASTROW) {
        j0 = m_begin % d.Mw; const int t = m_begin / d.Mw; i0 = t % d.Mh; b0 = t / d.Mh;
    } else {
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int m = m_begin + q_row[i];
            qj[i] = m % d.Mw; const int t = m / d.Mw; qi[i] = t % d.Mh; qb[i] = t / d.Mh;
        }
    }
    const bool refl = d.pad_mode == UIG_PAD_REFLECT;
    unsigned poff[PI];
#pragma unroll
    for (int i = 0; i < PI; ++i) poff[i] = (unsigned)(((m_begin + p_row[i]) * d.Np + p_n[i]) * (int)sizeof(T));
    auto load_tile = [&](int ks) {
        const int mk = m_begin + ks * BKP;
        if constexpr (FASTROW) {
            // all staged rows share the image row: (image, row) term once, column term per row.  The tap (dh, dw) and the
            // channel chunk are the same for every row slot of a thread (256 % QCH == 0).
            const int hi = i0 * d.stride + q_dh[0];
            const bool inb_h = (unsigned)hi < (unsigned)d.Hq;
            const int hr = refl ? reflect_idx(hi, d.Hq) : hi;
            const int rowoff = (b0 * d.Hq + hr) * d.Wq;
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int wi = (j0 + q_row[i]) * d.stride + q_dw[0];
                const bool inb = inb_h & ((unsigned)wi < (unsigned)d.Wq);
                const bool ok = q_ok[i] & (mk + q_row[i] < m_end) & (refl | inb);
                const int wr = refl ? reflect_idx(wi, d.Wq) : wi;
                const unsigned off = ok ? (unsigned)(((rowoff + wr) * d.Cq + q_c[0]) * (int)sizeof(T)) : 0xFFFFFFFFu;
                rq[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsq, (int)off, 0, 0));
            }
            j0 += BKP;
            if (j0 >= d.Mw) { j0 = 0; if (++i0 == d.Mh) { i0 = 0; ++b0; } }
        } else {
#pragma unroll
            for (int i = 0; i < QI; ++i) {
                const int m = mk + q_row[i];
                const int hi = qi[i] * d.stride + q_dh[i], wi = qj[i] * d.stride + q_dw[i];
                const bool inb = ((unsigned)hi < (unsigned)d.Hq) & ((unsigned)wi < (unsigned)d.Wq);
                const bool ok = q_ok[i] & (m < m_end) & (refl | inb);
                const int hr = refl ? reflect_idx(hi, d.Hq) : hi, wr = refl ? reflect_idx(wi, d.Wq) : wi;
                const unsigned off = ok ? (unsigned)((((qb[i] * d.Hq + hr) * d.Wq + wr) * d.Cq + q_c[i]) * (int)sizeof(T)) : 0xFFFFFFFFu;
                rq[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsq, (int)off, 0, 0));
                // advance this row by BKP pixels for the next call: carries instead of divisions (three integer divisions per
                // staged row and K-step made the 31x31 / 32x32 PatchGAN layers VALU-bound: 222 us for the 256->512 layer)
                int j = qj[i] + BKP, r = qi[i], bb = qb[i];
                while (j >= d.Mw) { j -= d.Mw; if (++r == d.Mh) { r = 0; ++bb; } }
                qj[i] = j; qi[i] = r; qb[i] = bb;
            }
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const bool ok = p_ok[i] & (mk + p_row[i] < m_end);
            rp[i] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rsp, ok ? (int)poff[i] : -1, 0, 0));
            poff[i] += (unsigned)(BKP * d.Np * (int)sizeof(T));
        }
    };
    auto store_tile = [&](int stage) {
        unsigned char* sp = smem + stage * STAGE;
        unsigned char* sq = sp + BKP * PROW;
#pragma unroll
        for (int i = 0; i < QI; ++i) {
            const int id = tid + NTHR * i;
            if (id < BKP * QCH) *reinterpret_cast<u32x4_t*>(sq + q_row[i] * QROW + (id % QCH) * 16) = rq[i];
        }
#pragma unroll
        for (int i = 0; i < PI; ++i) {
            const int id = tid + NTHR * i;
            if (id < BKP * PCH) *reinterpret_cast<u32x4_t*>(sp + p_row[i] * PROW + (id % PCH) * 16) = rp[i];
        }
    };

    f32x4_t acc[CT][NT];
#pragma unroll
    for (int a = 0; a < CT; ++a)
#pragma unroll
        for (int b = 0; b < NT; ++b) acc[a][b] = f32x4_t{0.f, 0.f, 0.f, 0.f};

    const int wn = wave % WAVES_N, wc = wave / WAVES_N;
    const int l16 = lane & 15, g = lane >> 4;

    if (nk > 0) {
        load_tile(0);
        store_tile(0);
    }
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const int cur = ks & 1;
        if (ks + 1 < nk) load_tile(ks + 1);
        const unsigned char* sp = smem + cur * STAGE;
        const unsigned char* sq = sp + BKP * PROW;
        if constexpr (sizeof(T) == 2) {
            // bf16: transpose reads. lane 4q+p of a 16-lane group addresses pixel row (4g+q [+16]), channels 4p..4p+3;
            // it receives channel l16 of those 4 pixel rows. fragment k order: {4g..4g+3, 16+4g..16+4g+3} for both operands.
            const int qq = l16 >> 2, pp = l16 & 3;
#pragma unroll
            for (int kg = 0; kg < BKP / 32; ++kg) {
                bf16x8_t af[CT], bf[NT];
#pragma unroll
                for (int a = 0; a < CT; ++a) {
                    const unsigned char* base = sq + (32 * kg + 4 * g + qq) * QROW + (wc * WC + a * 16 + 4 * pp) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(base));
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(base + 16 * QROW));
                    af[a] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int b = 0; b < NT; ++b) {
                    const unsigned char* base = sp + (32 * kg + 4 * g + qq) * PROW + (wn * WN + b * 16 + 4 * pp) * 2;
                    bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(base));
                    bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(base + 16 * PROW));
                    bf[b] = bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int a = 0; a < CT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        } else {
            // f32: A[i = l16][k = g] per 16x16x4 step; pixel row = 4*kk + g
#pragma unroll
            for (int kk = 0; kk < BKP / 4; ++kk) {
                float af[CT], bf[NT];
#pragma unroll
                for (int a = 0; a < CT; ++a)
                    af[a] = *reinterpret_cast<const float*>(sq + (4 * kk + g) * QROW + (wc * WC + a * 16 + l16) * 4);
#pragma unroll
                for (int b = 0; b < NT; ++b)
                    bf[b] = *reinterpret_cast<const float*>(sp + (4 * kk + g) * PROW + (wn * WN + b * 16 + l16) * 4);
#pragma unroll
                for (int a = 0; a < CT; ++a)
#pragma unroll
                    for (int b = 0; b < NT; ++b)
                        acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[a], bf[b], acc[a][b], 0, 0, 0);
            }
        }
        if (ks + 1 < nk) store_tile(cur ^ 1);
        __syncthreads();
    }

    // D[col][n]: lane holds n = l16 (B-operand column), cols 4g..4g+3 (rows) -> one float4 per tile into part[s][n][col]
    float* out = part + (long)blockIdx.y * d.Np * d.ncols;
#pragma unroll
    for (int b = 0; b < NT; ++b) {
        const int n = n_base + wn * WN + b * 16 + l16;
        if (n >= d.Np) continue;
#pragma unroll
        for (int a = 0; a < CT; ++a) {
            const int col = col_base + wc * WC + a * 16 + 4 * g;
            if (col >= d.ncols) continue;     // ncols is a multiple of 8, col of 4: a chunk is fully in or out
            *reinterpret_cast<f32x4_t*>(out + (long)n * d.ncols + col) = acc[a][b];
        }
    }
}

// Optional rider on the reduce launch: the bias gradient from the InstanceNorm backward's column-sum partials
// (partial[slab][C][2], see instnorm.hip) - saves one tiny launch per layer on the backward's side stream.
struct BiasRider { const float* cpart; float* db; int nslab, C, nreal, accumulate, main_blocks; const float* cpart2; int nslab2; };      // cpart2: optional second run of slabs (the other generator pass's)
__device__ __forceinline__ void bias_rider_block(const BiasRider& br, int blk) {
    const int sl = threadIdx.x & 15, c = blk * 16 + (threadIdx.x >> 4);
    double a = 0.0;
    if (c < br.C)
        for (int s = sl; s < br.nslab; s += 16) a += (double)br.cpart[((long)s * br.C + c) * 2];
    if (c < br.C && br.cpart2 != nullptr)
        for (int s = sl; s < br.nslab2; s += 16) a += (double)br.cpart2[((long)s * br.C + c) * 2];
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) a += __shfl_xor(a, o, 16);
    if (sl == 0 && c < br.nreal) br.db[c] = br.accumulate ? br.db[c] + (float)a : (float)a;
}

// dW[d0][d1][tap] (+)= sum_s part[s][d0][tap][d1]
// One thread per (d0, d1): for every tap the reads of consecutive threads are consecutive d1 (coalesced slabs), and each
// thread writes its `taps` consecutive output floats, so a wave writes one contiguous 64*taps*4-byte run (the earlier
// one-thread-per-element form wrote with a stride of `taps` floats and ran at 1.6 TB/s).
template <int TAPS>
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dW, int Np, int Cq,
                                                            int taps_rt, int splits, int D0, int D1, int accumulate, BiasRider br,
                                                            float* __restrict__ dW2, const BiasRider br2) {
    const int taps = TAPS > 0 ? TAPS : taps_rt;
    const long slab = (long)Np * taps * Cq;
    if (blockIdx.y) { part += (long)splits * slab; dW = dW2; br = br2; }      // second network of a paired launch
    if ((int)blockIdx.x >= br.main_blocks) { if (br.cpart) bias_rider_block(br, blockIdx.x - br.main_blocks); return; }
    const int total = D0 * D1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += br.main_blocks * blockDim.x) {
        const int d1 = i % D1, d0 = i / D1;
        const float* src = part + (long)d0 * taps * Cq + d1;
        float* dst = dW + (long)i * taps;
        if constexpr (TAPS > 0) {
            float acc[TAPS > 0 ? TAPS : 1];
#pragma unroll
            for (int t = 0; t < TAPS; ++t) acc[t] = 0.f;
            for (int k = 0; k < splits; ++k) {
#pragma unroll
                for (int t = 0; t < TAPS; ++t) acc[t] += src[k * slab + (long)t * Cq];
            }
#pragma unroll
            for (int t = 0; t < TAPS; ++t) dst[t] = accumulate ? dst[t] + acc[t] : acc[t];
        } else {
            for (int t = 0; t < taps; ++t) {
                float a = 0.f;
                for (int k = 0; k < splits; ++k) a += src[k * slab + (long)t * Cq];
                dst[t] = accumulate ? dst[t] + a : a;
            }
        }
    }
}

// few (d0, d1) pairs (3-channel stem / head, 1-channel discriminator head): 16 output ELEMENTS per block, 16 split lanes per
// element (these layers run with up to 512 splits; one thread per element walked them serially: 100+ us)
__global__ __launch_bounds__(256) void wgrad_reduce_elem_kernel(const float* __restrict__ part, float* __restrict__ dW, int Np, int Cq, int taps,
                                                                 int splits, int D0, int D1, int accumulate, BiasRider br,
                                                                 float* __restrict__ dW2, const BiasRider br2) {
    const long total = (long)D0 * D1 * taps;
    const long slab = (long)Np * taps * Cq;
    if (blockIdx.y) { part += (long)splits * slab; dW = dW2; br = br2; }      // second network of a paired launch
    if ((int)blockIdx.x >= br.main_blocks) { if (br.cpart) bias_rider_block(br, blockIdx.x - br.main_blocks); return; }
    const int sl = threadIdx.x & 15;
    for (long i = (long)blockIdx.x * 16 + (threadIdx.x >> 4); i < total + 15; i += (long)br.main_blocks * 16) {   // uniform trip count per 16-lane group
        const bool ok = i < total;
        const long ii = ok ? i : 0;
        const int d1 = (int)(ii % D1); const long r = ii / D1; const int tap = (int)(r % taps); const int d0 = (int)(r / taps);
        const long src = ((long)d0 * taps + tap) * Cq + d1;
        float sacc = 0.f;
        if (ok) for (int k = sl; k < splits; k += 16) sacc += part[k * slab + src];
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) sacc += __shfl_xor(sacc, o, 16);
        if (ok && sl == 0) {
            const long dst = ((long)d0 * D1 + d1) * taps + tap;
            dW[dst] = accumulate ? dW[dst] + sacc : sacc;
        }
    }
}

extern "C" size_t uig_wgrad_workspace_bytes(int Np, int Cq, int kH, int kW, int splits) {
    return (size_t)splits * Np * kH * kW * Cq * sizeof(float);
}

template <typename T, int BN, bool FASTROW>
static int launch_wgrad(const void* P, const void* Q, float* ws, const WgradDesc& d, int splits, hipStream_t s, const void* P2 = nullptr, const void* Q2 = nullptr) {
    constexpr int PROW = BN * (int)sizeof(T) + WgTraits<T>::PAD, QROW = 128 * (int)sizeof(T) + WgTraits<T>::PAD;
    const size_t smem = 2 * (size_t)(sizeof(T) == 2 ? 64 : 32) * (size_t)(PROW + QROW);
    auto kern = wgrad_kernel<T, BN, FASTROW>;
    static SmemAttrOnce attr_once;
    {
        hipError_t e = attr_once.ensure(reinterpret_cast<const void*>(kern), (size_t)(int)smem);
        if (e != hipSuccess) return uig_set_error((int)e, "wgrad: hipFuncSetAttribute failed: %s", hipGetErrorString(e));
    }
    const int ntn = (d.Np + BN - 1) / BN, ntc = (d.ncols + 127) / 128;
    hipLaunchKernelGGL(kern, dim3(ntn * ntc, splits * (d.group_M > 0 ? 2 : 1)), dim3(BN >= 256 ? 512 : 256), smem, s, (const T*)P, (const T*)Q, (const T*)P2, (const T*)Q2, ws, d);
    UIG_LAUNCH_CHECK("uig_wgrad_partial");
    return 0;
}

static int g_wgrad_bn256 = 0;   // measured slower (73.9 vs 68 us at batch 8: one 8-wave block per CU runs in lockstep, two 4-wave blocks cover each other)
extern "C" void uig_debug_set_wgrad_wide(int on) { g_wgrad_bn256 = on; }
// rows of the dense operand one block covers for this launch (16, 128 or 256): the caller sizes `splits` from it
extern "C" int uig_wgrad_tile_rows(int Np, int Mw, int dtype) {
    const int bkp = dtype == UIG_BF16 ? 64 : 32;
    if (g_wgrad_bn256 && Np % 256 == 0 && (Mw % bkp) == 0) return 256;
    return Np <= 16 ? 16 : (Np <= 64 ? 64 : 128);      // 64: the 7x7 stem and the PatchGAN's first layer (64 output channels): half the dense tile of the 128-row form was padding
}

// wgrad_rows.hip: stride-1 3x3 "same" convs on 64-pixel rows, bf16
// (round 3: also the stride-2 3x3 layers with 64-pixel output rows; pad_mode defaults to zero padding for the split-count queries, which do not carry it)
bool uig_wgrad_rows_applicable(int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int dtype, int pad_mode = UIG_PAD_ZERO);
int uig_wgrad_rows_tiles(int Np, int Cq);
int uig_launch_wgrad_rows(const void* P, const void* Q, float* ws, int B, int H, int W, int Np, int Cq, int pad_mode, int splits, int group_images, hipStream_t s, int stride = 1);
// wgrad_head.hip: 7x7 stride-1 pad-3 convs with 64 input and <= 8 (padded) output channels, bf16
bool uig_wgrad_head_applicable(int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int dtype);
int uig_launch_wgrad_head(const void* P, const void* Q, float* ws, int B, int H, int W, int Np, int pad_mode, int splits, int group_images, hipStream_t s);
int uig_wgrad_head_splits(int B, int group_images, int H);
int uig_wgrad_head_splits2(int B1, int g1, int B2, int g2, int swap2, int H);
int uig_launch_wgrad_head_runs(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, int B1, int g1, int B2, int g2, int swap2,
                               int H, int W, int Np, int pad_mode, int splits, hipStream_t s);

// number of pixel-range splits (= fp32 partial slabs) uig_wgrad_partial should run with for this shape: the kernel it will
// dispatch to decides (row kernel: one 8-wave block per CU; generic kernel: `target_blocks` 4-wave blocks, two per CU)
extern "C" int uig_wgrad_splits(int B, int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad,
                                int dtype, int target_blocks) {
    const long M = (long)B * Mh * Mw;
    if (uig_wgrad_rows_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype))
        return (int)std::max<long>(1, std::min<long>(256 / uig_wgrad_rows_tiles(Np, Cq), (long)B * Mh));
    if (uig_wgrad_head_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype))
        return uig_wgrad_head_splits(B, 0, Mh);
    const int bn = uig_wgrad_tile_rows(Np, Mw, dtype);
    const int tiles = ((Np + bn - 1) / bn) * ((kH * kW * Cq + 127) / 128);
    if (bn >= 256) return (int)std::max<long>(1, std::min<long>(256 / tiles, M / 128));
    return (int)std::max<long>(1, std::min<long>(std::max(target_blocks, 1) / std::max(tiles, 1), M / 128));
}

struct WgRun2 { const void* P2; const void* Q2; int B2, g2, swap2, splits0; };      // second run of the generic kernel (uig_wgrad_partial_pair2)
static int wgrad_partial_impl(const void* P, const void* Q, float* workspace, int B, int group_images, int Mh, int Mw, int Np,
                              int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int pad_mode,
                              int splits, int dtype, void* stream, const WgRun2* r2 = nullptr) {
    UIG_CHECK_ARG(P && Q && workspace, "uig_wgrad_partial: null pointer");
    UIG_CHECK_ARG(group_images >= 0 && group_images < B, "uig_wgrad_partial: bad group_images %d of %d", group_images, B);
    UIG_CHECK_ARG(Np % 8 == 0 && Cq % 8 == 0 && Np > 0 && Cq > 0, "uig_wgrad_partial: channels must be padded to 8 (Np=%d Cq=%d)", Np, Cq);
    UIG_CHECK_ARG(splits >= 1 && splits <= 65535, "uig_wgrad_partial: bad splits %d", splits);
    UIG_CHECK_ARG(stride == 1 || stride == 2, "uig_wgrad_partial: stride=%d unsupported", stride);
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "uig_wgrad_partial: bad dtype %d", dtype);
    UIG_CHECK_ARG((long)B * Mh * Mw * Np * (dtype == UIG_BF16 ? 2 : 4) < (1L << 32) - 64 && (long)B * Hq * Wq * Cq * (dtype == UIG_BF16 ? 2 : 4) < (1L << 32) - 64, "uig_wgrad_partial: tensor too large for 32-bit byte offsets");
    if (pad_mode == UIG_PAD_REFLECT) {
        UIG_CHECK_ARG(pad < Hq && pad < Wq, "uig_wgrad_partial: reflect pad %d >= dim", pad);
    }
    // every gathered coordinate must be reachable: (Mh-1)*stride + kH-1 - pad <= Hq-1 + pad
    UIG_CHECK_ARG((Mh - 1) * stride + kH - 1 - pad <= Hq - 1 + pad && (Mw - 1) * stride + kW - 1 - pad <= Wq - 1 + pad,
                  "uig_wgrad_partial: gather window exceeds the padded input (Mh=%d Hq=%d k=%d s=%d p=%d)", Mh, Hq, kH, stride, pad);
    WgradDesc d{};
    d.B = B; d.Mh = Mh; d.Mw = Mw; d.Np = Np; d.Hq = Hq; d.Wq = Wq; d.Cq = Cq;
    d.kW = kW; d.taps = kH * kW; d.stride = stride; d.pad = pad; d.pad_mode = pad_mode;
    d.ncols = kH * kW * Cq; d.M = B * Mh * Mw;
    d.group_M = group_images * Mh * Mw; d.splits = splits;
    const int M_net = group_images > 0 ? std::max(d.group_M, d.M - d.group_M) : d.M;
    const long esz = dtype == UIG_BF16 ? 2 : 4;
    d.p_bytes = (unsigned)((long)B * Mh * Mw * Np * esz); d.q_bytes = (unsigned)((long)B * Hq * Wq * Cq * esz);
    hipStream_t s = (hipStream_t)stream;
    const void* P2 = nullptr; const void* Q2 = nullptr;
    if (r2 != nullptr) {      // generic kernel, two runs: the first splits0 splits of a network on (P, Q), the others on (P2, Q2)
        UIG_CHECK_ARG(group_images > 0 && r2->P2 && r2->Q2 && r2->g2 > 0 && r2->g2 < r2->B2 && r2->splits0 >= 1 && r2->splits0 < splits, "uig_wgrad_partial_pair2: bad second run");
        UIG_CHECK_ARG((long)r2->B2 * Mh * Mw * Np * esz < (1L << 32) - 64 && (long)r2->B2 * Hq * Wq * Cq * esz < (1L << 32) - 64, "uig_wgrad_partial_pair2: tensor too large for 32-bit byte offsets");
        P2 = r2->P2; Q2 = r2->Q2;
        d.splits0 = r2->splits0; d.M2 = r2->B2 * Mh * Mw; d.group_M2 = r2->g2 * Mh * Mw; d.swap2 = r2->swap2;
        d.p2_bytes = (unsigned)((long)r2->B2 * Mh * Mw * Np * esz); d.q2_bytes = (unsigned)((long)r2->B2 * Hq * Wq * Cq * esz);
        d.Mper = ((M_net + d.splits0 - 1) / d.splits0 + 63) / 64 * 64;
        const int M_net2 = std::max(d.group_M2, d.M2 - d.group_M2), s1 = splits - d.splits0;
        d.Mper2 = ((M_net2 + s1 - 1) / s1 + 63) / 64 * 64;
    } else {
        d.Mper = ((M_net + splits - 1) / splits + 63) / 64 * 64;
    }
    if (r2 == nullptr && uig_wgrad_rows_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype, pad_mode) &&
        splits <= (group_images > 0 ? std::min(group_images, B - group_images) : B) * Mh)
        return uig_launch_wgrad_rows(P, Q, workspace, B, Mh, Mw, Np, Cq, pad_mode, splits, group_images, s, stride);
    if (r2 == nullptr && uig_wgrad_head_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype) &&
        splits <= (group_images > 0 ? std::min(group_images, B - group_images) : B) * Mh)
        return uig_launch_wgrad_head(P, Q, workspace, B, Mh, Mw, Np, pad_mode, splits, group_images, s);
    const int bkp = dtype == UIG_BF16 ? 64 : 32;
    const bool fast = (Mw % bkp) == 0;          // a K-step never leaves its image row (Mper is a multiple of bkp)
    if (r2 == nullptr && g_wgrad_bn256 && Np % 256 == 0 && fast) {       // wide dense tile (the caller halves `splits`: uig_wgrad_tile_rows)
        return dtype == UIG_BF16 ? launch_wgrad<bf16_t, 256, true>(P, Q, workspace, d, splits, s)
                                 : launch_wgrad<float, 256, true>(P, Q, workspace, d, splits, s);
    }
    if (dtype == UIG_BF16) {
        if (Np <= 16) return fast ? launch_wgrad<bf16_t, 16, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<bf16_t, 16, false>(P, Q, workspace, d, splits, s, P2, Q2);
        if (Np <= 64) return fast ? launch_wgrad<bf16_t, 64, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<bf16_t, 64, false>(P, Q, workspace, d, splits, s, P2, Q2);
        return fast ? launch_wgrad<bf16_t, 128, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<bf16_t, 128, false>(P, Q, workspace, d, splits, s, P2, Q2);
    }
    if (Np <= 16) return fast ? launch_wgrad<float, 16, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<float, 16, false>(P, Q, workspace, d, splits, s, P2, Q2);
    if (Np <= 64) return fast ? launch_wgrad<float, 64, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<float, 64, false>(P, Q, workspace, d, splits, s, P2, Q2);
    return fast ? launch_wgrad<float, 128, true>(P, Q, workspace, d, splits, s, P2, Q2) : launch_wgrad<float, 128, false>(P, Q, workspace, d, splits, s, P2, Q2);
}

extern "C" int uig_wgrad_partial(const void* P, const void* Q, float* workspace, int B, int Mh, int Mw, int Np,
                                 int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int pad_mode,
                                 int splits, int dtype, void* stream) {
    return wgrad_partial_impl(P, Q, workspace, B, 0, Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, pad_mode, splits, dtype, stream);
}

// Two networks of identical layer shape in ONE partial launch (images [0, group_images) belong to the first, the rest to the
// second): twice the output tiles, so half the pixel splits - half the fp32 partial-slab traffic per network, which is what
// the split-K weight gradient spends most of its time on beyond the MFMAs.
// Workspace: [2][splits][Np][kH*kW*Cq] floats; reduce each network's half with uig_wgrad_reduce*.
extern "C" int uig_wgrad_pair_splits(int B, int group_images, int Mh, int Mw, int Np, int Hq, int Wq, int Cq, int kH, int kW,
                                     int stride, int pad, int dtype, int target_blocks) {
    if (group_images <= 0 || group_images >= B) return 0;
    const int gmin = std::min(group_images, B - group_images);
    if (uig_wgrad_rows_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype))
        return (int)std::max<long>(1, std::min<long>(256 / (2 * uig_wgrad_rows_tiles(Np, Cq)), (long)gmin * Mh));
    if (uig_wgrad_head_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype))
        return uig_wgrad_head_splits(B, group_images, Mh);
    const int bn = uig_wgrad_tile_rows(Np, Mw, dtype);
    const int tiles = 2 * ((Np + bn - 1) / bn) * ((kH * kW * Cq + 127) / 128);
    const long M_net = (long)gmin * Mh * Mw;
    return (int)std::max<long>(1, std::min<long>((bn >= 256 ? 256 : std::max(target_blocks, 1)) / tiles, M_net / 128));
}
extern "C" int uig_wgrad_partial_pair(const void* P, const void* Q, float* workspace, int B, int group_images, int Mh, int Mw,
                                      int Np, int Hq, int Wq, int Cq, int kH, int kW, int stride, int pad, int pad_mode,
                                      int splits, int dtype, void* stream) {
    UIG_CHECK_ARG(group_images > 0 && group_images < B, "uig_wgrad_partial_pair: bad group_images %d of %d", group_images, B);
    return wgrad_partial_impl(P, Q, workspace, B, group_images, Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, pad_mode, splits, dtype, stream);
}

// The weight gradient of one layer pair over TWO batches in one launch (round 2): the step's two generator passes use the same
// two weight sets, pass 1 on (P, Q) = B1 images of which the first g1 are network 0's, pass 2 on (P2, Q2) = B2 images of which
// the first g2 are network (swap2 ? 1 : 0)'s.  One launch instead of two: the fixed cost of a split-K launch (fill / drain, the
// partial slabs and their reduce) is paid once - measured 147 us against 111 + 74 us for 16 + 8 images incl. the reduce.
// Image-row kernel shapes: two runs of whole images per network inside the kernel's row walk.  Every other shape except the 7x7
// head kernel's: the generic split-K kernel with the pixel splits divided between the two tensor pairs (a split never leaves its
// pair); measured per layer pair at 16 + 8 images (scripts/bench_wgrad_combine_generic.py): down1 / up2 200 -> 169 us, down2 / up1
// 122 -> 102 us, stem 270 -> 248 us.  uig_wgrad_pair2_splits returns 0 where the two launches must stay.
// Workspace: [2][splits][Np][kH*kW*Cq] floats, reduced like uig_wgrad_partial_pair's.
int uig_launch_wgrad_rows_runs(const void* P, const void* Q, const void* P2, const void* Q2, float* ws, int B1, int B2, int H, int W, int Np, int Cq,
                               int pad_mode, int splits, const int* imgs, const int* img0, const int* sel, hipStream_t s, int stride = 1);
// generic kernel, two runs: S splits per network in all (the grid of uig_wgrad_pair_splits for the whole batch), divided between
// the runs in proportion to their pixels, at least one each
static void pair2_generic_splits(int B1, int g1, int B2, int g2, int swap2, int Mh, int Mw, int Np, int Cq, int kH, int kW, int dtype,
                                 int* splits, int* splits0) {
    const int bn = uig_wgrad_tile_rows(Np, Mw, dtype);
    const int tiles = 2 * ((Np + bn - 1) / bn) * ((kH * kW * Cq + 127) / 128);
    const long px = (long)Mh * Mw;
    const long m1 = (long)std::min(g1, B1 - g1) * px, m2 = (long)std::min(g2, B2 - g2) * px;
    long S = std::max<long>(2, std::min<long>(512 / std::max(tiles, 1), (m1 + m2) / 128));
    if (*splits >= 2) S = *splits;      // a total the caller already sized its workspace for: keep it, divide it between the runs
    long s0 = std::max<long>(1, std::min<long>(S - 1, (S * ((long)B1 * px) + ((long)(B1 + B2) * px) / 2) / ((long)(B1 + B2) * px)));
    s0 = std::min<long>(s0, std::max<long>(1, m1 / 64));
    long s1 = std::max<long>(1, std::min<long>(S - s0, std::max<long>(1, m2 / 64)));
    (void)swap2;
    *splits = (int)(s0 + s1); *splits0 = (int)s0;
}
extern "C" int uig_wgrad_pair2_splits(int B1, int g1, int B2, int g2, int swap2, int Mh, int Mw, int Np, int Hq, int Wq, int Cq,
                                      int kH, int kW, int stride, int pad, int dtype) {
    if (g1 <= 0 || g1 >= B1 || g2 <= 0 || g2 >= B2) return 0;
    if ((long)(B1 + B2) * Mh * Mw * std::max(Np, Cq) * 2 >= (1L << 32) - 64) return 0;
    if (uig_wgrad_rows_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype)) {
        const int n0 = g1 + (swap2 ? B2 - g2 : g2), n1 = (B1 - g1) + (swap2 ? g2 : B2 - g2);
        return (int)std::max<long>(1, std::min<long>(256 / (2 * uig_wgrad_rows_tiles(Np, Cq)), (long)std::min(n0, n1) * Mh));
    }
    if (uig_wgrad_head_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype))
        return uig_wgrad_head_splits2(B1, g1, B2, g2, swap2, Mh);      // the all-rows 7x7 head kernel: an image is a block's unit, so the second pair is a pointer select
    if (dtype != UIG_BF16 && dtype != UIG_F32) return 0;
    int splits = 0, splits0 = 0;      // 0: the shape's own total
    pair2_generic_splits(B1, g1, B2, g2, swap2, Mh, Mw, Np, Cq, kH, kW, dtype, &splits, &splits0);
    return splits;
}
extern "C" int uig_wgrad_partial_pair2(const void* P, const void* Q, const void* P2, const void* Q2, float* workspace,
                                       int B1, int g1, int B2, int g2, int swap2, int Mh, int Mw, int Np, int Hq, int Wq, int Cq,
                                       int kH, int kW, int stride, int pad, int pad_mode, int splits, int dtype, void* stream) {
    UIG_CHECK_ARG(P && Q && P2 && Q2 && workspace, "uig_wgrad_partial_pair2: null pointer");
    const int want = uig_wgrad_pair2_splits(B1, g1, B2, g2, swap2, Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype);
    UIG_CHECK_ARG(want > 0, "uig_wgrad_partial_pair2: shape not supported (query uig_wgrad_pair2_splits)");
    if (uig_wgrad_head_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype)) {
        UIG_CHECK_ARG(splits == want, "uig_wgrad_partial_pair2: splits %d != uig_wgrad_pair2_splits() = %d", splits, want);
        return uig_launch_wgrad_head_runs(P, Q, P2, Q2, workspace, B1, g1, B2, g2, swap2, Mh, Mw, Np, pad_mode, splits, (hipStream_t)stream);
    }
    if (!uig_wgrad_rows_applicable(Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, dtype, pad_mode)) {
        // generic split-K kernel: the split between the two runs is fixed by the shape, so `splits` must be the queried value
        UIG_CHECK_ARG(splits == want, "uig_wgrad_partial_pair2: splits %d != uig_wgrad_pair2_splits() = %d", splits, want);
        // `splits` (== the query's answer, which does not know the pad mode: for a reflection-padded stride-2 3x3 layer with 64-pixel
        // output rows it is the image-row kernel's count although this generic kernel runs) is what the workspace was sized for: the
        // division between the two runs is made of THAT total, so layout and launch always agree
        int s_all = splits, s0 = 0;
        pair2_generic_splits(B1, g1, B2, g2, swap2, Mh, Mw, Np, Cq, kH, kW, dtype, &s_all, &s0);
        UIG_CHECK_ARG(s_all == splits && s0 >= 1 && s0 < splits, "uig_wgrad_partial_pair2: cannot divide %d splits between the two runs", splits);
        const WgRun2 r2{P2, Q2, B2, g2, swap2, s0};
        return wgrad_partial_impl(P, Q, workspace, B1, g1, Mh, Mw, Np, Hq, Wq, Cq, kH, kW, stride, pad, pad_mode, splits, dtype, stream, &r2);
    }
    const int n0 = g1 + (swap2 ? B2 - g2 : g2), n1 = (B1 - g1) + (swap2 ? g2 : B2 - g2);
    UIG_CHECK_ARG(splits >= 1 && splits <= std::min(n0, n1) * Mh, "uig_wgrad_partial_pair2: bad splits %d", splits);
    // network 0: pass-1 images [0, g1), then pass-2's share; network 1: pass-1 images [g1, B1), then pass-2's share
    const int imgs[4] = {g1, swap2 ? B2 - g2 : g2, B1 - g1, swap2 ? g2 : B2 - g2};
    const int img0[4] = {0, swap2 ? g2 : 0, g1, swap2 ? 0 : g2};
    const int sel[4] = {0, 1, 0, 1};
    return uig_launch_wgrad_rows_runs(P, Q, P2, Q2, workspace, B1, B2, Mh, Mw, Np, Cq, pad_mode, splits, imgs, img0, sel, (hipStream_t)stream, stride);
}

static int wgrad_reduce_impl(const float* workspace, float* dW, int Np, int Cq, int taps, int splits, int D0, int D1, int accumulate,
                             BiasRider br, void* stream, float* dW2 = nullptr, BiasRider br2 = BiasRider{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0}) {
    UIG_CHECK_ARG(workspace && dW, "uig_wgrad_reduce: null pointer");
    UIG_CHECK_ARG(D0 <= Np && D1 <= Cq && D0 > 0 && D1 > 0 && taps > 0 && splits > 0, "uig_wgrad_reduce: bad dims");
    const long total = (long)D0 * D1;
    hipStream_t s = (hipStream_t)stream;
    const int extra = (br.cpart || br2.cpart) ? (std::max(br.C, br2.C) + 15) / 16 : 0;
    const int ny = dW2 ? 2 : 1;                        // paired launch: blockIdx.y = network
    if (total < 8192) {
        const long tot_e = total * taps;
        br.main_blocks = br2.main_blocks = (int)std::max<long>(1, std::min<long>((tot_e + 15) / 16, 4096));
        hipLaunchKernelGGL(wgrad_reduce_elem_kernel, dim3(br.main_blocks + extra, ny), dim3(256), 0, s, workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate, br, dW2, br2);
    } else {
        br.main_blocks = br2.main_blocks = (int)std::max<long>(1, std::min<long>((total + 255) / 256, 4096));
        const dim3 g(br.main_blocks + extra, ny);
        if (taps == 9) hipLaunchKernelGGL(wgrad_reduce_kernel<9>, g, dim3(256), 0, s, workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate, br, dW2, br2);
        else if (taps == 16) hipLaunchKernelGGL(wgrad_reduce_kernel<16>, g, dim3(256), 0, s, workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate, br, dW2, br2);
        else hipLaunchKernelGGL(wgrad_reduce_kernel<0>, g, dim3(256), 0, s, workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate, br, dW2, br2);
    }
    UIG_LAUNCH_CHECK("uig_wgrad_reduce");
    return 0;
}

// Both halves of a uig_wgrad_partial_pair workspace in ONE launch: network a -> dW_a (+ optional bias rider a), network b ->
// dW_b.  colsum_* may be NULL (no bias gradient on this launch).
extern "C" int uig_wgrad_reduce_pair(const float* workspace, float* dW_a, float* dW_b, int Np, int Cq, int taps, int splits,
                                     int D0, int D1, int accumulate, const float* colsum_a, const float* colsum_b,
                                     int nslab_a, int nslab_b, int C, int Nreal, float* db_a, float* db_b, int accumulate_db,
                                     void* stream) {
    UIG_CHECK_ARG(dW_a && dW_b, "uig_wgrad_reduce_pair: null pointer");
    UIG_CHECK_ARG((colsum_a == nullptr) == (colsum_b == nullptr), "uig_wgrad_reduce_pair: bias riders for both networks or for none");
    BiasRider ra{nullptr, nullptr, 0, 0, 0, 0, 0}, rb = ra;
    if (colsum_a != nullptr) {
        UIG_CHECK_ARG(db_a && db_b && nslab_a > 0 && nslab_b > 0 && Nreal > 0 && Nreal <= C, "uig_wgrad_reduce_pair: bad bias args");
        ra = BiasRider{colsum_a, db_a, nslab_a, C, Nreal, accumulate_db, 0, nullptr, 0};
        rb = BiasRider{colsum_b, db_b, nslab_b, C, Nreal, accumulate_db, 0, nullptr, 0};
    }
    return wgrad_reduce_impl(workspace, dW_a, Np, Cq, taps, splits, D0, D1, accumulate, ra, stream, dW_b, rb);
}

// uig_wgrad_reduce_pair whose bias riders sum TWO runs of column-sum slabs per network (the two generator passes whose weight
// gradients uig_wgrad_partial_pair2 reduced together): db_x (+)= sum colsum_x[0..nslab_x) + sum colsum_x2[0..nslab_x2).
extern "C" int uig_wgrad_reduce_pair2(const float* workspace, float* dW_a, float* dW_b, int Np, int Cq, int taps, int splits,
                                      int D0, int D1, int accumulate, const float* colsum_a, const float* colsum_b, int nslab_a, int nslab_b,
                                      const float* colsum_a2, const float* colsum_b2, int nslab_a2, int nslab_b2,
                                      int C, int Nreal, float* db_a, float* db_b, int accumulate_db, void* stream) {
    UIG_CHECK_ARG(dW_a && dW_b && colsum_a && colsum_b && colsum_a2 && colsum_b2 && db_a && db_b, "uig_wgrad_reduce_pair2: null pointer");
    UIG_CHECK_ARG(nslab_a > 0 && nslab_b > 0 && nslab_a2 > 0 && nslab_b2 > 0 && Nreal > 0 && Nreal <= C, "uig_wgrad_reduce_pair2: bad bias args");
    const BiasRider ra{colsum_a, db_a, nslab_a, C, Nreal, accumulate_db, 0, colsum_a2, nslab_a2};
    const BiasRider rb{colsum_b, db_b, nslab_b, C, Nreal, accumulate_db, 0, colsum_b2, nslab_b2};
    return wgrad_reduce_impl(workspace, dW_a, Np, Cq, taps, splits, D0, D1, accumulate, ra, stream, dW_b, rb);
}

extern "C" int uig_wgrad_reduce(const float* workspace, float* dW, int Np, int Cq, int taps, int splits,
                                int D0, int D1, int accumulate, void* stream) {
    return wgrad_reduce_impl(workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate, BiasRider{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr, 0}, stream);
}

// the same launch also finishes the layer's bias gradient from column-sum partials (uig_instnorm_act_bwd_colsum)
extern "C" int uig_wgrad_reduce_bias(const float* workspace, float* dW, int Np, int Cq, int taps, int splits,
                                     int D0, int D1, int accumulate, const float* colsum_partial, int nslab_total, int C,
                                     int Nreal, float* db, int accumulate_db, void* stream) {
    UIG_CHECK_ARG(colsum_partial && db && nslab_total > 0 && Nreal > 0 && Nreal <= C, "uig_wgrad_reduce_bias: bad bias args");
    return wgrad_reduce_impl(workspace, dW, Np, Cq, taps, splits, D0, D1, accumulate,
                             BiasRider{colsum_partial, db, nslab_total, C, Nreal, accumulate_db, 0, nullptr, 0}, stream);
}
