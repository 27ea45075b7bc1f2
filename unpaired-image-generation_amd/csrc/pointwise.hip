// pointwise.hip — the HBM-bound pointwise / small-reduction kernels of the train step (gfx950):
// fused loss forward+gradient (L1, MSE-vs-constant), flat Adam, weight packing, reflection-pad fold, activation
// backward, layout plumbing.  16-byte accesses per lane, grid-stride loops, deterministic two-stage reductions.
#include "uig_common.h"
#include <algorithm>

static inline int grid_for(long n, int per_thread = 1) {
    long b = (n + 256L * per_thread - 1) / (256L * per_thread);
    return (int)std::max<long>(1, std::min<long>(b, 2048));
}

// ------------------------------------------------------------------ weight packing
// src fp32 (D0, D1, kH, kW) -> dst[row][tap][col] with row = d(row_dim), col = the other dim, zero padded
template <typename T>
__global__ void pack_weight_kernel(const float* __restrict__ w, T* __restrict__ wp, int D0, int D1, int taps, int row_dim,
                                   int flip, int rows_p, int cols_p) {
    const long total = (long)rows_p * taps * cols_p;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int col = (int)(i % cols_p); const long r = i / cols_p; const int tap = (int)(r % taps); const int row = (int)(r / taps);
        const int d0 = row_dim == 0 ? row : col, d1 = row_dim == 0 ? col : row;
        float v = 0.f;
        if (d0 < D0 && d1 < D1) v = w[((long)d0 * D1 + d1) * taps + (flip ? taps - 1 - tap : tap)];
        ElemTraits<T>::st(wp + i, v);
    }
}
extern "C" int uig_pack_weight(const float* w, void* wp, int D0, int D1, int kH, int kW, int row_dim, int flip,
                               int rows_padded, int cols_padded, int dtype, void* stream) {
    UIG_CHECK_ARG(w && wp, "uig_pack_weight: null pointer");
    UIG_CHECK_ARG(row_dim == 0 || row_dim == 1, "uig_pack_weight: bad row_dim");
    const int R = row_dim == 0 ? D0 : D1, Cc = row_dim == 0 ? D1 : D0;
    UIG_CHECK_ARG(rows_padded >= R && cols_padded >= Cc && cols_padded % 8 == 0, "uig_pack_weight: bad padding rows %d>=%d cols %d>=%d", rows_padded, R, cols_padded, Cc);
    const long total = (long)rows_padded * kH * kW * cols_padded;
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, (bf16_t*)wp, D0, D1, kH * kW, row_dim, flip, rows_padded, cols_padded);
    else if (dtype == UIG_F32)
        hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, w, (float*)wp, D0, D1, kH * kW, row_dim, flip, rows_padded, cols_padded);
    else return uig_set_error(-1, "uig_pack_weight: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_pack_weight");
    return 0;
}

// All layers of all networks in ONE launch (140 separate pack launches cost 0.67 ms per step).  items[] lives in device
// memory: per layer and operand one record with absolute pointers; work_end is the inclusive prefix sum of TILES.
// A block transposes one source tile of 8 (d0) x TD1 (d1) x taps floats through LDS: the source rows w[d0][d1base..][*] are
// contiguous runs (coalesced reads), and every thread then writes one whole 16-byte chunk of the packed operand
// ([d0][tap][d1 chunk] for row_dim 0, [d1][tap][d0 chunk] for row_dim 1).  The first version (one thread per output
// element, a 7-step binary search over the item table each) took 0.43 ms per step; this one is bandwidth-shaped.
struct PackItem {
    const float* w; void* dst;
    int D0, D1, taps, row_dim, rows_p, cols_p;
    long work_end;
};
static_assert(sizeof(PackItem) == 48, "PackItem layout is mirrored by the Python host code");
namespace {
constexpr int PK_TD0 = 8, PK_MAXSEG = 64 * 16;               // 8 d0 rows; at most 1024 floats of one row per tile
__host__ __device__ inline int pack_td1(int taps) { return taps <= 16 ? 64 : 16; }
}
extern "C" int uig_pack_tiles(int D0, int D1, int kH, int kW, int row_dim, int rows_padded, int cols_padded) {
    const int taps = kH * kW;
    if (taps > 64 || D0 <= 0 || D1 <= 0) return -1;
    // the tile grid covers the real d0 x d1 range plus the zero padding of the column dimension
    const int c0 = row_dim == 0 ? D0 : std::max(D0, cols_padded), c1 = row_dim == 0 ? std::max(D1, cols_padded) : D1;
    (void)rows_padded;
    return ((c0 + PK_TD0 - 1) / PK_TD0) * ((c1 + pack_td1(taps) - 1) / pack_td1(taps));
}
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi_kernel(const PackItem* __restrict__ items, int nitems, long total) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float seg[PK_TD0][PK_MAXSEG + 1];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    for (long blk = blockIdx.x; blk < total; blk += gridDim.x) {
        if (tid == 0) {
            int lo = 0, hi = nitems - 1;                 // first item whose work_end > blk
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].work_end > blk) hi = mid; else lo = mid + 1; }
            s_item = lo;
        }
        __syncthreads();
        const int ii = s_item;
        const PackItem it = items[ii];
        const long j = blk - (ii ? items[ii - 1].work_end : 0);
        const int taps = it.taps, TD1 = pack_td1(taps);
        const int c1 = it.row_dim == 0 ? max(it.D1, it.cols_p) : it.D1;
        const int n1 = (c1 + TD1 - 1) / TD1;
        const int d0b = (int)(j / n1) * PK_TD0, d1b = (int)(j % n1) * TD1;
        // ---- load: seg[g][i] = w[d0b + g][d1b + i / taps][i % taps], zero outside the real tensor
        const int nd1 = min(TD1, it.D1 - d1b);           // real d1 entries in this tile (may be <= 0: pure padding tile)
        const int len = max(nd1, 0) * taps;
        for (int g = 0; g < PK_TD0; ++g) {
            const bool rok = d0b + g < it.D0;
            const float* src = it.w + ((long)(d0b + g) * it.D1 + d1b) * taps;
            for (int i = tid; i < TD1 * taps; i += 256) seg[g][i] = (rok && i < len) ? src[i] : 0.f;
        }
        __syncthreads();
        T* dst = static_cast<T*>(it.dst);
        if (it.row_dim == 0) {          // dst[d0][tap][d1]: chunks of E consecutive d1
            const int cpt = TD1 / E;                     // chunks per (d0, tap)
            for (int c = tid; c < PK_TD0 * taps * cpt; c += 256) {
                const int k = c % cpt, tap = (c / cpt) % taps, g = c / (cpt * taps);
                const int d0 = d0b + g, d1 = d1b + k * E;
                if (d0 >= it.rows_p || d1 >= it.cols_p) continue;
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = seg[g][(k * E + e) * taps + tap];
                *reinterpret_cast<u32x4_t*>(dst + ((long)d0 * taps + tap) * it.cols_p + d1) = f32_to_chunk<T>(v);
            }
        } else {                         // dst[d1][tap][d0]: chunks of E consecutive d0 (the tile's 8 d0 rows = 8 / E chunks)
            constexpr int CPG = PK_TD0 / E;
            for (int c = tid; c < TD1 * taps * CPG; c += 256) {
                const int k = c % CPG, tap = (c / CPG) % taps, i1 = c / (CPG * taps);
                const int d1 = d1b + i1, d0 = d0b + k * E;
                if (d1 >= it.rows_p || d0 >= it.cols_p) continue;
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = seg[k * E + e][i1 * taps + tap];
                *reinterpret_cast<u32x4_t*>(dst + ((long)d1 * taps + tap) * it.cols_p + d0) = f32_to_chunk<T>(v);
            }
        }
        __syncthreads();
    }
}
// Round 3: BOTH kernel-side operands of a layer from ONE pass over its fp32 weights (the forward and the input-gradient operand are
// the two transpositions of the same tile: [d0][tap][d1 chunk] and [d1][tap][d0 chunk]).  The two-record form read every weight twice
// (273 MB moved per generator pack, 83 us); this one reads it once.  Record: the first operand as in PackItem (row_dim, cols_p), the
// second with the other row_dim; the tile grid covers the union of both operands' padded ranges.
struct PackItem2 {
    const float* w; void* dst; void* dst2;
    int D0, D1, taps, row_dim, cols_p, cols2_p;          // dst: rows = dim row_dim, columns padded to cols_p; dst2: rows = the other dim, columns padded to cols2_p
    long work_end;
};
static_assert(sizeof(PackItem2) == 56, "PackItem2 layout is mirrored by the Python host code");
static __host__ __device__ inline void pack2_ranges(int D0, int D1, int row_dim, int cols_p, int cols2_p, int* c0, int* c1) {
    // operand 1 pads the dimension that is NOT its row; operand 2 pads the one that is operand 1's row
    if (row_dim == 0) { *c0 = D0 > cols2_p ? D0 : cols2_p; *c1 = D1 > cols_p ? D1 : cols_p; }
    else              { *c0 = D0 > cols_p ? D0 : cols_p;   *c1 = D1 > cols2_p ? D1 : cols2_p; }
}
extern "C" int uig_pack_tiles2(int D0, int D1, int kH, int kW, int row_dim, int cols_padded, int cols2_padded) {
    const int taps = kH * kW;
    if (taps > 64 || D0 <= 0 || D1 <= 0 || (row_dim != 0 && row_dim != 1)) return -1;
    int c0, c1;
    pack2_ranges(D0, D1, row_dim, cols_padded, cols2_padded, &c0, &c1);
    return ((c0 + PK_TD0 - 1) / PK_TD0) * ((c1 + pack_td1(taps) - 1) / pack_td1(taps));
}
template <typename T>
__global__ __launch_bounds__(256) void pack_weights_multi2_kernel(const PackItem2* __restrict__ items, int nitems, long total) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float seg[PK_TD0][PK_MAXSEG + 1];
    __shared__ int s_item;
    const int tid = threadIdx.x;
    for (long blk = blockIdx.x; blk < total; blk += gridDim.x) {
        if (tid == 0) {
            int lo = 0, hi = nitems - 1;                 // first item whose work_end > blk
            while (lo < hi) { const int mid = (lo + hi) >> 1; if (items[mid].work_end > blk) hi = mid; else lo = mid + 1; }
            s_item = lo;
        }
        __syncthreads();
        const int ii = s_item;
        const PackItem2 it = items[ii];
        const long j = blk - (ii ? items[ii - 1].work_end : 0);
        const int taps = it.taps, TD1 = pack_td1(taps);
        int c0, c1;
        pack2_ranges(it.D0, it.D1, it.row_dim, it.cols_p, it.cols2_p, &c0, &c1);
        const int n1 = (c1 + TD1 - 1) / TD1;
        const int d0b = (int)(j / n1) * PK_TD0, d1b = (int)(j % n1) * TD1;
        const int nd1 = min(TD1, it.D1 - d1b);
        const int len = max(nd1, 0) * taps;
        for (int g = 0; g < PK_TD0; ++g) {
            const bool rok = d0b + g < it.D0;
            const float* src = it.w + ((long)(d0b + g) * it.D1 + d1b) * taps;
            for (int i = tid; i < TD1 * taps; i += 256) seg[g][i] = (rok && i < len) ? src[i] : 0.f;
        }
        __syncthreads();
        // operand with rows = d0 ([d0][tap][d1 chunk]) and operand with rows = d1 ([d1][tap][d0 chunk]); rows are never padded
        T* dstA = static_cast<T*>(it.row_dim == 0 ? it.dst : it.dst2);
        T* dstB = static_cast<T*>(it.row_dim == 0 ? it.dst2 : it.dst);
        const int colsA = it.row_dim == 0 ? it.cols_p : it.cols2_p, colsB = it.row_dim == 0 ? it.cols2_p : it.cols_p;
        {
            const int cpt = TD1 / E;
            for (int c = tid; c < PK_TD0 * taps * cpt; c += 256) {
                const int k = c % cpt, tap = (c / cpt) % taps, g = c / (cpt * taps);
                const int d0 = d0b + g, d1 = d1b + k * E;
                if (d0 >= it.D0 || d1 >= colsA) continue;
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = seg[g][(k * E + e) * taps + tap];
                *reinterpret_cast<u32x4_t*>(dstA + ((long)d0 * taps + tap) * colsA + d1) = f32_to_chunk<T>(v);
            }
        }
        {
            constexpr int CPG = PK_TD0 / E;
            for (int c = tid; c < TD1 * taps * CPG; c += 256) {
                const int k = c % CPG, tap = (c / CPG) % taps, i1 = c / (CPG * taps);
                const int d1 = d1b + i1, d0 = d0b + k * E;
                if (d1 >= it.D1 || d0 >= colsB) continue;
                float v[E];
#pragma unroll
                for (int e = 0; e < E; ++e) v[e] = seg[k * E + e][i1 * taps + tap];
                *reinterpret_cast<u32x4_t*>(dstB + ((long)d1 * taps + tap) * colsB + d0) = f32_to_chunk<T>(v);
            }
        }
        __syncthreads();
    }
}
extern "C" int uig_pack_weights_multi2(const void* items_dev, int nitems, int64_t total_work, int dtype, void* stream) {
    UIG_CHECK_ARG(items_dev && nitems > 0 && total_work > 0, "uig_pack_weights_multi2: bad args");
    const int g = (int)std::min<long>(total_work, 8192);
    if (dtype == UIG_BF16) hipLaunchKernelGGL((pack_weights_multi2_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const PackItem2*)items_dev, nitems, (long)total_work);
    else if (dtype == UIG_F32) hipLaunchKernelGGL((pack_weights_multi2_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const PackItem2*)items_dev, nitems, (long)total_work);
    else return uig_set_error(-1, "uig_pack_weights_multi2: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_pack_weights_multi2");
    return 0;
}

extern "C" int uig_pack_weights_multi(const void* items_dev, int nitems, int64_t total_work, int dtype, void* stream) {
    UIG_CHECK_ARG(items_dev && nitems > 0 && total_work > 0, "uig_pack_weights_multi: bad args");
    const int g = (int)std::min<long>(total_work, 8192);
    if (dtype == UIG_BF16) hipLaunchKernelGGL((pack_weights_multi_kernel<bf16_t>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev, nitems, (long)total_work);
    else if (dtype == UIG_F32) hipLaunchKernelGGL((pack_weights_multi_kernel<float>), dim3(g), dim3(256), 0, (hipStream_t)stream, (const PackItem*)items_dev, nitems, (long)total_work);
    else return uig_set_error(-1, "uig_pack_weights_multi: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_pack_weights_multi");
    return 0;
}

// ------------------------------------------------------------------ reflection-pad backward (fold)
template <typename T>
__global__ void reflect_fold_kernel(const T* __restrict__ dyp, T* __restrict__ dx, int B, int H, int W, int C, int P) {
    constexpr int E = ElemTraits<T>::E;
    const int CC = C / E, Hp = H + 2 * P, Wp = W + 2 * P;
    const long total = (long)B * H * W * CC;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int cc = (int)(i % CC); long r = i / CC; const int w = (int)(r % W); r /= W; const int h = (int)(r % H); const int b = (int)(r / H);
        // padded coordinates that mirror onto (h, w): the interior image plus up to one reflection per axis
        int hs[3], ws[3], nh = 1, nw = 1;
        hs[0] = h + P; ws[0] = w + P;
        if (h >= 1 && h <= P) hs[nh++] = P - h;
        if (h <= H - 2 && h >= H - 1 - P) hs[nh++] = 2 * (H - 1) - h + P;
        if (w >= 1 && w <= P) ws[nw++] = P - w;
        if (w <= W - 2 && w >= W - 1 - P) ws[nw++] = 2 * (W - 1) - w + P;
        float acc[E];
#pragma unroll
        for (int e = 0; e < E; ++e) acc[e] = 0.f;
        for (int a = 0; a < nh; ++a)
            for (int c = 0; c < nw; ++c) {
                float v[E];
                chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(dyp + (((long)b * Hp + hs[a]) * Wp + ws[c]) * C + cc * E), v);
#pragma unroll
                for (int e = 0; e < E; ++e) acc[e] += v[e];
            }
        *reinterpret_cast<u32x4_t*>(dx + (((long)b * H + h) * W + w) * C + cc * E) = f32_to_chunk<T>(acc);
    }
}
extern "C" int uig_reflect_fold(const void* dyp, void* dx, int B, int H, int W, int C, int pad, int dtype, void* stream) {
    UIG_CHECK_ARG(dyp && dx, "uig_reflect_fold: null pointer");
    UIG_CHECK_ARG(C % 8 == 0 && pad >= 1 && 2 * pad < H && 2 * pad < W, "uig_reflect_fold: bad shape C=%d pad=%d H=%d W=%d (needs 2*pad < dim)", C, pad, H, W);
    const long total = (long)B * H * W * (C / (dtype == UIG_BF16 ? 8 : 4));
    if (dtype == UIG_BF16) hipLaunchKernelGGL((reflect_fold_kernel<bf16_t>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dyp, (bf16_t*)dx, B, H, W, C, pad);
    else if (dtype == UIG_F32) hipLaunchKernelGGL((reflect_fold_kernel<float>), dim3(grid_for(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dyp, (float*)dx, B, H, W, C, pad);
    else return uig_set_error(-1, "uig_reflect_fold: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_reflect_fold");
    return 0;
}

// ------------------------------------------------------------------ activation backward on the OUTPUT y
template <typename T>
__global__ void act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx, long nchunks, int act, float slope) {
    constexpr int E = ElemTraits<T>::E;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        float g[E], v[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(dy + i * E), g);
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(y + i * E), v);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if (act == UIG_ACT_TANH) g[e] *= (1.f - v[e] * v[e]);
            else if (act == UIG_ACT_RELU) g[e] = v[e] > 0.f ? g[e] : 0.f;
            else if (act == UIG_ACT_LRELU) g[e] = v[e] > 0.f ? g[e] : g[e] * slope;
        }
        *reinterpret_cast<u32x4_t*>(dx + i * E) = f32_to_chunk<T>(g);
    }
}
extern "C" int uig_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(dy && y && dx && n > 0 && n % 8 == 0, "uig_act_bwd: bad args (n=%ld must be a multiple of 8)", (long)n);
    if (dtype == UIG_BF16) hipLaunchKernelGGL((act_bwd_kernel<bf16_t>), dim3(grid_for(n / 8)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, (const bf16_t*)y, (bf16_t*)dx, (long)(n / 8), act, slope);
    else if (dtype == UIG_F32) hipLaunchKernelGGL((act_bwd_kernel<float>), dim3(grid_for(n / 4)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, (const float*)y, (float*)dx, (long)(n / 4), act, slope);
    else return uig_set_error(-1, "uig_act_bwd: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_act_bwd");
    return 0;
}

// ------------------------------------------------------------------ fused losses (forward + gradient in one pass)
__device__ __forceinline__ float block_sum_256(float v, float* sh) {
    v = wave_sum(v);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
    __syncthreads();
    float r = (threadIdx.x < 4) ? sh[threadIdx.x] : 0.f;
    r = wave_sum(r);   // every wave computes it; wave 0's value is the block sum
    __syncthreads();
    return r;
}

// LOSS 0: L1(a, b)   LOSS 1: MSE(a, const)
template <typename T, int LOSS>
__global__ __launch_bounds__(256) void loss_partial_kernel(const T* __restrict__ a, const T* __restrict__ b, float target,
                                                            T* __restrict__ grad, float* __restrict__ partial,
                                                            long nchunks, long tail_start, long n, float gscale,
                                                            float* __restrict__ loss_single, float loss_scale) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float sh[4];
    float s = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nchunks; i += (long)gridDim.x * blockDim.x) {
        float av[E], bv[E], g[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(a + i * E), av);
        if constexpr (LOSS == 0) chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(b + i * E), bv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            if constexpr (LOSS == 0) {
                const float dlt = av[e] - bv[e];
                s += fabsf(dlt);
                g[e] = dlt > 0.f ? gscale : (dlt < 0.f ? -gscale : 0.f);
            } else {
                const float dlt = av[e] - target;
                s += dlt * dlt;
                g[e] = 2.f * dlt * gscale;
            }
        }
        if (grad != nullptr) *reinterpret_cast<u32x4_t*>(grad + i * E) = f32_to_chunk<T>(g);
    }
    // scalar tail (n not a multiple of the chunk size): handled by block 0
    if (blockIdx.x == 0) {
        for (long i = tail_start + threadIdx.x; i < n; i += blockDim.x) {
            const float av = ElemTraits<T>::ld(a + i);
            float g;
            if constexpr (LOSS == 0) { const float dlt = av - ElemTraits<T>::ld(b + i); s += fabsf(dlt); g = dlt > 0.f ? gscale : (dlt < 0.f ? -gscale : 0.f); }
            else { const float dlt = av - target; s += dlt * dlt; g = 2.f * dlt * gscale; }
            if (grad != nullptr) ElemTraits<T>::st(grad + i, g);
        }
    }
    const float bs = block_sum_256(s, sh);
    if (threadIdx.x == 0) {
        if (loss_single != nullptr) loss_single[0] = bs * loss_scale;      // one-block launch (small tensors): no second kernel
        else partial[blockIdx.x] = bs;
    }
}
__global__ __launch_bounds__(256) void loss_final_kernel(const float* __restrict__ partial, int np, float* __restrict__ loss, double scale) {
    __shared__ double shd[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < np; i += 256) s += (double)partial[i];
    shd[threadIdx.x] = s;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if (threadIdx.x < o) shd[threadIdx.x] += shd[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) loss[0] = (float)(shd[0] * scale);
}
extern "C" size_t uig_loss_workspace_floats(void) { return 2048; }

template <int LOSS>
static int launch_loss(const void* a, const void* b, float target, float* loss, void* grad, float* ws, long n, long n_real,
                       float weight, int dtype, hipStream_t s, const char* name) {
    UIG_CHECK_ARG(a && loss && ws && n > 0 && n_real > 0, "%s: bad args", name);
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "%s: bad dtype %d", name, dtype);
    const int E = dtype == UIG_BF16 ? 8 : 4;
    const long nchunks = n / E, tail = nchunks * E;
    const int blocks = grid_for(std::max<long>(nchunks, 1), 2);
    const float gscale = weight / (float)n_real;
    float* single = blocks == 1 ? loss : nullptr;          // the PatchGAN score maps (900 values per image) fit one block
    if (dtype == UIG_BF16) hipLaunchKernelGGL((loss_partial_kernel<bf16_t, LOSS>), dim3(blocks), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, target, (bf16_t*)grad, ws, nchunks, tail, n, gscale, single, gscale);
    else hipLaunchKernelGGL((loss_partial_kernel<float, LOSS>), dim3(blocks), dim3(256), 0, s, (const float*)a, (const float*)b, target, (float*)grad, ws, nchunks, tail, n, gscale, single, gscale);
    UIG_LAUNCH_CHECK(name);
    if (single == nullptr) {
        hipLaunchKernelGGL(loss_final_kernel, dim3(1), dim3(256), 0, s, ws, blocks, loss, (double)weight / (double)n_real);
        UIG_LAUNCH_CHECK(name);
    }
    return 0;
}
extern "C" int uig_l1_loss_fwd_bwd(const void* a, const void* b, float* loss, void* grad_a, float* workspace,
                                   int64_t n, int64_t n_real, float weight, int dtype, void* stream) {
    UIG_CHECK_ARG(b, "uig_l1_loss_fwd_bwd: null b");
    return launch_loss<0>(a, b, 0.f, loss, grad_a, workspace, n, n_real, weight, dtype, (hipStream_t)stream, "uig_l1_loss_fwd_bwd");
}
extern "C" int uig_mse_const_fwd_bwd(const void* a, float target, float* loss, void* grad_a, float* workspace,
                                     int64_t n, float weight, int dtype, void* stream) {
    return launch_loss<1>(a, nullptr, target, loss, grad_a, workspace, n, n, weight, dtype, (hipStream_t)stream, "uig_mse_const_fwd_bwd");
}

template <typename T>
__global__ void scale_by_scalar_kernel(const T* __restrict__ g, const float* __restrict__ sc, T* __restrict__ o, long n) {
    const float s = sc[0];
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        ElemTraits<T>::st(o + i, ElemTraits<T>::ld(g + i) * s);
}
extern "C" int uig_scale_by_scalar(const void* g, const float* scalar, void* g_out, int64_t n, int dtype, void* stream) {
    UIG_CHECK_ARG(g && scalar && g_out && n > 0, "uig_scale_by_scalar: bad args");
    if (dtype == UIG_BF16) hipLaunchKernelGGL((scale_by_scalar_kernel<bf16_t>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)g, scalar, (bf16_t*)g_out, (long)n);
    else if (dtype == UIG_F32) hipLaunchKernelGGL((scale_by_scalar_kernel<float>), dim3(grid_for(n, 4)), dim3(256), 0, (hipStream_t)stream, (const float*)g, scalar, (float*)g_out, (long)n);
    else return uig_set_error(-1, "uig_scale_by_scalar: bad dtype %d", dtype);
    UIG_LAUNCH_CHECK("uig_scale_by_scalar");
    return 0;
}

// ------------------------------------------------------------------ Adam over a flat fp32 buffer
// torch.optim.Adam / aten::_fused_adam (amsgrad=False, maximize=False, weight_decay=0):
//   m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= (lr / bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adam_flat_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                                 long n, float b1, float b2, float eps, float step_size, float inv_sqrt_bc2, float gscale,
                                 const float* __restrict__ dev_state) {
    if (dev_state != nullptr) { step_size = dev_state[1]; inv_sqrt_bc2 = dev_state[2]; }   // graph replay: scalars live on the device
    const long n4 = n >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
        f32x4_t pv = reinterpret_cast<f32x4_t*>(p)[i], gv = reinterpret_cast<const f32x4_t*>(g)[i];
        f32x4_t mv = reinterpret_cast<f32x4_t*>(m)[i], vv = reinterpret_cast<f32x4_t*>(v)[i];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gv[e] * gscale;
            mv[e] = b1 * mv[e] + (1.f - b1) * gg;
            vv[e] = b2 * vv[e] + (1.f - b2) * gg * gg;
            pv[e] -= step_size * mv[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
        }
        reinterpret_cast<f32x4_t*>(p)[i] = pv; reinterpret_cast<f32x4_t*>(m)[i] = mv; reinterpret_cast<f32x4_t*>(v)[i] = vv;
    }
    if (blockIdx.x == 0)
        for (long i = (n4 << 2) + threadIdx.x; i < n; i += blockDim.x) {
            const float gg = g[i] * gscale;
            const float mm = b1 * m[i] + (1.f - b1) * gg, vv = b2 * v[i] + (1.f - b2) * gg * gg;
            m[i] = mm; v[i] = vv;
            p[i] -= step_size * mm / (sqrtf(vv) * inv_sqrt_bc2 + eps);
        }
}
extern "C" int uig_adam_flat(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                             float eps, int step, float grad_scale, void* stream) {
    UIG_CHECK_ARG(p && g && m && v && n > 0 && step >= 1, "uig_adam_flat: bad args");
    UIG_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) == 0, "uig_adam_flat: buffers must be 16-byte aligned");
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for(n / 4 + 1, 2)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                       beta1, beta2, eps, (float)(lr / bc1), (float)(1.0 / sqrt(bc2)), grad_scale, (const float*)nullptr);
    UIG_LAUNCH_CHECK("uig_adam_flat");
    return 0;
}

// Graph-replayable form: the step counter and the bias-correction scalars live in a 16-byte device record
// state = {int step; float lr*scale/bc1; float 1/sqrt(bc2); float lr_scale}; every call increments step on the device first.
// lr_scale is the schedule's multiplier: the host writes it into the record between replays (no re-capture).
__global__ void adam_tick_kernel(int* st, float lr, float b1, float b2) {
    const int s = st[0] + 1;
    st[0] = s;
    const double bc1 = 1.0 - pow((double)b1, (double)s), bc2 = 1.0 - pow((double)b2, (double)s);
    reinterpret_cast<float*>(st)[1] = (float)((double)lr * (double)reinterpret_cast<float*>(st)[3] / bc1);
    reinterpret_cast<float*>(st)[2] = (float)(1.0 / sqrt(bc2));
}
extern "C" int uig_adam_flat_graph(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
                                   float eps, void* state16, float grad_scale, void* stream) {
    UIG_CHECK_ARG(p && g && m && v && state16 && n > 0, "uig_adam_flat_graph: bad args");
    UIG_CHECK_ARG((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v | (uintptr_t)state16) & 15) == 0, "uig_adam_flat_graph: buffers must be 16-byte aligned");
    hipLaunchKernelGGL(adam_tick_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, (int*)state16, lr, beta1, beta2);
    UIG_LAUNCH_CHECK("uig_adam_flat_graph(tick)");
    hipLaunchKernelGGL(adam_flat_kernel, dim3(grid_for(n / 4 + 1, 2)), dim3(256), 0, (hipStream_t)stream, p, g, m, v, (long)n,
                       beta1, beta2, eps, 0.f, 0.f, grad_scale, (const float*)state16);
    UIG_LAUNCH_CHECK("uig_adam_flat_graph");
    return 0;
}

// ------------------------------------------------------------------ layout plumbing at the module surface
template <typename TS, typename TD>
__global__ void to_nhwc_kernel(const TS* __restrict__ src, long sb, long sc, long sh, long sw, TD* __restrict__ dst,
                               int B, int C, int H, int W, int Cp) {
    const long total = (long)B * H * W * Cp;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % Cp); long r = i / Cp; const int w = (int)(r % W); r /= W; const int h = (int)(r % H); const int b = (int)(r / H);
        float v = 0.f;
        if (c < C) v = ElemTraits<TS>::ld(src + b * sb + c * sc + h * sh + w * sw);
        ElemTraits<TD>::st(dst + i, v);
    }
}
template <typename TS, typename TD>
__global__ void from_nhwc_kernel(const TS* __restrict__ src, TD* __restrict__ dst, long sb, long sc, long sh, long sw,
                                 int B, int C, int H, int W, int Cp) {
    const long total = (long)B * C * H * W;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        // iterate in (b, c, h, w) order: coalesced writes for the usual contiguous NCHW destination
        const int w = (int)(i % W); long r = i / W; const int h = (int)(r % H); r /= H; const int c = (int)(r % C); const int b = (int)(r / C);
        ElemTraits<TD>::st(dst + b * sb + c * sc + h * sh + w * sw, ElemTraits<TS>::ld(src + (((long)b * H + h) * W + w) * Cp + c));
    }
}
extern "C" int uig_to_nhwc(const void* src, int src_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw,
                           void* dst, int B, int C, int H, int W, int Cp, int dtype, void* stream) {
    UIG_CHECK_ARG(src && dst && Cp >= C, "uig_to_nhwc: bad args (C=%d Cp=%d)", C, Cp);
    const long total = (long)B * H * W * Cp;
    hipStream_t s = (hipStream_t)stream; const int g = grid_for(total, 2);
    if (src_dtype == UIG_F32 && dtype == UIG_F32) hipLaunchKernelGGL((to_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, sb, sc, sh, sw, (float*)dst, B, C, H, W, Cp);
    else if (src_dtype == UIG_F32 && dtype == UIG_BF16) hipLaunchKernelGGL((to_nhwc_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, sb, sc, sh, sw, (bf16_t*)dst, B, C, H, W, Cp);
    else if (src_dtype == UIG_BF16 && dtype == UIG_F32) hipLaunchKernelGGL((to_nhwc_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, sb, sc, sh, sw, (float*)dst, B, C, H, W, Cp);
    else if (src_dtype == UIG_BF16 && dtype == UIG_BF16) hipLaunchKernelGGL((to_nhwc_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, sb, sc, sh, sw, (bf16_t*)dst, B, C, H, W, Cp);
    else return uig_set_error(-1, "uig_to_nhwc: bad dtypes %d -> %d", src_dtype, dtype);
    UIG_LAUNCH_CHECK("uig_to_nhwc");
    return 0;
}
extern "C" int uig_from_nhwc(const void* src, int B, int C, int H, int W, int Cp, int dtype,
                             void* dst, int dst_dtype, int64_t sb, int64_t sc, int64_t sh, int64_t sw, void* stream) {
    UIG_CHECK_ARG(src && dst && Cp >= C, "uig_from_nhwc: bad args (C=%d Cp=%d)", C, Cp);
    const long total = (long)B * C * H * W;
    hipStream_t s = (hipStream_t)stream; const int g = grid_for(total, 2);
    if (dtype == UIG_F32 && dst_dtype == UIG_F32) hipLaunchKernelGGL((from_nhwc_kernel<float, float>), dim3(g), dim3(256), 0, s, (const float*)src, (float*)dst, sb, sc, sh, sw, B, C, H, W, Cp);
    else if (dtype == UIG_BF16 && dst_dtype == UIG_F32) hipLaunchKernelGGL((from_nhwc_kernel<bf16_t, float>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (float*)dst, sb, sc, sh, sw, B, C, H, W, Cp);
    else if (dtype == UIG_F32 && dst_dtype == UIG_BF16) hipLaunchKernelGGL((from_nhwc_kernel<float, bf16_t>), dim3(g), dim3(256), 0, s, (const float*)src, (bf16_t*)dst, sb, sc, sh, sw, B, C, H, W, Cp);
    else if (dtype == UIG_BF16 && dst_dtype == UIG_BF16) hipLaunchKernelGGL((from_nhwc_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, s, (const bf16_t*)src, (bf16_t*)dst, sb, sc, sh, sw, B, C, H, W, Cp);
    else return uig_set_error(-1, "uig_from_nhwc: bad dtypes %d -> %d", dtype, dst_dtype);
    UIG_LAUNCH_CHECK("uig_from_nhwc");
    return 0;
}
