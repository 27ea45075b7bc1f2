// instnorm.hip — InstanceNorm2d (no affine, biased variance) fused with ReLU / LeakyReLU / residual add, fwd + bwd,
// and the per-channel column sum used for conv bias gradients.  NHWC, HBM-bound: every lane moves 16-byte chunks
// (8 bf16 / 4 f32 channels), a thread keeps the same channel chunk for all its pixels, per-thread fp32 partial sums are
// combined across the block's pixel lanes through LDS, written as per-slab partials and finished in fp64 by a tiny
// finalize kernel (deterministic: no float atomics).
//   aten::instance_norm(use_input_stats=True, weight=None)   /   aten::native_batch_norm_backward   
#include "uig_common.h"
#include <algorithm>

// MODE 0: (sum x, sum x^2)      MODE 1: (sum g, sum g*xhat) with g = dy * act'(xhat)
template <typename T, int MODE>
__global__ __launch_bounds__(256) void in_stats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const float* __restrict__ stats, float* __restrict__ partial,
                                                        long HW, int C, int CC, int nslab, int act, float slope, const UigFin fin = UigFin{}) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float red[256 * E * 2];
    const int tid = threadIdx.x;
    const int PL = 256 / CC, pl = tid / CC, cc = tid % CC;
    const int b = blockIdx.y, slab = blockIdx.x;
    const long sp = (HW + nslab - 1) / nslab;
    const long p0 = slab * sp, p1 = min(HW, p0 + sp);
    const T* xb = x + (long)b * HW * C + cc * E;
    float s1[E], s2[E], mu[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { s1[e] = 0.f; s2[e] = 0.f; mu[e] = 0.f; rs[e] = 1.f; }
    if constexpr (MODE == 1) {
#pragma unroll
        for (int e = 0; e < E; ++e) { mu[e] = stats[((long)b * C + cc * E + e) * 2]; rs[e] = stats[((long)b * C + cc * E + e) * 2 + 1]; }
    }
    const T* dyb = (MODE == 1) ? dy + (long)b * HW * C + cc * E : nullptr;
    for (long p = p0 + pl; p < p1; p += PL) {
        float xv[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(xb + p * C), xv);
        if constexpr (MODE == 0) {
#pragma unroll
            for (int e = 0; e < E; ++e) { s1[e] += xv[e]; s2[e] += xv[e] * xv[e]; }
        } else {
            float gv[E];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(dyb + p * C), gv);
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                float g = gv[e];
                if (act == UIG_ACT_RELU) g = xh > 0.f ? g : 0.f;
                else if (act == UIG_ACT_LRELU) g = xh > 0.f ? g : g * slope;
                s1[e] += g; s2[e] += g * xh;
            }
        }
    }
#pragma unroll
    for (int e = 0; e < E; ++e) { red[(tid * E + e) * 2] = s1[e]; red[(tid * E + e) * 2 + 1] = s2[e]; }
    __syncthreads();
    // channel c = cc*E + e lives at red[((pl*CC + cc)*E + e)*2] = red[(pl*C + c)*2]
    for (int c = tid; c < C; c += 256) {
        float a = 0.f, q = 0.f;
        for (int l = 0; l < PL; ++l) { a += red[(l * C + c) * 2]; q += red[(l * C + c) * 2 + 1]; }
        float* out = partial + (((long)b * nslab + slab) * C + c) * 2;
        if (fin.tickets != nullptr) uig_store8_sc1(out, a, q);      // block-uniform: read back inside this launch by the image's last arriver
        else { out[0] = a; out[1] = q; }
    }
    // round 4: no finalize launch - the image's last-arriving block finalises (uig_common.h, UigFin); red[] is free behind the barrier
    if (fin.tickets != nullptr) uig_fin_arrive<256>(fin, b, 1u, reinterpret_cast<unsigned*>(red), tid);
}

// fin MODE 0: stats = (mean, rstd)   MODE 1: out = (mean_g, mean_gxhat)   MODE 2: db[c] (+)= sum (B folded into slabs)
// 256 threads = 16 (b,c) items x 16 slab lanes; fp64 combine across the slab lanes by wave shuffles (lanes of one item are
// 16 consecutive lanes of a wave).
__global__ __launch_bounds__(256) void in_finalize_kernel(const float* __restrict__ partial, float* __restrict__ out, int BC,
                                                           int C, int nslab, double inv_n, float eps, int mode, int nreal,
                                                           int accumulate) {
    const int sl = threadIdx.x & 15, it = threadIdx.x >> 4;
    const int i = blockIdx.x * 16 + it;
    double a = 0.0, q = 0.0;
    int b = 0, c = 0;
    if (i < BC) {
        b = i / C; c = i % C;
        for (int s = sl; s < nslab; s += 16) {
            const float* p = partial + (((long)b * nslab + s) * C + c) * 2;
            a += (double)p[0]; q += (double)p[1];
        }
    }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); q += __shfl_xor(q, o, 16); }
    if (sl != 0 || i >= BC) return;
    if (mode == 0) {
        const double mean = a * inv_n;
        double var = q * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        out[(long)i * 2] = (float)mean;
        out[(long)i * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    } else if (mode == 1) {
        out[(long)i * 2] = (float)(a * inv_n);
        out[(long)i * 2 + 1] = (float)(q * inv_n);
    } else {
        if (c < nreal) out[c] = accumulate ? out[c] + (float)a : (float)a;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void in_apply_fwd_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                            T* __restrict__ y, const float* __restrict__ stats,
                                                            long HW, int C, int CC, int nslab, int act, float slope,
                                                            unsigned char* __restrict__ mxq = nullptr, unsigned char* __restrict__ mxs = nullptr) {
    constexpr int E = ElemTraits<T>::E;
    const int tid = threadIdx.x;
    const int PL = 256 / CC, pl = tid / CC, cc = tid % CC;
    const int b = blockIdx.y;
    const long sp = (HW + nslab - 1) / nslab;
    const long p0 = blockIdx.x * sp, p1 = min(HW, p0 + sp);
    const long base = (long)b * HW * C + cc * E;
    float mu[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = stats[((long)b * C + cc * E + e) * 2]; rs[e] = stats[((long)b * C + cc * E + e) * 2 + 1]; }
    for (long p = p0 + pl; p < p1; p += PL) {
        float v[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(x + base + p * C), v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = apply_act((v[e] - mu[e]) * rs[e], act, slope);
        if (res != nullptr) {
            float r[E];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(res + base + p * C), r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        const u32x4_t packed = f32_to_chunk<T>(v);
        *reinterpret_cast<u32x4_t*>(y + base + p * C) = packed;
        if constexpr (E == 8) {
            if (mxq != nullptr) {      // also the MX fp8 form of the stored (bf16-rounded) values for the fp8 convolution that consumes them
                float rv[E];
                chunk_to_f32<T>(packed, rv);
                int sb;
                const u32x2_t w = mx_quantize8(rv, sb);
                const long pix = (long)b * HW + p;
                *reinterpret_cast<u32x2_t*>(mxq + pix * C + cc * 8) = w;
                if ((cc & 3) == 0) mxs[pix * (C / 32) + (cc >> 2)] = (unsigned char)sb;
            }
        }
    }
}

// Inference form (InstanceNorm fused for latency): the apply kernel FINALISES the statistics itself,
// so an InstanceNorm behind a convolution that emitted the partials in its epilogue is ONE launch instead of two (finalize +
// apply), and no (mean, rstd) tensor is written: nothing is kept for a backward pass.  Phase 1: thread c of the block
// reduces channel c's `np` partial (sum, sum^2) pairs in fp64 - in exactly the association order of in_finalize_kernel (16
// strided accumulators, then the xor-shuffle tree), so the result is bit-identical to the three-launch path - and leaves
// (mean, rstd) in LDS; phase 2 is in_apply_fwd_kernel.  Each block re-reads the image's partials (np * C * 8 bytes, L2
// resident): that is what makes this slower than the separate finalize launch at training batch sizes (measured in round 1:
// 21.7 vs 16.4 us at batch 16) and faster on the batch-1 dependent chain of a single image, where every launch boundary counts.
template <typename T>
__global__ __launch_bounds__(256) void in_apply_fwd_fin_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                               const float* __restrict__ partial, int np, double inv_n, float eps,
                                                               long HW, int C, int CC, int nslab, int act, float slope) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float st[2048 * 2];                        // C <= 2048 (CC <= 256)
    const int tid = threadIdx.x, b = blockIdx.y;
    for (int c = tid; c < C; c += 256) {
        double a16[16], q16[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { a16[i] = 0.0; q16[i] = 0.0; }
        for (int s0 = 0; s0 < np; s0 += 16) {
            // all 16 loads of the round are issued unconditionally (clamped index) and selected afterwards: a load under a
            // run-time condition makes the compiler branch around each one and wait for it alone - 16 dependent L2 round trips
            // per round (guide: "register or load" trap; measured here: +7 us per launch)
            u32x2_t v[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)
                v[i] = *reinterpret_cast<const u32x2_t*>(partial + (((long)b * np + min(s0 + i, np - 1)) * C + c) * 2);
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const bool ok = s0 + i < np;
                a16[i] += ok ? (double)__uint_as_float(v[i][0]) : 0.0; q16[i] += ok ? (double)__uint_as_float(v[i][1]) : 0.0;
            }
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & o) == 0 && i < o) { a16[i] += a16[i ^ o]; q16[i] += q16[i ^ o]; }
        const double mean = a16[0] * inv_n;
        double var = q16[0] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        st[2 * c] = (float)mean; st[2 * c + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int PL = 256 / CC, pl = tid / CC, cc = tid % CC;
    const long sp = (HW + nslab - 1) / nslab;
    const long p0 = blockIdx.x * sp, p1 = min(HW, p0 + sp);
    const long base = (long)b * HW * C + cc * E;
    float mu[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = st[2 * (cc * E + e)]; rs[e] = st[2 * (cc * E + e) + 1]; }
    for (long p = p0 + pl; p < p1; p += PL) {
        float v[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(x + base + p * C), v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = apply_act((v[e] - mu[e]) * rs[e], act, slope);
        if (res != nullptr) {
            float r[E];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(res + base + p * C), r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        *reinterpret_cast<u32x4_t*>(y + base + p * C) = f32_to_chunk<T>(v);
    }
}

// Round 3, channel-sliced form of the kernel above for small batches: a block owns 64 CHANNELS of a pixel range, so its finalize
// prologue reads np * 64 * 8 bytes (32 KB at 64 partials) instead of np * C * 8 - at batch 1 the 23 finalize launches of a generator
// call were 20 % of its latency (7.4 us each in the kernel trace) and the whole-C prologue cost more than it saved.  The 16
// accumulators of a channel (the finalize kernel's 16 slab lanes) are spread over 4 threads (one load round), exchanged through LDS
// and combined by the channel's thread in exactly in_finalize_kernel's order: bit-identical (mean, rstd).
template <typename T>
__global__ __launch_bounds__(256) void in_apply_fwd_fin_cs_kernel(const T* __restrict__ x, const T* __restrict__ res, T* __restrict__ y,
                                                                  const float* __restrict__ partial, int np, double inv_n, float eps,
                                                                  long HW, int C, int npx, int act, float slope) {
    constexpr int E = ElemTraits<T>::E, CG = 64, CCG = CG / E, PLG = 256 / CCG;
    __shared__ double acc_a[16][CG], acc_q[16][CG];
    __shared__ float st[CG * 2];
    const int tid = threadIdx.x, b = blockIdx.z, cg = blockIdx.y;
    {
        const int c = tid & (CG - 1), qd = tid >> 6;                 // channel of the group, quarter of its 16 accumulators
        const float* pb = partial + ((long)b * np * C + cg * CG + c) * 2;
        double a4[4] = {0.0, 0.0, 0.0, 0.0}, q4[4] = {0.0, 0.0, 0.0, 0.0};
        for (int s0 = 0; s0 < np; s0 += 16) {
            u32x2_t v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const u32x2_t*>(pb + (long)min(s0 + 4 * qd + i, np - 1) * C * 2);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const bool ok = s0 + 4 * qd + i < np;
                a4[i] += ok ? (double)__uint_as_float(v[i][0]) : 0.0; q4[i] += ok ? (double)__uint_as_float(v[i][1]) : 0.0;
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) { acc_a[4 * qd + i][c] = a4[i]; acc_q[4 * qd + i][c] = q4[i]; }
    }
    __syncthreads();
    if (tid < CG) {
        double a16[16], q16[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) { a16[i] = acc_a[i][tid]; q16[i] = acc_q[i][tid]; }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int i = 0; i < 16; ++i)
                if ((i & o) == 0 && i < o) { a16[i] += a16[i ^ o]; q16[i] += q16[i ^ o]; }
        const double mean = a16[0] * inv_n;
        double var = q16[0] * inv_n - mean * mean;
        if (var < 0.0) var = 0.0;
        st[2 * tid] = (float)mean; st[2 * tid + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
    __syncthreads();
    const int pl = tid / CCG, cc = tid % CCG;
    const long sp = (HW + npx - 1) / npx;
    const long p0 = blockIdx.x * sp, p1 = min(HW, p0 + sp);
    const long base = (long)b * HW * C + cg * CG + cc * E;
    float mu[E], rs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) { mu[e] = st[2 * (cc * E + e)]; rs[e] = st[2 * (cc * E + e) + 1]; }
    for (long p = p0 + pl; p < p1; p += PLG) {
        float v[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(x + base + p * C), v);
#pragma unroll
        for (int e = 0; e < E; ++e) v[e] = apply_act((v[e] - mu[e]) * rs[e], act, slope);
        if (res != nullptr) {
            float r[E];
            chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(res + base + p * C), r);
#pragma unroll
            for (int e = 0; e < E; ++e) v[e] += r[e];
        }
        *reinterpret_cast<u32x4_t*>(y + base + p * C) = f32_to_chunk<T>(v);
    }
}

// COLSUM: additionally emit per-block partial sums of the written dx (as stored, i.e. after rounding to T) per channel:
// the bias gradient of the convolution in front of this InstanceNorm is the column sum of exactly this tensor, so the
// separate full read pass of uig_bias_grad disappears (it sat on the backward critical path).
template <typename T, bool COLSUM>
__global__ __launch_bounds__(256) void in_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ x,
                                                            T* __restrict__ dx, const float* __restrict__ stats,
                                                            const float* __restrict__ gm, float* __restrict__ colsum_partial,
                                                            long HW, int C, int CC, int nslab, int act, float slope,
                                                            unsigned char* __restrict__ mxq = nullptr, unsigned char* __restrict__ mxs = nullptr) {
    constexpr int E = ElemTraits<T>::E;
    __shared__ float red[COLSUM ? 256 * E : 1];
    const int tid = threadIdx.x;
    const int PL = 256 / CC, pl = tid / CC, cc = tid % CC;
    const int b = blockIdx.y;
    const long sp = (HW + nslab - 1) / nslab;
    const long p0 = blockIdx.x * sp, p1 = min(HW, p0 + sp);
    const long base = (long)b * HW * C + cc * E;
    float mu[E], rs[E], mg[E], mgx[E], cs[E];
#pragma unroll
    for (int e = 0; e < E; ++e) {
        const long i = ((long)b * C + cc * E + e) * 2;
        mu[e] = stats[i]; rs[e] = stats[i + 1]; mg[e] = gm[i]; mgx[e] = gm[i + 1]; cs[e] = 0.f;
    }
    for (long p = p0 + pl; p < p1; p += PL) {
        float xv[E], gv[E];
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(x + base + p * C), xv);
        chunk_to_f32<T>(*reinterpret_cast<const u32x4_t*>(dy + base + p * C), gv);
#pragma unroll
        for (int e = 0; e < E; ++e) {
            const float xh = (xv[e] - mu[e]) * rs[e];
            float g = gv[e];
            if (act == UIG_ACT_RELU) g = xh > 0.f ? g : 0.f;
            else if (act == UIG_ACT_LRELU) g = xh > 0.f ? g : g * slope;
            gv[e] = rs[e] * (g - mg[e] - xh * mgx[e]);
        }
        const u32x4_t packed = f32_to_chunk<T>(gv);
        *reinterpret_cast<u32x4_t*>(dx + base + p * C) = packed;
        if constexpr (COLSUM) {
            float rv[E];
            chunk_to_f32<T>(packed, rv);
#pragma unroll
            for (int e = 0; e < E; ++e) cs[e] += rv[e];
        }
        if constexpr (E == 8) {
            if (mxq != nullptr) {      // MX fp8 form of the stored dx: the input of the fp8 input-gradient convolution in front
                float rv[E];
                chunk_to_f32<T>(packed, rv);
                int sb;
                const u32x2_t w = mx_quantize8(rv, sb);
                const long pix = (long)b * HW + p;
                *reinterpret_cast<u32x2_t*>(mxq + pix * C + cc * 8) = w;
                if ((cc & 3) == 0) mxs[pix * (C / 32) + (cc >> 2)] = (unsigned char)sb;
            }
        }
    }
    if constexpr (COLSUM) {
#pragma unroll
        for (int e = 0; e < E; ++e) red[tid * E + e] = cs[e];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float a = 0.f;
            for (int l = 0; l < PL; ++l) a += red[l * C + c];
            float* out = colsum_partial + (((long)b * nslab + blockIdx.x) * C + c) * 2;
            out[0] = a; out[1] = 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------ fused backward (round 4)
// InstanceNorm backward in ONE launch and ONE pass over the data: statistics pass, finalize launch and apply pass (40 us for the
// ResBlock maps of the benchmark: 13.4 + 5.3 + 21.5, reading dy and x twice) become a kernel in which every thread loads its pixels of
// dy and x ONCE (8 pixels x 16 bytes x 2 tensors, all 16 loads in flight together), keeps them in registers across two in-kernel
// synchronisations of the blocks that share an image, and writes dx: 67 MB read + 33 MB written per 16-image launch instead of 134 + 33.
//   phase 1  per-thread (sum g, sum g*xhat) over its pixels, block reduce through LDS (in_stats_kernel's order), the block's partial
//            [C][2] stored write-through (sc1);
//   sync 1   arrival counter of the image (one relaxed agent-scope fetch_add per block, one lane polls with sc1 loads + s_sleep);
//   finalize DISTRIBUTED: block s of the image's NB blocks reduces the NB partials of channels [s * cpb, (s + 1) * cpb) in fp64
//            (16 slab lanes per channel, sequential sums, xor-shuffle tree: the finalize kernel's association) and stores
//            (mean g, mean g*xhat) write-through - 2 KB read per block, not NB * C * 8;
//   sync 2   second arrival counter; then every thread reads the 8 (mean g, mean g*xhat) pairs of its channels (sc1 loads);
//   phase 2  dx = rstd * (g - mean g - xhat * mean g*xhat) from the registers, stored; column sums of the stored dx (COLSUM) and the
//            MX fp8 form of dx as in in_apply_bwd_kernel; the block that DEPARTS last resets the image's three counters.
// Residency: blocks of one image wait for each other, so the whole grid must be co-resident: the host admits the launch only when
// grid <= CUs x (blocks per CU of this kernel) (instnorm_bwd_fused_plan) and images are laid out so that consecutive block ids cover
// whole images eight at a time (image = xcd + 8 * (j / NB): an image's blocks share an XCD where dispatch is round-robin - speed only).
// Every spin is BOUNDED: a block that waits longer than spin_limit polls sets *err and goes on with whatever it has (wrong numbers,
// no hang); the host checks the word when it next synchronises (ops.check_sync_errors) and switches the fused path off.
// Hand-off protocol as uig_common.h's UigFin (cdna_hip_programming.md Guideline 16, sc1 form): every handed-off byte is stored sc1
// and drained before the counter add that publishes it; every load of it is an sc1 load.
struct InFusedDesc {
    long HW; int C, CC, NB, B, act; float slope;
    float* partial;           // [B][NB][C][2]
    float* gm;                // [B][C][2]
    unsigned* sync;           // [B][4]: arrive 1, arrive 2, depart, -   (zero between launches)
    unsigned* err;            // set to 1 when a bounded spin ran out
    double inv_n; unsigned spin_limit; int cpb;
};

__device__ __forceinline__ bool in_fused_sync(unsigned* cnt, unsigned target, unsigned* err, unsigned limit, unsigned* lds_word, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's sc1 stores are through
    __syncthreads();
    if (tid == 0) {
        __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned ok = 1u, spins = 0u;
        while (__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(4);
            if (++spins > limit) { ok = 0u; __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); break; }
        }
        *reinterpret_cast<volatile unsigned*>(lds_word) = ok;
    }
    __syncthreads();
    return *reinterpret_cast<volatile unsigned*>(lds_word) != 0u;
}

template <typename T, bool COLSUM, int ACT>
__device__ __forceinline__ void in_bwd_fused_body(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                                  const float* __restrict__ stats, float* __restrict__ colsum_partial,
                                                  const InFusedDesc& d, float* red) {
    constexpr int E = ElemTraits<T>::E, PPT = 8;
    const int tid = threadIdx.x;
    const int pb = blockIdx.x, xcd = pb & 7, jb = pb >> 3;
    const int b = xcd + 8 * (jb / d.NB), slab = jb % d.NB;
    if (b >= d.B) return;                                  // block-uniform: images past the batch (grid rounded up to eight images)
    const int C = d.C, CC = d.CC, PL = 256 / CC, pl = tid / CC, cc = tid % CC;
    const long sp = (d.HW + d.NB - 1) / d.NB, p0 = slab * sp, p1 = min(d.HW, p0 + sp);
    const unsigned img_bytes = (unsigned)(d.HW * C * (long)sizeof(T));
    const __amdgpu_buffer_rsrc_t rx = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(x + (long)b * d.HW * C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rg = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(dy + (long)b * d.HW * C), 0, img_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dx + (long)b * d.HW * C, 0, img_bytes, 0x00020000);
    // byte offset of this thread's pixel k inside its image, or out of the buffer's range (loads return zeros: g = 0; stores are dropped).
    // Derived from ONE register (off0) that is made opaque again in front of phase 2: the eight offsets are recomputed there, not carried
    // across the waits
    int off0 = (int)(((p0 + pl) * C + cc * E) * (long)sizeof(T));
    const int kstride = PL * C * (int)sizeof(T);
    const int nvalid = (int)((p1 - p0 - pl + PL - 1) / PL);          // pixels k < nvalid of this thread lie inside the block's range (may be <= 0)
    auto poff = [&](int k) -> int { return k < nvalid ? off0 + k * kstride : -1; };
    // ---- every load of this thread, issued together
    u32x4_t xr[PPT], gr[PPT];
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int off = poff(k);
        xr[k] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rx, off, 0, 0));
        gr[k] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rg, off, 0, 0));
    }
    const float slope = d.slope;
    auto gval = [&](float g, float xh) -> float {
        if constexpr (ACT == UIG_ACT_RELU) return xh > 0.f ? g : 0.f;
        else if constexpr (ACT == UIG_ACT_LRELU) return xh > 0.f ? g : g * slope;
        else return g;
    };
    const float* stp = stats + ((long)b * C + cc * E) * 2;
    constexpr int EH = E / 2;
    auto unpack_half = [&](const u32x4_t& w, int h, float (&f)[EH]) {
        if constexpr (sizeof(T) == 2) {
#pragma unroll
            for (int i2 = 0; i2 < EH / 2; ++i2) { const unsigned v = w[h * (EH / 2) + i2]; f[2 * i2] = __uint_as_float(v << 16); f[2 * i2 + 1] = __uint_as_float(v & 0xffff0000u); }
        } else {
#pragma unroll
            for (int e = 0; e < EH; ++e) f[e] = __uint_as_float(w[h * EH + e]);
        }
    };
    // ---- phase 1: partial statistics of the block's pixels (one half of the thread's channels at a time: registers)
    {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            float mu[EH], rs[EH], s1[EH], s2[EH];
#pragma unroll
            for (int e = 0; e < EH; ++e) { mu[e] = stp[2 * (h * EH + e)]; rs[e] = stp[2 * (h * EH + e) + 1]; s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                float xv[EH], gv[EH];
                unpack_half(xr[k], h, xv); unpack_half(gr[k], h, gv);
#pragma unroll
                for (int e = 0; e < EH; ++e) {
                    const float xh = (xv[e] - mu[e]) * rs[e];
                    const float g = gval(gv[e], xh);
                    s1[e] += g; s2[e] += g * xh;
                }
            }
#pragma unroll
            for (int e = 0; e < EH; ++e) { red[(tid * E + h * EH + e) * 2] = s1[e]; red[(tid * E + h * EH + e) * 2 + 1] = s2[e]; }
        }
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float a = 0.f, q = 0.f;
            for (int l = 0; l < PL; ++l) { a += red[(l * C + c) * 2]; q += red[(l * C + c) * 2 + 1]; }
            uig_store8_sc1(d.partial + (((long)b * d.NB + slab) * C + c) * 2, a, q);
        }
    }
    // the packed pixels stay in registers across the waits; opaque from here on: nothing derived from them in phase 1 (the unpacked
    // values, xhat, g: 128 floats) may be carried over - phase 2 recomputes from the 64 packed registers
#pragma unroll
    for (int k = 0; k < PPT; ++k) asm volatile("" : "+v"(xr[k]), "+v"(gr[k]));
    asm volatile("" : "+v"(off0));
    unsigned* sy = d.sync + (long)b * 4;
    in_fused_sync(sy + 0, (unsigned)d.NB, d.err, d.spin_limit, reinterpret_cast<unsigned*>(red), tid);
    // ---- distributed finalize: this block's share of the channels
    {
        const int sl = tid & 15, it = tid >> 4;
        const int c_begin = slab * d.cpb, c_end = min(C, c_begin + d.cpb);
        const float* pbase = d.partial + (long)b * d.NB * C * 2;
        for (int c0 = c_begin; c0 < c_end; c0 += 16) {      // block-uniform trip count
            const int c = min(c0 + it, c_end - 1);
            double a = 0.0, q = 0.0;
            for (int s_ = sl; s_ < d.NB; s_ += 16) {
                const unsigned long long v = uig_load8_sc1(pbase + ((long)s_ * C + c) * 2);
                a += (double)__uint_as_float((unsigned)v); q += (double)__uint_as_float((unsigned)(v >> 32));
            }
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) { a += __shfl_xor(a, o, 16); q += __shfl_xor(q, o, 16); }
            if (sl == 0 && c0 + it < c_end) uig_store8_sc1(d.gm + ((long)b * C + c) * 2, (float)(a * d.inv_n), (float)(q * d.inv_n));
        }
    }
    in_fused_sync(sy + 1, (unsigned)d.NB, d.err, d.spin_limit, reinterpret_cast<unsigned*>(red), tid);
    // ---- phase 2: apply from the registers, one HALF of the thread's channels at a time (its (mean, rstd, mean g, mean g*xhat) then take
    //      16 registers instead of 32: with all eight channels the kernel spilled ~30 registers per thread - scratch traffic as large as
    //      the data - and measured 96 us against 36 for the three launches); each half is one 8-byte (bf16) / 8-byte (f32: two channels) store
    float cs[E];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float mu[EH], rs[EH], mg[EH], mgx[EH];
#pragma unroll
        for (int e = 0; e < EH; ++e) {
            const unsigned long long v = uig_load8_sc1(d.gm + ((long)b * C + cc * E + h * EH + e) * 2);
            mg[e] = __uint_as_float((unsigned)v); mgx[e] = __uint_as_float((unsigned)(v >> 32)); cs[h * EH + e] = 0.f;
            mu[e] = stp[2 * (h * EH + e)]; rs[e] = stp[2 * (h * EH + e) + 1];
        }
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            const int off = poff(k);
            float xv[EH], gv[EH];
            unpack_half(xr[k], h, xv); unpack_half(gr[k], h, gv);
#pragma unroll
            for (int e = 0; e < EH; ++e) {
                const float xh = (xv[e] - mu[e]) * rs[e];
                gv[e] = rs[e] * (gval(gv[e], xh) - mg[e] - xh * mgx[e]);
            }
            u32x2_t pk;
            float rv[EH];
            if constexpr (sizeof(T) == 2) {
                pk[0] = (unsigned)f32_to_bf16(gv[0]) | ((unsigned)f32_to_bf16(gv[1]) << 16);
                pk[1] = (unsigned)f32_to_bf16(gv[2]) | ((unsigned)f32_to_bf16(gv[3]) << 16);
                rv[0] = __uint_as_float(pk[0] << 16); rv[1] = __uint_as_float(pk[0] & 0xffff0000u);
                rv[2] = __uint_as_float(pk[1] << 16); rv[3] = __uint_as_float(pk[1] & 0xffff0000u);
            } else {
                pk[0] = __float_as_uint(gv[0]); pk[1] = __float_as_uint(gv[1]);
                rv[0] = gv[0]; rv[1] = gv[1];
            }
            __builtin_amdgcn_raw_buffer_store_b64(pk, rd, off >= 0 ? off + h * 8 : -1, 0, 0);      // out-of-range offset: dropped
            if constexpr (COLSUM) {
#pragma unroll
                for (int e = 0; e < EH; ++e) cs[h * EH + e] += off >= 0 ? rv[e] : 0.f;
            }
        }
    }
    if constexpr (COLSUM) {
        __syncthreads();                                   // red[] carried the sync flag
#pragma unroll
        for (int e = 0; e < E; ++e) red[tid * E + e] = cs[e];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float a = 0.f;
            for (int l = 0; l < PL; ++l) a += red[l * C + c];
            float* out = colsum_partial + (((long)b * d.NB + slab) * C + c) * 2;
            out[0] = a; out[1] = 0.f;
        }
    }
    // ---- depart: the image's last block leaves its counters zero for the next launch
    if (tid == 0) {
        const unsigned dcount = __hip_atomic_fetch_add(sy + 2, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (dcount == (unsigned)d.NB - 1u) {
            __hip_atomic_store(sy + 0, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sy + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(sy + 2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

template <typename T, bool COLSUM>
__global__ __launch_bounds__(256, 4) void in_bwd_fused_kernel(const T* __restrict__ dy, const T* __restrict__ x, T* __restrict__ dx,
                                                               const float* __restrict__ stats, float* __restrict__ colsum_partial,
                                                               const InFusedDesc d) {
    __shared__ float red[256 * ElemTraits<T>::E * 2];
    // the activation is a launch constant: ONE branch around the whole body
    if (d.act == UIG_ACT_RELU) in_bwd_fused_body<T, COLSUM, UIG_ACT_RELU>(dy, x, dx, stats, colsum_partial, d, red);
    else if (d.act == UIG_ACT_LRELU) in_bwd_fused_body<T, COLSUM, UIG_ACT_LRELU>(dy, x, dx, stats, colsum_partial, d, red);
    else in_bwd_fused_body<T, COLSUM, UIG_ACT_NONE>(dy, x, dx, stats, colsum_partial, d, red);
}

// ------------------------------------------------------------------------------------------------ host side
static int stats_slabs(long HW, int CC) { return (int)std::max<long>(1, std::min<long>(128, HW * CC / (256 * 4))); }
static int apply_slabs(long HW, int CC) { return (int)std::max<long>(1, std::min<long>(2048, HW * CC / (256 * 4))); }

static int check_in_args(const char* fn, int B, long HW, int C, int dtype, int* CC) {
    UIG_CHECK_ARG(dtype == UIG_F32 || dtype == UIG_BF16, "%s: bad dtype %d", fn, dtype);
    const int E = dtype == UIG_BF16 ? 8 : 4;
    UIG_CHECK_ARG(B > 0 && HW > 0 && C > 0 && C % E == 0, "%s: bad shape B=%d HW=%ld C=%d", fn, B, HW, C);
    *CC = C / E;
    UIG_CHECK_ARG(*CC <= 256 && (256 % *CC) == 0, "%s: C=%d unsupported (C/%d must divide 256)", fn, C, E);
    UIG_CHECK_ARG(B <= 65535, "%s: B too large", fn);
    return 0;
}

extern "C" size_t uig_instnorm_workspace_floats(int B, int64_t HW, int C) {
    (void)HW;
    return (size_t)B * 128 * C * 2 + (size_t)B * C * 2;
}
extern "C" size_t uig_colsum_workspace_floats(int C) { return (size_t)128 * C * 2; }

extern "C" int uig_instnorm_act_fwd(const void* x, const void* residual, void* y, float* stats, float* workspace,
                                    int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && stats && workspace, "uig_instnorm_act_fwd: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_fwd: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_fwd", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int ns = stats_slabs(HW, CC), na = apply_slabs(HW, CC);
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_stats_kernel<bf16_t, 0>), dim3(ns, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, ns, 0, 0.f);
    else
        hipLaunchKernelGGL((in_stats_kernel<float, 0>), dim3(ns, B), dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, ns, 0, 0.f);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd(stats)");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, s, workspace, stats, B * C, C, ns, 1.0 / (double)HW, eps, 0, 0, 0);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd(finalize)");
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_apply_fwd_kernel<bf16_t>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, stats, (long)HW, C, CC, na, act, slope);
    else
        hipLaunchKernelGGL((in_apply_fwd_kernel<float>), dim3(na, B), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, stats, (long)HW, C, CC, na, act, slope);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd(apply)");
    return 0;
}

// UigFin of a stand-alone statistics pass with ns slabs per image (one arrival per block)
static UigFin make_fin(float* out, unsigned* tickets, const float* partial, int ns, int C, long HW, float eps, int mode) {
    UigFin f{};
    f.out = out; f.tickets = tickets; f.partial = partial; f.nslab = ns; f.C = C; f.expected = (unsigned)ns; f.mode = mode | uig_in_tickets_dbg(); f.eps = eps;
    f.inv_n = 1.0 / (double)HW;
    return f;
}
static int g_in_tickets = 1;    // A/B and parity hook: 0 = the ticketed entry points run the finalize LAUNCH instead (bit-identical results)
extern "C" void uig_debug_set_in_tickets(int on) { g_in_tickets = on; }
bool uig_in_tickets_on() { return g_in_tickets != 0; }
int uig_in_tickets_dbg() { return g_in_tickets == 2 ? 16 : (g_in_tickets == 3 ? 32 + 16 : 0); }      // 2: arrive, skip the reduction; 3: no ticket either (timing only)

// pixel slabs (= blocks) per image of the backward apply pass that also emits the column sums.  Round 4 A/B in the full step (one box,
// two rounds): 32 -> 13.53-13.56 ms, 64 -> 13.56, 128 -> 14.04-14.09, 256 -> 14.17: more slabs buy no bandwidth and cost the
// column-sum partials' reduce (bias rider of the weight-gradient reduce).  32 stays.
static int g_colsum_slabs = 32;
extern "C" void uig_debug_set_colsum_slabs(int n) { g_colsum_slabs = n > 0 ? n : 32; }
static int colsum_slabs(long HW, int CC) { return (int)std::max<long>(1, std::min<long>(g_colsum_slabs, HW * CC / (256 * 4))); }
extern "C" int uig_instnorm_bwd_colsum_slabs(int B, int64_t HW, int C, int dtype) {
    const int E = dtype == UIG_BF16 ? 8 : 4;
    return B * colsum_slabs(HW, C / E);
}

static int instnorm_bwd_impl(const void* dy, const void* x, const float* stats, void* dx, float* workspace, float* colsum_partial,
                             int B, int64_t HW, int C, int act, float slope, int dtype, void* stream,
                             unsigned char* mxq = nullptr, unsigned char* mxs = nullptr, const float* pre_partial = nullptr, int pre_nslab = 0,
                             unsigned* tickets = nullptr, const float* pre_gm = nullptr) {
    UIG_CHECK_ARG(dy && x && stats && dx && workspace, "uig_instnorm_act_bwd: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_bwd: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_bwd", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int ns = stats_slabs(HW, CC), na = colsum_partial ? colsum_slabs(HW, CC) : apply_slabs(HW, CC);
    const float* gm = pre_gm;      // (mean g, mean g*xhat) already finalised inside the launch that wrote dy (round 4)
    if (gm == nullptr) {
        float* gmw = workspace + (size_t)B * 128 * C * 2;
        gm = gmw;
        const bool tk = tickets != nullptr && g_in_tickets && pre_partial == nullptr;      // the statistics pass finalises itself: no finalize launch
        const UigFin fin = tk ? make_fin(gmw, tickets, workspace, ns, C, (long)HW, 0.f, 1) : UigFin{};
        if (pre_partial == nullptr) {
            if (dtype == UIG_BF16)
                hipLaunchKernelGGL((in_stats_kernel<bf16_t, 1>), dim3(ns, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)dy, stats, workspace, (long)HW, C, CC, ns, act, slope, fin);
            else
                hipLaunchKernelGGL((in_stats_kernel<float, 1>), dim3(ns, B), dim3(256), 0, s, (const float*)x, (const float*)dy, stats, workspace, (long)HW, C, CC, ns, act, slope, fin);
            UIG_LAUNCH_CHECK("uig_instnorm_act_bwd(stats)");
        }
        if (!tk) {
            // (sum g, sum g*xhat) partials: this norm's own statistics pass, or the epilogue of the launch that wrote dy
            hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, s, pre_partial ? pre_partial : workspace, gmw, B * C, C,
                               pre_partial ? pre_nslab : ns, 1.0 / (double)HW, 0.f, 1, 0, 0);
            UIG_LAUNCH_CHECK("uig_instnorm_act_bwd(finalize)");
        }
    }
    if (colsum_partial) {
        if (dtype == UIG_BF16)
            hipLaunchKernelGGL((in_apply_bwd_kernel<bf16_t, true>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, stats, gm, colsum_partial, (long)HW, C, CC, na, act, slope, mxq, mxs);
        else
            hipLaunchKernelGGL((in_apply_bwd_kernel<float, true>), dim3(na, B), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, stats, gm, colsum_partial, (long)HW, C, CC, na, act, slope);
    } else {
        if (dtype == UIG_BF16)
            hipLaunchKernelGGL((in_apply_bwd_kernel<bf16_t, false>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, stats, gm, (float*)nullptr, (long)HW, C, CC, na, act, slope);
        else
            hipLaunchKernelGGL((in_apply_bwd_kernel<float, false>), dim3(na, B), dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, stats, gm, (float*)nullptr, (long)HW, C, CC, na, act, slope);
    }
    UIG_LAUNCH_CHECK("uig_instnorm_act_bwd(apply)");
    return 0;
}

// forward with the (sum, sum^2) partials already produced by the convolution's epilogue (uig_conv_gather_ex): finalize + apply
extern "C" int uig_instnorm_act_fwd_pre(const void* x, const void* residual, void* y, float* stats, const float* partial, int nslab,
                                        int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && stats && partial && nslab > 0, "uig_instnorm_act_fwd_pre: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_fwd_pre: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_fwd_pre", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int na = apply_slabs(HW, CC);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, s, partial, stats, B * C, C, nslab, 1.0 / (double)HW, eps, 0, 0, 0);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_pre(finalize)");
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_apply_fwd_kernel<bf16_t>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, stats, (long)HW, C, CC, na, act, slope);
    else
        hipLaunchKernelGGL((in_apply_fwd_kernel<float>), dim3(na, B), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, stats, (long)HW, C, CC, na, act, slope);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_pre(apply)");
    return 0;
}

// the finalize launch alone: epilogue partials of uig_conv_gather_ex (nslab per image) -> stats (mean, rstd) fp32[B][C][2]; for a
// consumer that applies the norm itself (uig_conv3x3_innorm_fwd)
extern "C" int uig_instnorm_finalize(const float* partial, int nslab, float* stats, int B, int64_t HW, int C, float eps, void* stream) {
    UIG_CHECK_ARG(partial && stats && nslab > 0 && B > 0 && C > 0 && HW > 0, "uig_instnorm_finalize: bad arguments");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partial, stats, B * C, C, nslab, 1.0 / (double)HW, eps, 0, 0, 0);
    UIG_LAUNCH_CHECK("uig_instnorm_finalize");
    return 0;
}

// the backward counterpart: (sum g, sum g*xhat) partials -> gm fp32[B][C][2] = (mean g, mean g*xhat)
int uig_instnorm_finalize_bwd(const float* partial, int nslab, float* gm, int B, int64_t HW, int C, void* stream) {
    UIG_CHECK_ARG(partial && gm && nslab > 0 && B > 0 && C > 0 && HW > 0, "uig_instnorm_finalize_bwd: bad arguments");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, (hipStream_t)stream, partial, gm, B * C, C, nslab, 1.0 / (double)HW, 0.f, 1, 0, 0);
    UIG_LAUNCH_CHECK("uig_instnorm_finalize_bwd");
    return 0;
}

extern "C" int uig_instnorm_act_bwd(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                    int B, int64_t HW, int C, int act, float slope, int dtype, void* stream) {
    return instnorm_bwd_impl(dy, x, stats, dx, workspace, nullptr, B, HW, C, act, slope, dtype, stream);
}

extern "C" int uig_instnorm_act_bwd_colsum(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                           float* colsum_partial, int B, int64_t HW, int C, int act, float slope, int dtype,
                                           void* stream) {
    UIG_CHECK_ARG(colsum_partial, "uig_instnorm_act_bwd_colsum: null colsum_partial");
    return instnorm_bwd_impl(dy, x, stats, dx, workspace, colsum_partial, B, HW, C, act, slope, dtype, stream);
}

extern "C" int uig_bias_grad_from_partials(const float* colsum_partial, float* db, int nslab_total, int C, int Nreal,
                                           int accumulate, void* stream) {
    UIG_CHECK_ARG(colsum_partial && db && nslab_total > 0 && Nreal > 0 && Nreal <= C, "uig_bias_grad_from_partials: bad args");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, colsum_partial, db, C, C, nslab_total, 1.0, 0.f, 2, Nreal, accumulate);
    UIG_LAUNCH_CHECK("uig_bias_grad_from_partials");
    return 0;
}

extern "C" int uig_bias_grad(const void* dy, float* db, float* workspace, int64_t pixels, int C, int Nreal,
                             int accumulate, int dtype, void* stream) {
    UIG_CHECK_ARG(dy && db && workspace, "uig_bias_grad: null pointer");
    UIG_CHECK_ARG(Nreal > 0 && Nreal <= C, "uig_bias_grad: bad Nreal=%d C=%d", Nreal, C);
    int CC; if (int r = check_in_args("uig_bias_grad", 1, pixels, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int ns = stats_slabs(pixels, CC);
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_stats_kernel<bf16_t, 0>), dim3(ns, 1), dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)nullptr, (const float*)nullptr, workspace, (long)pixels, C, CC, ns, 0, 0.f);
    else
        hipLaunchKernelGGL((in_stats_kernel<float, 0>), dim3(ns, 1), dim3(256), 0, s, (const float*)dy, (const float*)nullptr, (const float*)nullptr, workspace, (long)pixels, C, CC, ns, 0, 0.f);
    UIG_LAUNCH_CHECK("uig_bias_grad(partial)");
    hipLaunchKernelGGL(in_finalize_kernel, dim3((C + 15) / 16), dim3(256), 0, s, workspace, db, C, C, ns, 1.0, 0.f, 2, Nreal, accumulate);
    UIG_LAUNCH_CHECK("uig_bias_grad(finalize)");
    return 0;
}

// Inference forward (no statistics kept): partial != NULL -> the producing convolution's epilogue partials (np per image), ONE
// launch; partial == NULL -> statistics pass into `workspace`, then the fused apply: two launches instead of three.
static int g_infer_cs = 1;      // tuning / A-B hook: 0 = the whole-C form of the inference InstanceNorm (round 2)
extern "C" void uig_debug_set_infer_cs(int on) { g_infer_cs = on; }
extern "C" int uig_instnorm_act_fwd_infer(const void* x, const void* residual, void* y, const float* partial, int np, float* workspace,
                                          int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && (partial || workspace), "uig_instnorm_act_fwd_infer: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_fwd_infer: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_fwd_infer", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    if (partial == nullptr) {
        np = stats_slabs(HW, CC);
        if (dtype == UIG_BF16)
            hipLaunchKernelGGL((in_stats_kernel<bf16_t, 0>), dim3(np, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, np, 0, 0.f);
        else
            hipLaunchKernelGGL((in_stats_kernel<float, 0>), dim3(np, B), dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, np, 0, 0.f);
        UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_infer(stats)");
        partial = workspace;
    }
    UIG_CHECK_ARG(np > 0, "uig_instnorm_act_fwd_infer: np=%d", np);
    // blocks per image: every block pays the np * C * 8-byte finalize prologue (L2 reads).  Measured (scripts/bench_in_fin.py, 64x64x256,
    // MI355X): 4 pixels per thread (the training apply's shape) is best at batch 1, 8 at batch 8 and 16; 16 is worse everywhere
    if (g_infer_cs && C % 64 == 0 && B <= 65535) {      // channel-sliced blocks: the finalize prologue reads 64 channels' partials only
        const int E = dtype == UIG_BF16 ? 8 : 4, PLG = 256 / (64 / E);
        const int npx = (int)std::max<long>(1, std::min<long>(4096, HW / (PLG * (B <= 2 ? 2 : 8))));
        if (dtype == UIG_BF16)
            hipLaunchKernelGGL((in_apply_fwd_fin_cs_kernel<bf16_t>), dim3(npx, C / 64, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, partial, np, 1.0 / (double)HW, eps, (long)HW, C, npx, act, slope);
        else
            hipLaunchKernelGGL((in_apply_fwd_fin_cs_kernel<float>), dim3(npx, C / 64, B), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, partial, np, 1.0 / (double)HW, eps, (long)HW, C, npx, act, slope);
        UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_infer(apply, channel-sliced)");
        return 0;
    }
    const int na = (int)std::max<long>(1, std::min<long>(2048, HW * CC / (256 * (B <= 2 ? 4 : 8))));
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_apply_fwd_fin_kernel<bf16_t>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, partial, np, 1.0 / (double)HW, eps, (long)HW, C, CC, na, act, slope);
    else
        hipLaunchKernelGGL((in_apply_fwd_fin_kernel<float>), dim3(na, B), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, partial, np, 1.0 / (double)HW, eps, (long)HW, C, CC, na, act, slope);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_infer(apply)");
    return 0;
}

// ---- the same two launches with the MX fp8 form of their output as a second result (BASELINE configs[4]): mx_q [B*HW][C] e4m3
// bytes + mx_s [B*HW][C/32] E8M0 bytes of exactly the bf16 values written to y / dx (uig_mx_quantize of that tensor, fused: the
// stand-alone quantiser is a full extra read + write pass per fp8 convolution).  bf16 only, C a multiple of 32.
extern "C" int uig_instnorm_act_fwd_mx(const void* x, const void* residual, void* y, float* stats, const float* partial, int nslab,
                                       float* workspace, void* mx_q, void* mx_s,
                                       int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && stats && mx_q && mx_s && (partial || workspace), "uig_instnorm_act_fwd_mx: null pointer");
    UIG_CHECK_ARG(dtype == UIG_BF16 && C % 32 == 0, "uig_instnorm_act_fwd_mx: bf16 and C %% 32 == 0 only (C=%d)", C);
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_fwd_mx: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_fwd_mx", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    int ns = nslab;
    if (partial == nullptr) {
        ns = stats_slabs(HW, CC);
        hipLaunchKernelGGL((in_stats_kernel<bf16_t, 0>), dim3(ns, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, ns, 0, 0.f);
        UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_mx(stats)");
        partial = workspace;
    }
    const int na = apply_slabs(HW, CC);
    hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, s, partial, stats, B * C, C, ns, 1.0 / (double)HW, eps, 0, 0, 0);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_mx(finalize)");
    hipLaunchKernelGGL((in_apply_fwd_kernel<bf16_t>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, stats, (long)HW, C, CC, na, act, slope,
                       (unsigned char*)mx_q, (unsigned char*)mx_s);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_mx(apply)");
    return 0;
}

extern "C" int uig_instnorm_act_bwd_colsum_mx(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                              float* colsum_partial, void* mx_q, void* mx_s, int B, int64_t HW, int C, int act, float slope,
                                              int dtype, void* stream) {
    UIG_CHECK_ARG(colsum_partial && mx_q && mx_s, "uig_instnorm_act_bwd_colsum_mx: null pointer");
    UIG_CHECK_ARG(dtype == UIG_BF16 && C % 32 == 0, "uig_instnorm_act_bwd_colsum_mx: bf16 and C %% 32 == 0 only (C=%d)", C);
    return instnorm_bwd_impl(dy, x, stats, dx, workspace, colsum_partial, B, HW, C, act, slope, dtype, stream, (unsigned char*)mx_q, (unsigned char*)mx_s);
}

extern "C" int uig_instnorm_act_bwd_colsum_pre(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                               float* colsum_partial, const float* partial, int nslab, void* mx_q, void* mx_s,
                                               int B, int64_t HW, int C, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(colsum_partial && partial && nslab > 0, "uig_instnorm_act_bwd_colsum_pre: null pointer");
    UIG_CHECK_ARG((mx_q == nullptr) == (mx_s == nullptr), "uig_instnorm_act_bwd_colsum_pre: mx_q and mx_s go together");
    if (mx_q != nullptr) UIG_CHECK_ARG(dtype == UIG_BF16 && C % 32 == 0, "uig_instnorm_act_bwd_colsum_pre: MX output needs bf16 and C %% 32 == 0 (C=%d)", C);
    return instnorm_bwd_impl(dy, x, stats, dx, workspace, colsum_partial, B, HW, C, act, slope, dtype, stream,
                             (unsigned char*)mx_q, (unsigned char*)mx_s, partial, nslab);
}

// ---- round 4: the same operators without the finalize LAUNCH (in-launch finalize by arrival tickets: uig_common.h, UigFin).
// tickets: >= B zero-initialised 32-bit words the launch leaves zero (the caller's arena; one per image).  Results are bit-identical to
// the entry points above (same slabs, same fp64 association order).

// forward, statistics already FINAL (stats fp32[B][C][2] = (mean, rstd): produced inside the convolution launch, uig_conv_gather_fin):
// the apply launch alone.  mx_q / mx_s optional (both or none; bf16, C % 32 == 0): as uig_instnorm_act_fwd_mx.
extern "C" int uig_instnorm_apply_fwd(const void* x, const void* residual, void* y, const float* stats, void* mx_q, void* mx_s,
                                      int B, int64_t HW, int C, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && stats, "uig_instnorm_apply_fwd: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_apply_fwd: bad act %d", act);
    UIG_CHECK_ARG((mx_q == nullptr) == (mx_s == nullptr), "uig_instnorm_apply_fwd: mx_q and mx_s go together");
    if (mx_q != nullptr) UIG_CHECK_ARG(dtype == UIG_BF16 && C % 32 == 0, "uig_instnorm_apply_fwd: MX output needs bf16 and C %% 32 == 0 (C=%d)", C);
    int CC; if (int r = check_in_args("uig_instnorm_apply_fwd", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int na = apply_slabs(HW, CC);
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_apply_fwd_kernel<bf16_t>), dim3(na, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)y, stats, (long)HW, C, CC, na, act, slope,
                           (unsigned char*)mx_q, (unsigned char*)mx_s);
    else
        hipLaunchKernelGGL((in_apply_fwd_kernel<float>), dim3(na, B), dim3(256), 0, s, (const float*)x, (const float*)residual, (float*)y, stats, (long)HW, C, CC, na, act, slope);
    UIG_LAUNCH_CHECK("uig_instnorm_apply_fwd");
    return 0;
}

// forward with its own statistics pass, which finalises itself (two launches instead of three).  mx_q / mx_s optional.
extern "C" int uig_instnorm_act_fwd_t(const void* x, const void* residual, void* y, float* stats, float* workspace, unsigned* tickets,
                                      void* mx_q, void* mx_s, int B, int64_t HW, int C, float eps, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(x && y && stats && workspace && tickets, "uig_instnorm_act_fwd_t: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_fwd_t: bad act %d", act);
    int CC; if (int r = check_in_args("uig_instnorm_act_fwd_t", B, HW, C, dtype, &CC)) return r;
    hipStream_t s = (hipStream_t)stream;
    const int ns = stats_slabs(HW, CC);
    const UigFin fin = g_in_tickets ? make_fin(stats, tickets, workspace, ns, C, (long)HW, eps, 0) : UigFin{};
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_stats_kernel<bf16_t, 0>), dim3(ns, B), dim3(256), 0, s, (const bf16_t*)x, (const bf16_t*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, ns, 0, 0.f, fin);
    else
        hipLaunchKernelGGL((in_stats_kernel<float, 0>), dim3(ns, B), dim3(256), 0, s, (const float*)x, (const float*)nullptr, (const float*)nullptr, workspace, (long)HW, C, CC, ns, 0, 0.f, fin);
    UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_t(stats)");
    if (!g_in_tickets) {
        hipLaunchKernelGGL(in_finalize_kernel, dim3((B * C + 15) / 16), dim3(256), 0, s, workspace, stats, B * C, C, ns, 1.0 / (double)HW, eps, 0, 0, 0);
        UIG_LAUNCH_CHECK("uig_instnorm_act_fwd_t(finalize)");
    }
    return uig_instnorm_apply_fwd(x, residual, y, stats, mx_q, mx_s, B, HW, C, act, slope, dtype, stream);
}

// backward (+ column-sum partials of dx, + optional MX form of dx).  Where (mean g, mean g*xhat) come from:
//   pre_gm != NULL      fp32[B][C][2], already final (finalised inside the launch that wrote dy): the apply launch alone;
//   pre_partial != NULL that launch's epilogue partials (pre_nslab per image): finalize launch + apply (uig_instnorm_act_bwd_colsum_pre);
//   neither             this norm's own statistics pass, finalising itself through `tickets` (two launches instead of three).
extern "C" int uig_instnorm_act_bwd_colsum_t(const void* dy, const void* x, const float* stats, void* dx, float* workspace,
                                             float* colsum_partial, const float* pre_partial, int pre_nslab, const float* pre_gm,
                                             unsigned* tickets, void* mx_q, void* mx_s,
                                             int B, int64_t HW, int C, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(colsum_partial, "uig_instnorm_act_bwd_colsum_t: null colsum_partial");
    UIG_CHECK_ARG(pre_gm || pre_partial || tickets, "uig_instnorm_act_bwd_colsum_t: needs pre_gm, pre_partial or tickets");
    UIG_CHECK_ARG(pre_partial == nullptr || pre_nslab > 0, "uig_instnorm_act_bwd_colsum_t: pre_nslab=%d", pre_nslab);
    UIG_CHECK_ARG((mx_q == nullptr) == (mx_s == nullptr), "uig_instnorm_act_bwd_colsum_t: mx_q and mx_s go together");
    if (mx_q != nullptr) UIG_CHECK_ARG(dtype == UIG_BF16 && C % 32 == 0, "uig_instnorm_act_bwd_colsum_t: MX output needs bf16 and C %% 32 == 0 (C=%d)", C);
    return instnorm_bwd_impl(dy, x, stats, dx, workspace, colsum_partial, B, HW, C, act, slope, dtype, stream,
                             (unsigned char*)mx_q, (unsigned char*)mx_s, pre_partial, pre_nslab, tickets, pre_gm);
}

// ---- fused backward (round 4): plan and entry point
static int g_in_fused = 1;      // A/B hook: 0 = uig_instnorm_bwd_fused_applicable answers 0
extern "C" void uig_debug_set_in_fused(int on) { g_in_fused = on; }
static int device_cus_in() {
    static const int n = [] {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) v = 0;
        return v;
    }();
    return n;
}
template <typename K> static int blocks_per_cu(K kern) {
    int n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kern, 256, 0) != hipSuccess) n = 0;
    return n;
}
// blocks per image of the fused backward for this shape, or 0 when it does not apply: 8 pixels per thread (NB = ceil(HW / (8 * pixel
// lanes))), at most 256 blocks per image, and the WHOLE grid (images rounded up to eight) resident at once.
extern "C" int uig_instnorm_bwd_fused_applicable(int B, int64_t HW, int C, int dtype) {
    if (!g_in_fused || (dtype != UIG_BF16 && dtype != UIG_F32) || B <= 0 || HW <= 0 || C <= 0) return 0;
    const int E = dtype == UIG_BF16 ? 8 : 4;
    if (C % E != 0) return 0;
    const int CC = C / E;
    if (CC > 256 || 256 % CC != 0 || (long)HW * C * (dtype == UIG_BF16 ? 2 : 4) >= (1L << 31)) return 0;
    const int PL = 256 / CC;
    const long NB = (HW + 8L * PL - 1) / (8L * PL);
    if (NB < 1 || NB > 256) return 0;
    static const int occ_b = blocks_per_cu(in_bwd_fused_kernel<bf16_t, true>), occ_f = blocks_per_cu(in_bwd_fused_kernel<float, true>);
    const long capacity = (long)device_cus_in() * (dtype == UIG_BF16 ? occ_b : occ_f);
    const long grid = 8L * ((B + 7) / 8) * NB;
    return grid <= capacity ? (int)NB : 0;
}

// aten::native_batch_norm_backward (+ activation backward) in one launch (see in_bwd_fused_kernel).  partial fp32[B][NB][C][2] and
// gm fp32[B][C][2]: scratch; colsum_partial fp32[B][NB][C][2] (NB slabs per image: the bias gradient of the convolution in front);
// sync: >= 4 * B zero-initialised 32-bit words (left zero), err: one word the kernel sets to 1 if a bounded wait ran out (results are
// then garbage: check it at the next host synchronisation); mx_q / mx_s optional (as uig_instnorm_act_bwd_colsum_mx).
// Only where uig_instnorm_bwd_fused_applicable(B, HW, C, dtype) = NB > 0.
extern "C" int uig_instnorm_act_bwd_fused(const void* dy, const void* x, const float* stats, void* dx, float* partial, float* gm,
                                          float* colsum_partial, unsigned* sync, unsigned* err, void* mx_q, void* mx_s,
                                          int B, int64_t HW, int C, int act, float slope, int dtype, void* stream) {
    UIG_CHECK_ARG(dy && x && stats && dx && partial && gm && colsum_partial && sync && err, "uig_instnorm_act_bwd_fused: null pointer");
    UIG_CHECK_ARG(act == UIG_ACT_NONE || act == UIG_ACT_RELU || act == UIG_ACT_LRELU, "uig_instnorm_act_bwd_fused: bad act %d", act);
    UIG_CHECK_ARG(mx_q == nullptr && mx_s == nullptr, "uig_instnorm_act_bwd_fused: the MX fp8 form of dx is not produced by this launch (pass NULL; use uig_instnorm_act_bwd_colsum_mx)");
    const int NB = uig_instnorm_bwd_fused_applicable(B, HW, C, dtype);
    UIG_CHECK_ARG(NB > 0, "uig_instnorm_act_bwd_fused: shape B=%d HW=%ld C=%d not admitted (query uig_instnorm_bwd_fused_applicable)", B, (long)HW, C);
    InFusedDesc d{};
    const int E = dtype == UIG_BF16 ? 8 : 4;
    d.HW = HW; d.C = C; d.CC = C / E; d.NB = NB; d.B = B; d.act = act; d.slope = slope;
    d.partial = partial; d.gm = gm; d.sync = sync; d.err = err; d.inv_n = 1.0 / (double)HW;
    d.spin_limit = 4000000u;                               // ~1 s of s_sleep polls: far beyond any healthy wait (microseconds)
    d.cpb = (C + NB - 1) / NB;
    const dim3 grid(8 * ((B + 7) / 8) * NB);
    hipStream_t s = (hipStream_t)stream;
    if (dtype == UIG_BF16)
        hipLaunchKernelGGL((in_bwd_fused_kernel<bf16_t, true>), grid, dim3(256), 0, s, (const bf16_t*)dy, (const bf16_t*)x, (bf16_t*)dx, stats, colsum_partial, d);
    else
        hipLaunchKernelGGL((in_bwd_fused_kernel<float, true>), grid, dim3(256), 0, s, (const float*)dy, (const float*)x, (float*)dx, stats, colsum_partial, d);
    UIG_LAUNCH_CHECK("uig_instnorm_act_bwd_fused");
    return 0;
}
