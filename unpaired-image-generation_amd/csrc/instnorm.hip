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
