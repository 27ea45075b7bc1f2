// uig_common.h — shared device helpers for the gfx950 kernels (wave64, MFMA 16x16, LDS tiles of 128-byte rows).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/uig.h"

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4_t;
typedef __attribute__((ext_vector_type(2))) unsigned int u32x2_t;

struct bf16_t { unsigned short v; };   // storage-only bf16

__device__ __forceinline__ float bf16_to_f32(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
// round-to-nearest-even; NaN stays NaN (compiler emits v_cvt_pk_bf16_f32 for the __bf16 cast on gfx950)
__device__ __forceinline__ unsigned short f32_to_bf16(float f) {
    __bf16 b = (__bf16)f;
    return __builtin_bit_cast(unsigned short, b);
}

template <typename T> struct ElemTraits;
template <> struct ElemTraits<float> {
    static constexpr int E = 4;   // elements per 16-byte chunk
    static __device__ __forceinline__ float ld(const float* p) { return *p; }
    static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <> struct ElemTraits<bf16_t> {
    static constexpr int E = 8;
    static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(p->v); }
    static __device__ __forceinline__ void st(bf16_t* p, float v) { p->v = f32_to_bf16(v); }
};

// 16-byte chunk <-> E floats
template <typename T> __device__ __forceinline__ void chunk_to_f32(const u32x4_t& c, float* f);
template <> __device__ __forceinline__ void chunk_to_f32<float>(const u32x4_t& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = __uint_as_float(c[i]);
}
template <> __device__ __forceinline__ void chunk_to_f32<bf16_t>(const u32x4_t& c, float* f) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { f[2 * i] = __uint_as_float(c[i] << 16); f[2 * i + 1] = __uint_as_float(c[i] & 0xffff0000u); }
}
template <typename T> __device__ __forceinline__ u32x4_t f32_to_chunk(const float* f);
template <> __device__ __forceinline__ u32x4_t f32_to_chunk<float>(const float* f) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = __float_as_uint(f[i]);
    return c;
}
template <> __device__ __forceinline__ u32x4_t f32_to_chunk<bf16_t>(const float* f) {
    u32x4_t c;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] = (unsigned)f32_to_bf16(f[2 * i]) | ((unsigned)f32_to_bf16(f[2 * i + 1]) << 16);
    return c;
}

__device__ __forceinline__ float apply_act(float v, int act, float slope) {
    switch (act) {
        case UIG_ACT_RELU: return v > 0.f ? v : 0.f;
        case UIG_ACT_LRELU: return v > 0.f ? v : v * slope;
        case UIG_ACT_TANH: return tanhf(v);
        default: return v;
    }
}

// ONE branch on the (launch-constant) activation around a whole epilogue loop: body(actf) is instantiated per activation.  A
// switch per element carries an inlined tanhf body per element to jump over (see store_tile_via_lds).
template <typename Body>
__device__ __forceinline__ void with_act(int act, float slope, Body&& body) {
    if (act == UIG_ACT_NONE) body([](float v) { return v; });
    else if (act == UIG_ACT_RELU) body([](float v) { return v > 0.f ? v : 0.f; });
    else if (act == UIG_ACT_LRELU) body([slope](float v) { return v > 0.f ? v : v * slope; });
    else body([](float v) { return tanhf(v); });
}

__device__ __forceinline__ int reflect_idx(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * n - 2 - i : i); }

// wave64 all-reduce sum via DPP-free shuffles (width 64)
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// ---------------------------------------------------------------------------------------------------------------
// In-launch finalize of InstanceNorm statistics (round 4).  The kernels that produce an image's partial statistics slabs
// (convolution epilogues, the stand-alone statistics pass) used to be followed by a finalize LAUNCH (in_finalize_kernel: 112 per train
// step, 5.3 us each, almost all of it launch boundary).  Here every block takes an ARRIVAL TICKET per image after its slabs are out;
// the block that draws the image's last ticket reduces that image's slabs itself, in exactly in_finalize_kernel's association order
// (16 slab lanes per channel, sequential fp64 sums, xor-shuffle tree): bit-identical (mean, rstd), no extra launch.
// Protocol (cdna_hip_programming.md, "In-launch split-K reduction", sc1 form - a release fence per tile would write back the XCD's
// whole dirty L2, output tile included: 6.5 us): slabs are stored WRITE-THROUGH (sc1: 8-byte agent-scope relaxed atomic stores), every
// storing wave drains its stores (s_waitcnt vmcnt(0)), the block's barrier, ONE relaxed agent-scope fetch_add by one lane; the last
// arriver reads the slabs with sc1 loads only (agent-scope relaxed atomic loads: never served by this CU's L1, no acquire fence
// needed).  One ticket per block and image, not per element.  The ticket words are zero between launches: the last arriver resets its
// word (plus one memset per step by the owner of the arena, so that an aborted launch cannot poison the next step).
struct UigFin {
    float* out;               // fp32[B][C][2]: mode 0 (mean, rstd), mode 1 (mean g, mean g*xhat)
    unsigned* tickets;        // [>= B] zero on entry, zero on exit
    const float* partial;     // fp32[B][nslab][C][2], written by THIS launch with uig_store8_sc1
    int nslab, C;
    unsigned expected;        // sum of the arrival counts of one image
    int mode; float eps;
    double inv_n;
};

__device__ __forceinline__ void uig_store8_sc1(float* p, float a, float b) {      // p 8-byte aligned
    const unsigned long long v = (unsigned long long)__float_as_uint(a) | ((unsigned long long)__float_as_uint(b) << 32);
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long uig_load8_sc1(const float* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// all NTHREADS threads of the block; bit-identical to in_finalize_kernel (modes 0 / 1) on image `img`.
// 16 slab lanes per channel PAIR (one 16-byte sc1 load = both channels' (sum, sum^2) of one slab); the loads of four passes x four slab
// rounds are issued before the first sum (the last arriver is alone with a cold 128-KB read: its time is round trips, not bytes - the
// first form, one pass of four loads at a time, cost 20 us per convolution launch); slab rows past nslab are read out of the buffer's
// range (zeros: x + 0.0 == x for the never-negative-zero partial sums).  Sums per channel in slab order, then the xor-shuffle tree.
template <int NTHREADS>
__device__ __forceinline__ void uig_fin_image(const UigFin& f, int img, int tid) {
    static_assert(NTHREADS % 64 == 0, "whole waves");
    constexpr int IPP = NTHREADS / 16, PU = 4, SU = 4;   // channel pairs per pass; passes / slab rounds in flight together
    const int sl = tid & 15, it = tid >> 4;
    const int npair = f.C >> 1;
    const unsigned rowb = (unsigned)f.C * 8u, imgb = (unsigned)f.nslab * rowb;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(f.partial + (long)img * f.nslab * f.C * 2), 0, imgb, 0x00020000);
    for (int p0 = 0; p0 < npair; p0 += IPP * PU) {
        double acc[PU][4];
        unsigned poff[PU];
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            acc[u][0] = acc[u][1] = acc[u][2] = acc[u][3] = 0.0;
            poff[u] = (unsigned)min(p0 + u * IPP + it, npair - 1) * 16u;      // clamped: a pair past the end re-reads the last one, never stored
        }
        for (int s0 = 0; s0 < f.nslab; s0 += 16 * SU) {
            u32x4_t v[PU][SU];
#pragma unroll
            for (int k = 0; k < SU; ++k) {
                const int srow = s0 + 16 * k + sl;
                const unsigned ro = srow < f.nslab ? (unsigned)srow * rowb : imgb;      // out of range -> zeros
#pragma unroll
                for (int u = 0; u < PU; ++u)
                    v[u][k] = __builtin_bit_cast(u32x4_t, __builtin_amdgcn_raw_buffer_load_b128(rs, (int)(ro + (srow < f.nslab ? poff[u] : 0u)), 0, 16));   // aux 16 = sc1
            }
#pragma unroll
            for (int k = 0; k < SU; ++k)
#pragma unroll
                for (int u = 0; u < PU; ++u)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[u][e] += (double)__uint_as_float(v[u][k][e]);
        }
#pragma unroll
        for (int o = 8; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < PU; ++u)
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[u][e] += __shfl_xor(acc[u][e], o, 16);
#pragma unroll
        for (int u = 0; u < PU; ++u) {
            const int pr = p0 + u * IPP + it;
            if (sl == 0 && pr < npair) {
                float r[4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const double a = acc[u][2 * h], q = acc[u][2 * h + 1];
                    if ((f.mode & 15) == 0) {
                        const double mean = a * f.inv_n;
                        double var = q * f.inv_n - mean * mean;
                        if (var < 0.0) var = 0.0;
                        r[2 * h] = (float)mean; r[2 * h + 1] = (float)(1.0 / sqrt(var + (double)f.eps));
                    } else {
                        r[2 * h] = (float)(a * f.inv_n); r[2 * h + 1] = (float)(q * f.inv_n);
                    }
                }
                *reinterpret_cast<f32x4_t*>(f.out + ((long)img * f.C + 2 * pr) * 2) = f32x4_t{r[0], r[1], r[2], r[3]};
            }
        }
    }
}

// Called by ALL threads of the block (block-uniform arguments) after the block's slabs of image `img` were stored with
// uig_store8_sc1; `count` = this block's share of f.expected; lds_word = any 4-byte LDS word nobody else uses around the call.
template <int NTHREADS>
__device__ __forceinline__ void uig_fin_arrive(const UigFin& f, int img, unsigned count, unsigned* lds_word, int tid) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's slab stores are through
    __syncthreads();                                       // ... and every other wave's of the block
    const int dbg = f.mode >> 4;                          // timing diagnostics only (uig_debug_set_in_tickets 2 / 3): wrong statistics
    if (tid == 0) {
        unsigned last = 0u;
        if (!(dbg & 2)) {
            const unsigned old = __hip_atomic_fetch_add(f.tickets + img, count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            last = (old + count == f.expected) ? 1u : 0u;
            if (last) __hip_atomic_store(f.tickets + img, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        *reinterpret_cast<volatile unsigned*>(lds_word) = last;
    }
    __syncthreads();
    if (*reinterpret_cast<volatile unsigned*>(lds_word) != 0u && !(dbg & 1)) uig_fin_image<NTHREADS>(f, img, tid);
}

// ---------------------------------------------------------------------------------------------------------------
// Output-tile store through LDS for the MFMA conv kernels.  After the MFMAs a lane holds 4 consecutive channels of one
// pixel per 16x16 tile; storing those directly is 16 scattered 8-byte stores per lane (measured: 26 % of the kernel).
// Instead each wave writes its 64-pixel x 64-channel tile (bias + activation applied, converted to T) into its own LDS
// scratch as [pixel][channel] rows with a 16-byte-chunk XOR swizzle, then reads it back row-wise so that consecutive
// lanes hold consecutive 16-byte chunks of one pixel and every store instruction writes whole 128/256-byte rows.
// row_ptr(r) returns the output pointer of tile row r (pixel) at channel 0 of this wave's 64 channels, or nullptr.
// stat_out (optional, wave-uniform): fp32[64][2] slot of the InstanceNorm partial-statistics buffer for this wave's 64
// channels; receives (sum, sum of squares) over the first nvalid pixel rows of the tile AS STORED (rounded to T), so the
// separate statistics pass of the following InstanceNorm disappears.
struct NoRowAdd { __device__ __forceinline__ u32x4_t operator()(int, int, const u32x4_t& v, int) const { return v; } };
struct NoMidHook { __device__ __forceinline__ void operator()() const {} };
// row_add(r, c, chunk, i) may modify the 16-byte chunk c of tile row r just before it is stored (border terms of the
// reflection-pad input gradient); i is the (compile-time, after unrolling) index of the store instruction: r = r0 + RPI*i.
// mid() runs between the tile's LDS writes (the accumulators are dead from there on) and the row reads: a place to issue
// loads whose destination registers may then reuse the accumulators'.
template <typename T, int MT, int NT, typename RowPtr, typename RowAdd = NoRowAdd, typename MidHook = NoMidHook>
__device__ __forceinline__ void store_tile_via_lds(const f32x4_t (&acc)[NT][MT], unsigned char* scratch, int lane,
                                                   const float* bias4 /* NT*4 bias values of this lane, or nullptr */,
                                                   int act, float slope, RowPtr row_ptr, float* stat_out = nullptr,
                                                   int nvalid = 64, RowAdd row_add = RowAdd(), MidHook mid = MidHook(), bool stat_sc1 = false) {
    static_assert(MT == 4 && (NT == 4 || NT == 2 || NT == 1) && NT * 16 * sizeof(T) >= 32, "64-pixel x 64- / 32- / 16-channel wave tile");
    constexpr int ROWB = NT * 16 * (int)sizeof(T);     // bytes per pixel row of the wave tile: 128 (bf16) / 256 (f32) at 64 channels; 32 at 16 bf16 channels
    constexpr int NCH = ROWB / 16;                     // 16-byte chunks per row
    const int l16 = lane & 15, q = lane >> 4;
    // The activation is a launch constant: ONE branch around the whole conversion loop, not a switch per element.  With the switch
    // inside, each of the 64 elements carried an inlined tanhf body to jump over: ~20k instructions of epilogue whose fetch (not
    // its arithmetic) cost 7.5k cycles per tile in the persistent strip kernel (s_memtime stamps).
    auto to_lds = [&](auto actf) {
#pragma unroll
        for (int b = 0; b < MT; ++b) {
            const int m = b * 16 + l16;
#pragma unroll
            for (int a = 0; a < NT; ++a) {
                float v[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = actf(acc[a][b][e] + (bias4 ? bias4[a * 4 + e] : 0.f));
                if constexpr (sizeof(T) == 4) {
                    const int chunk = (4 * a + q) ^ (m & (NCH - 1));
                    *reinterpret_cast<f32x4_t*>(scratch + m * ROWB + chunk * 16) = f32x4_t{v[0], v[1], v[2], v[3]};
                } else {
                    const int chunk = (2 * a + (q >> 1)) ^ (m & (NCH - 1));
                    u32x2_t pk;
                    pk[0] = (unsigned)f32_to_bf16(v[0]) | ((unsigned)f32_to_bf16(v[1]) << 16);
                    pk[1] = (unsigned)f32_to_bf16(v[2]) | ((unsigned)f32_to_bf16(v[3]) << 16);
                    *reinterpret_cast<u32x2_t*>(scratch + m * ROWB + chunk * 16 + (q & 1) * 8) = pk;
                }
            }
        }
    };
    with_act(act, slope, to_lds);
    // The tile is written as 8-byte / f32x4 vectors and read back as u32x4: different vector types, which type-based alias
    // analysis may treat as non-aliasing (seen in conv_cin8.hip: reads scheduled above the writes).  Compiler barrier.
    asm volatile("" ::: "memory");
    mid();
    // the scratch is private to the wave: only its own LDS writes must have landed (the compiler inserts lgkmcnt waits)
    constexpr int RPI = 64 / NCH;                      // rows per store instruction (8 for bf16, 4 for f32)
    constexpr int EC = ElemTraits<T>::E;               // elements per 16-byte chunk
    const int c = lane % NCH, r0 = lane / NCH;
    // Fused InstanceNorm statistics: every lane sums the chunks it stores (its 64 / RPI rows of chunk c, as stored = rounded
    // to T), then the RPI lanes that hold the same chunk are combined by xor-shuffles: no second pass over the scratch (the
    // first version walked the 64 rows with one dependent 2-byte LDS read per row: ~7.7k cycles of the epilogue).
    float s1[EC], s2[EC];
#pragma unroll
    for (int e = 0; e < EC; ++e) { s1[e] = 0.f; s2[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < 64 / RPI; ++i) {
        const int r = r0 + RPI * i;
        T* dst = row_ptr(r);
        const u32x4_t val = *reinterpret_cast<const u32x4_t*>(scratch + r * ROWB + ((c ^ (r & (NCH - 1))) * 16));
        if (dst != nullptr) *reinterpret_cast<u32x4_t*>(reinterpret_cast<unsigned char*>(dst) + c * 16) = row_add(r, c, val, i);
        if (stat_out != nullptr) {                     // wave-uniform
            float f[EC];
            chunk_to_f32<T>(val, f);
            const bool in = r < nvalid;
#pragma unroll
            for (int e = 0; e < EC; ++e) { const float v = in ? f[e] : 0.f; s1[e] += v; s2[e] += v * v; }
        }
    }
    if (stat_out != nullptr) {
#pragma unroll
        for (int o = NCH; o < 64; o <<= 1)
#pragma unroll
            for (int e = 0; e < EC; ++e) { s1[e] += __shfl_xor(s1[e], o, 64); s2[e] += __shfl_xor(s2[e], o, 64); }
        if (lane < NCH) {                              // lane c holds channels c*EC .. c*EC+EC-1: 2*EC contiguous floats
            float* o = stat_out + lane * EC * 2;
            if (stat_sc1) {                            // wave-uniform: an in-launch finalize reads them (UigFin): write-through stores
#pragma unroll
                for (int e = 0; e < EC; ++e) uig_store8_sc1(o + 2 * e, s1[e], s2[e]);
            } else {
#pragma unroll
                for (int e = 0; e < EC; e += 2) *reinterpret_cast<f32x4_t*>(o + 2 * e) = f32x4_t{s1[e], s2[e], s1[e + 1], s2[e + 1]};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
// MX block-scaled fp8 quantisation of 8 consecutive channels held by this lane; the 4 lanes that hold one 32-channel block are
// adjacent (lane & 3 = position in the block) and must all be active.  Returns the 8 e4m3 bytes; sbyte = the block's E8M0
// scale byte: 2^(floor(log2(amax)) - 8) (e4m3 emax = 8), 127 (x1) for an all-zero block.  Elements = RNE(x / scale) clamped to
// +-448 first (v_cvt_pk_fp8_f32 returns NaN above 464).  One definition for the stand-alone quantiser and the fused epilogues.
__device__ __forceinline__ u32x2_t mx_quantize8(const float* f, int& sbyte) {
    float am = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) am = fmaxf(am, fabsf(f[e]));
    am = fmaxf(am, __shfl_xor(am, 1, 64));
    am = fmaxf(am, __shfl_xor(am, 2, 64));
    const int eb = (int)((__float_as_uint(am) >> 23) & 0xff);                  // biased exponent of amax (0 for zero / subnormal)
    const int sb = am == 0.f ? 127 : min(max(eb - 8, 0), 254);
    const float inv = __uint_as_float((unsigned)(254 - sb) << 23);              // 2^(127 - sb), exact (sb <= 246 for finite input)
    unsigned w[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fminf(fmaxf(f[4 * h + e] * inv, -448.f), 448.f);
        int p = __builtin_amdgcn_cvt_pk_fp8_f32(v[0], v[1], 0, false);
        p = __builtin_amdgcn_cvt_pk_fp8_f32(v[2], v[3], p, true);
        w[h] = (unsigned)p;
    }
    sbyte = sb;
    return u32x2_t{w[0], w[1]};
}

// optional requests that ride on a uig_conv_gather* launch; `done` is set by the kernel family that served the request inside its
// launch - conv_gather_impl runs the finalize launch for whatever is left (the results are bit-identical either way)
struct UigBst { const void* x; const float* stats; float* partial; int act; float slope; float* gm; unsigned* tickets; int done; };
struct UigFinReq { float* stats; float eps; unsigned* tickets; int done; };
bool uig_in_tickets_on();      // instnorm.hip: the A/B hook uig_debug_set_in_tickets
int uig_in_tickets_dbg();

// host-side error plumbing (defined in uig_capi.hip)
int uig_set_error(int code, const char* fmt, ...);
void uig_note_conv_kernel(int id);      // UIG_K_* of include/uig.h, read back by uig_debug_last_conv_kernel()

// One-time raise of a kernel's dynamic-LDS limit (hipFuncAttributeMaxDynamicSharedMemorySize), callable from any host
// thread: the attribute call is idempotent, the flag is published (release) only after it succeeded, and a function-local
// `static SmemAttrOnce` is initialised thread-safely by the language.  One device per process (the DP model of this library).
#include <atomic>
struct SmemAttrOnce {
    std::atomic<bool> done{false};
    hipError_t ensure(const void* fn, size_t bytes) {
        if (done.load(std::memory_order_acquire)) return hipSuccess;
        hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
        if (e == hipSuccess) done.store(true, std::memory_order_release);
        return e;
    }
};
#define UIG_CHECK_ARG(cond, ...) do { if (!(cond)) return uig_set_error(-1, __VA_ARGS__); } while (0)
#define UIG_LAUNCH_CHECK(name) do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return uig_set_error((int)e_, "%s: launch failed: %s", name, hipGetErrorString(e_)); } while (0)
