// Probe (diagnostic, not part of the library): what issues BESIDE a wave's back-to-back MFMAs on gfx950 - from its SIMD partner wave and
// from the wave itself.  One 512-thread block per CU: waves 0-3 ("M") run ITERS x 32 v_mfma_f32_16x16x32_bf16 on 16 independent
// accumulators; waves 4-7 ("R", their SIMD partners) run ITERS x a block of other instructions.  s_memtime around each wave's loop.
//   hipcc -O3 --offload-arch=gfx950 scripts/probes/mfma_coissue.hip -o /tmp/mfma_coissue && /tmp/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int MODE_R, bool RUN_M, int MODE_SELF>
__global__ __launch_bounds__(512, 2) void probe(unsigned long long* out, float* sink, const unsigned* gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    const unsigned a0 = (unsigned)(lane * 16);       // conflict-free 16-byte reads: consecutive lanes, consecutive chunks
    if (wave < 4) {
        if (RUN_M) {
            f32x4 acc[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
            bf16x8 a, b;
#pragma unroll
            for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane & 3); b[i] = (__bf16)1.0f; }
            u32x4 r[8];
            unsigned v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[i] = lane + i; r[i] = u32x4{0u, 0u, 0u, 0u}; }
            unsigned sv0 = wave;
            const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(gsrc), 0, 1 << 20, 0x00020000);
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
            for (int it = 0; it < iters; ++it) {
#pragma unroll
                for (int k = 0; k < 32; ++k) {
                    acc[k & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 15], 0, 0, 0);
                    if (MODE_SELF == 1 && (k & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[(k >> 1) & 7]) : "v"(a0), "n"((k >> 1) * 1024) : "memory");
                    if (MODE_SELF == 2) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[k & 7]) : "v"(lane));
                    if (MODE_SELF == 4 || MODE_SELF == 5) {      // 3 LDS-DMA pieces (+ 5: 16 s_add, 4 v_add) spread over the wave's own MFMAs; no wait in the loop body
                        if (k == 2 || k == 12 || k == 22)
                            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) unsigned char*)smem + 32768 + (wave * 3 + k / 10) * 1024), 16, (int)(lane * 16), (wave * 3 + k / 10) * 1024, 0, 0);
                        if (MODE_SELF == 5 && (k & 1)) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv0) :: "scc");
                        if (MODE_SELF == 5 && (k & 7) == 3) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[k & 7]) : "v"(lane));
                        if (k == 31) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
                    }
                    if (MODE_SELF == 3) { asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[k & 7]) : "v"(lane)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[(k + 4) & 7]) : "v"(lane)); asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[(k + 2) & 7]) : "v"(lane)); }
                }
                if (MODE_SELF == 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            }
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
            for (int i = 0; i < 8; ++i) s += (float)(v[i] + r[i][0]);
            s += (float)sv0;
            if (s == 12345.678f) sink[tid] = s;
        }
    } else {
        u32x4 r[16];
        unsigned v[8];
        unsigned sv0 = wave, sv1 = wave + 1, sv2 = wave + 2, sv3 = wave + 3;
#pragma unroll
        for (int i = 0; i < 8; ++i) v[i] = lane + i;
#pragma unroll
        for (int i = 0; i < 16; ++i) r[i] = u32x4{0u, 0u, 0u, 0u};
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(gsrc), 0, 1 << 20, 0x00020000);
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
        if (MODE_R != 0)
        for (int it = 0; it < iters; ++it) {
            if (MODE_R == 1 || MODE_R == 5) {        // 16 ds_read_b128, one address register, immediate offsets
#pragma unroll
                for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a0), "n"(k * 1024) : "memory");
            }
            if (MODE_R == 2 || MODE_R == 5) {        // 32 (mode 5: 16) independent VALU adds
#pragma unroll
                for (int k = 0; k < (MODE_R == 5 ? 16 : 32); ++k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[k & 7]) : "v"(lane));
            }
            if (MODE_R == 3 || MODE_R == 5) {        // 32 SALU adds
#pragma unroll
                for (int k = 0; k < 8; ++k) { asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv0) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv1) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv2) :: "scc"); asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv3) :: "scc"); }
            }
            if (MODE_R == 4 || MODE_R == 5) {        // 3 LDS-DMA pieces of 1 KiB (L2-resident source)
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) unsigned char*)smem + 32768 + (wave * 3 + k) * 1024), 16, (int)(lane * 16), (wave * 3 + k) * 1024, 0, 0);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (MODE_R == 1 || MODE_R == 5) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
        unsigned s = sv0 + sv1 + sv2 + sv3;
#pragma unroll
        for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
        for (int i = 0; i < 16; ++i) s += r[i][0] ^ r[i][3];
        if (s == 0x12345678u) sink[tid] = (float)s;
    }
    if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

// The phased K-step skeleton of conv_strip_pk.hip (DM 6 / 8): waves 4-7 one barrier behind waves 0-3; step = R (16 ds_read_b128 [+ VALU address
// arithmetic] [+ 3 LDS-DMA + counted wait]) | lgkmcnt(0) | barrier | M (32 MFMA [+ 3 LDS-DMA between them + counted wait]) | barrier.
// V: 0 reads only, 1 + 16 v_add in R, 2 + DMAs in R, 3 DMAs in M instead, 4 = 2 with 20 s_add in R as well, 5 = no skew (all waves R then M together)
template <int V>
__global__ __launch_bounds__(512, 2) void probe_phased(unsigned long long* out, float* sink, const unsigned* gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (int i = tid; i < 16384; i += 512) reinterpret_cast<unsigned*>(smem)[i] = i;
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 r[16];
    unsigned v[8];
    unsigned sv0 = wave;
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = lane + i;
    const unsigned a0 = (unsigned)(lane * 16);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(gsrc), 0, 1 << 20, 0x00020000);
    auto dma3 = [&]() {
#pragma unroll
        for (int k = 0; k < 3; ++k)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) unsigned char*)smem + 32768 + (wave * 3 + k) * 1024), 16, (int)(lane * 16), (wave * 3 + k) * 1024, 0, 0);
    };
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    if (V != 5 && wave >= 4) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[k]) : "v"(a0), "n"(k * 1024) : "memory");
        if (V == 1 || V == 4) {
#pragma unroll
            for (int k = 0; k < 16; ++k) asm volatile("v_add_u32 %0, %1, %0" : "+v"(v[k & 7]) : "v"(lane));
        }
        if (V == 4) {
#pragma unroll
            for (int k = 0; k < 20; ++k) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sv0) :: "scc");
        }
        if (V == 2 || V == 4) { dma3(); asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            bf16x8 a = __builtin_bit_cast(bf16x8, r[k & 7]), b = __builtin_bit_cast(bf16x8, r[8 + (k & 7)]);
            acc[k & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[k & 15], 0, 0, 0);
            if (V == 3 && (k == 2 || k == 12 || k == 22))
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) unsigned char*)smem + 32768 + (wave * 3 + k / 10) * 1024), 16, (int)(lane * 16), (wave * 3 + k / 10) * 1024, 0, 0);
        }
        if (V == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        if (V != 5) __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (V != 5 && wave < 4) __builtin_amdgcn_s_barrier();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory");
    float s = (float)sv0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += (float)v[i];
    if (s == 12345.678f) sink[tid] = s;
    if (lane == 0) out[(size_t)blockIdx.x * 8 + wave] = t1 - t0;
}

template <int V>
static void run_phased(const char* name, unsigned long long* dout, float* dsink, unsigned* dsrc, int iters) {
    const int grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe_phased<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe_phased<V>), dim3(grid), dim3(512), 65536, 0, dout, dsink, dsrc, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> m;
    for (auto x : h) m.push_back((double)x / iters);
    std::sort(m.begin(), m.end());
    printf("%-72s %7.0f cycles per step (1024 of MFMA per SIMD)\n", name, m[m.size() / 2]);
}

template <int MODE_R, bool RUN_M, int MODE_SELF>
static void run(const char* name, unsigned long long* dout, float* dsink, unsigned* dsrc, int iters) {
    const int grid = 256;
    hipFuncSetAttribute(reinterpret_cast<const void*>(probe<MODE_R, RUN_M, MODE_SELF>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL((probe<MODE_R, RUN_M, MODE_SELF>), dim3(grid), dim3(512), 65536, 0, dout, dsink, dsrc, iters);
        hipDeviceSynchronize();
    }
    std::vector<unsigned long long> h(grid * 8);
    hipMemcpy(h.data(), dout, h.size() * 8, hipMemcpyDeviceToHost);
    std::vector<double> m, r;
    for (int b = 0; b < grid; ++b) for (int w = 0; w < 8; ++w) (w < 4 ? m : r).push_back((double)h[b * 8 + w] / iters);
    std::sort(m.begin(), m.end()); std::sort(r.begin(), r.end());
    printf("%-58s M waves %7.0f cycles/iter   R waves %7.0f cycles/iter\n", name, m[m.size() / 2], r[r.size() / 2]);
}

int main() {
    unsigned long long* dout; float* dsink; unsigned* dsrc;
    hipMalloc(&dout, 256 * 8 * 8); hipMalloc(&dsink, 4096); hipMalloc(&dsrc, 1 << 20); hipMemset(dsrc, 1, 1 << 20);
    const int it = 2000;
    printf("per iteration: M = 32 MFMA 16x16x32 bf16 (512 cycles of the matrix pipe); R = the block named\n");
    run<0, true, 0>("M alone", dout, dsink, dsrc, it);
    run<1, false, 0>("R alone: 16 ds_read_b128", dout, dsink, dsrc, it);
    run<1, true, 0>("M | R: 16 ds_read_b128", dout, dsink, dsrc, it);
    run<2, false, 0>("R alone: 32 v_add", dout, dsink, dsrc, it);
    run<2, true, 0>("M | R: 32 v_add", dout, dsink, dsrc, it);
    run<3, false, 0>("R alone: 32 s_add", dout, dsink, dsrc, it);
    run<3, true, 0>("M | R: 32 s_add", dout, dsink, dsrc, it);
    run<4, false, 0>("R alone: 3 LDS-DMA KiB + vmcnt(0)", dout, dsink, dsrc, it);
    run<4, true, 0>("M | R: 3 LDS-DMA KiB + vmcnt(0)", dout, dsink, dsrc, it);
    run<5, false, 0>("R alone: 16 ds_read + 16 v_add + 32 s_add + 3 DMA", dout, dsink, dsrc, it);
    run<5, true, 0>("M | R: 16 ds_read + 16 v_add + 32 s_add + 3 DMA", dout, dsink, dsrc, it);
    run<0, true, 1>("M with 16 ds_read_b128 between its own MFMAs", dout, dsink, dsrc, it);
    run<0, true, 2>("M with 32 v_add between its own MFMAs", dout, dsink, dsrc, it);
    run<0, true, 3>("M with 96 v_add between its own MFMAs", dout, dsink, dsrc, it);
    run<0, true, 4>("M with 3 LDS-DMA KiB between its own MFMAs (vmcnt(3) at the end)", dout, dsink, dsrc, it);
    run<0, true, 5>("M with 3 LDS-DMA + 16 s_add + 4 v_add between its own MFMAs", dout, dsink, dsrc, it);
    run<1, true, 5>("M with 3 DMA + 16 s_add + 4 v_add | R: 16 ds_read_b128", dout, dsink, dsrc, it);
    printf("phased K-step skeleton (two wave groups one barrier apart, R | barrier | M | barrier):\n");
    run_phased<0>("R = 16 ds_read_b128", dout, dsink, dsrc, it);
    run_phased<1>("R = 16 ds_read_b128 + 16 v_add", dout, dsink, dsrc, it);
    run_phased<2>("R = 16 ds_read_b128 + 3 LDS-DMA + vmcnt(3)", dout, dsink, dsrc, it);
    run_phased<4>("R = 16 ds_read_b128 + 16 v_add + 20 s_add + 3 LDS-DMA + vmcnt(3)", dout, dsink, dsrc, it);
    run_phased<3>("R = 16 ds_read_b128; the 3 LDS-DMA between M's MFMAs + vmcnt(3)", dout, dsink, dsrc, it);
    run_phased<5>("no skew, one barrier per step: all waves R | barrier | M", dout, dsink, dsrc, it);
    return 0;
}
