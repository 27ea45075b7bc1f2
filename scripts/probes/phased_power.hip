// Probe (diagnostic): what the strip kernel's data movement costs in sustained MFMA rate under the power limit.  The phased K-step skeleton
// (two wave groups one barrier apart; R | barrier | M: 32 MFMA 16x16x32 bf16 | barrier) on random bf16 operands, 256 blocks x 512 threads:
//   V 0: operands from registers only (no LDS traffic)          V 1: R = 16 ds_read_b128 (the kernel's 0.5 fragment reads per MFMA)
//   V 2: V 1 + 3 LDS-DMA KiB per wave and step from an L2-resident buffer (the kernel's 24 KB per step and CU)
//   V 3: V 1 with 8 reads per step (half the LDS bytes per MFMA)
// hipcc -O3 --offload-arch=gfx950 phased_power.hip -o phased_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* sink, const unsigned* gsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned h = (tid + blockIdx.x * 512) * 2654435761u + 12345u;
    for (int i = tid; i < 16384; i += 512) {                   // 64 KB of random bf16 pairs in (-1, 1) x small
        h = h * 1664525u + 1013904223u;
        const float a = ((h >> 8) & 0xffff) / 32768.f - 1.f;
        h = h * 1664525u + 1013904223u;
        const float b = (((h >> 8) & 0xffff) / 32768.f - 1.f) * 0.05f;
        __bf16 ab = (__bf16)a, bb = (__bf16)b;
        reinterpret_cast<unsigned*>(smem)[i] = (unsigned)__builtin_bit_cast(unsigned short, ab) | ((unsigned)__builtin_bit_cast(unsigned short, bb) << 16);
    }
    __syncthreads();
    f32x4 acc[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    u32x4 r[16];
    const unsigned a0 = (unsigned)(lane * 16);
#pragma unroll
    for (int kk = 0; kk < 16; ++kk) r[kk] = *reinterpret_cast<const u32x4*>(smem + a0 + kk * 1024);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(gsrc), 0, 1 << 20, 0x00020000);
    if (wave >= 4) __builtin_amdgcn_s_barrier();
    for (int it = 0; it < iters; ++it) {
        if (V == 1 || V == 2) {
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[kk]) : "v"(a0), "n"(kk * 1024) : "memory");
        }
        if (V == 3) {
#pragma unroll
            for (int kk = 0; kk < 8; ++kk) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r[2 * kk]) : "v"(a0), "n"(kk * 2048) : "memory");
        }
        if (V == 2) {
#pragma unroll
            for (int kk = 0; kk < 3; ++kk)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)((__attribute__((address_space(3))) unsigned char*)smem + 65536 + (wave * 3 + kk) * 1024), 16, (int)(lane * 16), ((it & 31) * 24 + wave * 3 + kk) * 1024, 0, 0);
            asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kk = 0; kk < 32; ++kk) {
            bf16x8 a = __builtin_bit_cast(bf16x8, r[kk & 7]), b = __builtin_bit_cast(bf16x8, r[8 + ((kk >> 2) & 7)]);
            acc[kk & 15] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[kk & 15], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    }
    if (wave < 4) __builtin_amdgcn_s_barrier();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    if (s == 12345.678f) sink[tid] = s;
}

template <int V>
static void run(const char* name, float* sink, unsigned* src) {
    const int iters = 4000;
    const double flop = 256.0 * 8 * iters * 32 * 16384.0;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<V>), hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 98304, 0, sink, src, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 4.0) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 98304, 0, sink, src, iters);
        hipDeviceSynchronize(); n += 20;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%-78s %6.0f TFLOP/s sustained\n", name, flop * n / dt / 1e12);
}

int main() {
    float* sink; unsigned* src;
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&src, 1 << 20); hipMemset(src, 0x3c, 1 << 20);
    run<0>("phased skeleton, operands in registers (no LDS traffic)", sink, src);
    run<1>("  + 16 ds_read_b128 per wave and step (0.5 per MFMA: the strip kernel)", sink, src);
    run<3>("  +  8 ds_read_b128 per wave and step (0.25 per MFMA)", sink, src);
    run<2>("  + 16 ds_read_b128 + 3 LDS-DMA KiB per wave and step (24 KB per CU and step)", sink, src);
    run<0>("phased skeleton, operands in registers (again)", sink, src);
    return 0;
}
