// Probe (diagnostic): sustained bf16 MFMA rate of gfx950 under its power limit, 16x16x32 against 32x32x16, random against zero operands.
// 256 blocks x 512 threads (two waves per SIMD), registers only.   hipcc -O3 --offload-arch=gfx950 mfma_power.hip -o mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;

template <int SHAPE, bool RANDOM>
__global__ __launch_bounds__(512, 2) void k(float* sink, int iters) {
    const int tid = threadIdx.x + blockIdx.x * 512;
    bf16x8 a[4], b[4];
    unsigned h = tid * 2654435761u + 12345u;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            h = h * 1664525u + 1013904223u;
            const float va = RANDOM ? ((h >> 8) & 0xffff) / 32768.f - 1.f : 0.f;
            h = h * 1664525u + 1013904223u;
            const float vb = RANDOM ? ((h >> 8) & 0xffff) / 32768.f - 1.f : 0.f;
            a[i][e] = (__bf16)va; b[i][e] = (__bf16)(vb * 0.05f);
        }
    float s = 0.f;
    if (SHAPE == 16) {
        f32x4 acc[16];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][3];
    } else {
        f32x16 acc[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i & 3], b[(i >> 1) & 3], acc[i & 3], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][15];
    }
    if (s == 12345.678f) sink[tid] = s;
}

template <int SHAPE, bool RANDOM>
static void run(const char* name, float* sink) {
    const int iters = 20000;                                 // per launch: 16 (8) MFMAs of 16384 (32768) FLOP per iteration and wave
    const double flop = 256.0 * 8 * iters * 16 * 16384.0;
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, sink, iters);
    hipDeviceSynchronize();
    auto t0 = std::chrono::steady_clock::now();
    int n = 0;
    while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 3.0) {
        for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<SHAPE, RANDOM>), dim3(256), dim3(512), 0, 0, sink, iters);
        hipDeviceSynchronize(); n += 20;
    }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("%-44s %7.0f TFLOP/s sustained over %.1f s\n", name, flop * n / dt / 1e12, dt);
}

int main() {
    float* sink; hipMalloc(&sink, 256 * 512 * 4);
    run<16, false>("16x16x32 bf16, zero operands", sink);
    run<32, false>("32x32x16 bf16, zero operands", sink);
    run<16, true>("16x16x32 bf16, random operands in [-1, 1)", sink);
    run<32, true>("32x32x16 bf16, random operands in [-1, 1)", sink);
    run<16, true>("16x16x32 bf16, random (again)", sink);
    return 0;
}
