// Probe (diagnostic): K pairing of the A and B operand bytes of v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3, scales = 1).
// A = one-hot 1.0 at (lane la, byte ja); B = one-hot 1.0 at (lane lb, byte jb): D != 0 iff both sit at the same K.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k(int* out /*[2048][16] : for (la,ja) and column c: (lb_group*32 + jb) | row<<16, or -1*/) {
    const int l = threadIdx.x, la = blockIdx.x >> 5, ja = blockIdx.x & 31;
    i32x8 a;
    for (int j = 0; j < 8; ++j) a[j] = (l == la && (ja >> 2) == j) ? (0x38 << (8 * (ja & 3))) : 0;
    for (int gb = 0; gb < 4; ++gb)
        for (int jb = 0; jb < 32; ++jb) {
            i32x8 b;
            for (int j = 0; j < 8; ++j) b[j] = ((l >> 4) == gb && (jb >> 2) == j) ? (0x38 << (8 * (jb & 3))) : 0;   // every column lane of group gb
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            for (int j = 0; j < 4; ++j)
                if (c[j] != 0.f) out[blockIdx.x * 16 + (l & 15)] = (gb * 32 + jb) | ((4 * (l >> 4) + j) << 16) | ((int)c[j] << 24);
        }
}

int main() {
    int* d; hipMalloc(&d, 2048 * 16 * 4);
    hipMemset(d, 0xff, 2048 * 16 * 4);
    hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d);
    std::vector<int> h(2048 * 16);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int natural = 0, other = 0, none = 0;
    for (int la = 0; la < 64; ++la)
        for (int ja = 0; ja < 32; ++ja)
            for (int c = 0; c < 16; ++c) {
                const int v = h[(la * 32 + ja) * 16 + c];
                if (v == -1) { ++none; continue; }
                const int kb = v & 0xffff, row = (v >> 16) & 255, val = v >> 24;
                const bool nat = kb == (la >> 4) * 32 + ja && row == (la & 15) && val == 1;
                nat ? ++natural : ++other;
                if (!nat && other <= 40) printf("A(lane %d, byte %d) pairs with B(group %d, byte %d) at column %d -> D row %d value %d\n", la, ja, kb >> 5, kb & 31, c, row, val);
            }
    printf("natural pairings (same lane group, same byte, row = lane & 15): %d, other: %d, no match: %d\n", natural, other, none);
    return 0;
}
