// Probe (diagnostic): which lane's scale byte applies to operand byte (lane la, byte ja) of the A operand of
// v_mfma_scale_f32_16x16x128_f8f6f4: A one-hot 1.0 at (la, ja), B = 1.0 everywhere, scale x2 in ONE lane ls at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
__global__ void k(int* out /*[2][2048]: scale lane that doubled the result, for A (0) and B (1) one-hot*/) {
    const int l = threadIdx.x, la = blockIdx.x >> 5, ja = blockIdx.x & 31;
    for (int which = 0; which < 2; ++which) {
        i32x8 hot, ones;
        for (int j = 0; j < 8; ++j) { hot[j] = (l == la && (ja >> 2) == j) ? (0x38 << (8 * (ja & 3))) : 0; ones[j] = 0x38383838; }
        int found = -1, cnt = 0;
        for (int ls = 0; ls < 64; ++ls) {
            const int s = l == ls ? 0x7f7f7f80 : 0x7f7f7f7f;
            f32x4 c = {0.f, 0.f, 0.f, 0.f};
            if (which == 0) c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(hot, ones, c, 0, 0, 0, s, 0, 0x7f7f7f7f);
            else            c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(ones, hot, c, 0, 0, 0, 0x7f7f7f7f, 0, s);
            bool two = false;
            for (int j = 0; j < 4; ++j) two |= c[j] == 2.f;
            if (__any(two)) { found = ls; ++cnt; }
        }
        if (l == 0) out[which * 2048 + blockIdx.x] = found | (cnt << 8);
    }
}
int main() {
    int* d; hipMalloc(&d, 4096 * 4); hipMemset(d, 0xff, 4096 * 4);
    hipLaunchKernelGGL(k, dim3(2048), dim3(64), 0, 0, d);
    std::vector<int> h(4096); hipMemcpy(h.data(), d, 4096 * 4, hipMemcpyDeviceToHost);
    for (int which = 0; which < 2; ++which) {
        int own = 0, oth = 0;
        for (int i = 0; i < 2048; ++i) {
            const int la = i >> 5, ja = i & 31, ls = h[which * 2048 + i] & 255, cnt = h[which * 2048 + i] >> 8;
            if (ls == la && cnt == 1) ++own; else { ++oth; if (oth <= 48) printf("%s operand (lane %d = row/col %d group %d, byte %d) is scaled by lane %d (group %d, idx %d)  [%d lanes matched]\n", which ? "B" : "A", la, la & 15, la >> 4, ja, ls, ls >> 4, ls & 15, cnt); }
        }
        printf("%s: scaled by its own lane: %d, by another lane: %d\n", which ? "B" : "A", own, oth);
    }
    return 0;
}
