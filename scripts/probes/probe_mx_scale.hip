// Probe (diagnostic): which (lane, byte) of the scale registers v_mfma_scale_f32_16x16x128_f8f6f4 consumes for each OPSEL.
// A = B = 1.0 (e4m3 0x38) everywhere, every scale byte = 127 (x1) except ONE byte of ONE lane of the A (or B) scale register
// = 128 (x2): D[i][j] = 128 + 32 for the (row / column, K block) that byte scales, 128 elsewhere.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int OSEL>
__global__ void k(float* out /*[2][64][4][256]*/) {
    const int l = threadIdx.x;
    i32x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = 0x38383838; b[j] = 0x38383838; }
    for (int which = 0; which < 2; ++which)
        for (int tl = 0; tl < 64; ++tl)
            for (int tb = 0; tb < 4; ++tb) {
                int s = 0x7f7f7f7f;
                if (l == tl) s = (s & ~(0xff << (8 * tb))) | (0x80 << (8 * tb));
                const int sa = which == 0 ? s : 0x7f7f7f7f, sb = which == 1 ? s : 0x7f7f7f7f;
                f32x4 c = {0.f, 0.f, 0.f, 0.f};
                c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, OSEL, sa, OSEL, sb);
                float* o = out + ((which * 64 + tl) * 4 + tb) * 256;
                for (int j = 0; j < 4; ++j) o[(4 * (l >> 4) + j) * 16 + (l & 15)] = c[j];
            }
}

int main() {
    const size_t n = 2 * 64 * 4 * 256;
    float* d; hipMalloc(&d, n * 4);
    std::vector<float> h(n);
    for (int osel = 0; osel < 4; ++osel) {
        hipMemset(d, 0, n * 4);
        if (osel == 0) hipLaunchKernelGGL(k<0>, dim3(1), dim3(64), 0, 0, d);
        if (osel == 1) hipLaunchKernelGGL(k<1>, dim3(1), dim3(64), 0, 0, d);
        if (osel == 2) hipLaunchKernelGGL(k<2>, dim3(1), dim3(64), 0, 0, d);
        if (osel == 3) hipLaunchKernelGGL(k<3>, dim3(1), dim3(64), 0, 0, d);
        hipMemcpy(h.data(), d, n * 4, hipMemcpyDeviceToHost);
        for (int which = 0; which < 2; ++which) {
            printf("OPSEL %d, %s scale register: (lane, byte) -> what it scaled\n", osel, which ? "B" : "A");
            for (int tl = 0; tl < 64; ++tl)
                for (int tb = 0; tb < 4; ++tb) {
                    const float* o = h.data() + ((which * 64 + tl) * 4 + tb) * 256;
                    // find affected rows/cols and the excess
                    int nrow = 0, ncol = 0, r0 = -1, c0 = -1; float ex = 0, base = 0;
                    for (int i = 0; i < 16; ++i) { bool any = false; for (int j = 0; j < 16; ++j) if (o[i * 16 + j] != 128.f) { any = true; ex = o[i * 16 + j] - 128.f; } if (any) { ++nrow; r0 = i; } }
                    for (int j = 0; j < 16; ++j) { bool any = false; for (int i = 0; i < 16; ++i) if (o[i * 16 + j] != 128.f) any = true; if (any) { ++ncol; c0 = j; } }
                    base = o[0];
                    if (nrow || ncol) printf("  lane %2d byte %d: rows %d (last %d) cols %d (last %d) excess %g\n", tl, tb, nrow, r0, ncol, c0, ex);
                    (void)base;
                }
        }
    }
    return 0;
}
