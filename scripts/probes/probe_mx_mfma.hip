// Probe (diagnostic, not product): operand layout of v_mfma_scale_f32_16x16x128_f8f6f4 with e4m3 operands, and the rounding /
// saturation of the f32 -> e4m3 conversion instruction, checked with exact small-integer data against a host reference.
// Layout (first guess H - 32 consecutive K per lane - was wrong; found with probe_mx_pair / probe_mx_which): lane l holds row (A) /
// column (B) l & 15; its registers 0-3 hold K = 16 g .. 16 g + 15 and registers 4-7 hold K = 64 + 16 g .. 64 + 16 g + 15 (g = l >> 4):
// the 16-byte chunks g and g + 4 of a 128-byte K row, exactly the bf16 16x16x32 addressing.  Scale operand: byte OPSEL of lane
// (row, g)'s register is the E8M0 scale of K block g (K = 32 g .. 32 g + 31) of that row.  C/D: col = l & 15, row = 4 * (l >> 4) + reg.
//   hipcc --offload-arch=gfx950 -O2 probe_mx_mfma.hip -o probe_mx_mfma && ./probe_mx_mfma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ void k_mfma(const unsigned char* A /*[16][128]*/, const unsigned char* B /*[128][16] stored [col][k]*/,
                       const unsigned char* sa /*[16][4]*/, const unsigned char* sb /*[16][4]*/, float* D /*[16][16]*/, int opsel_mode) {
    const int l = threadIdx.x, r = l & 15, g = l >> 4;
    i32x8 a, b;
    // layout found by probe_mx_which / probe_mx_pair: registers 0-3 hold K = 16 g .. 16 g + 15, registers 4-7 hold
    // K = 64 + 16 g .. 64 + 16 g + 15; the lane's scale byte scales K block g = K 32 g .. 32 g + 31 of its row / column
    for (int j = 0; j < 4; ++j) {
        a[j] = *reinterpret_cast<const int*>(A + r * 128 + 16 * g + 4 * j);
        a[4 + j] = *reinterpret_cast<const int*>(A + r * 128 + 64 + 16 * g + 4 * j);
        b[j] = *reinterpret_cast<const int*>(B + r * 128 + 16 * g + 4 * j);
        b[4 + j] = *reinterpret_cast<const int*>(B + r * 128 + 64 + 16 * g + 4 * j);
    }
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    // opsel_mode = 4 * OPSEL + byte: the lane's scale sits in byte `byte` of the register (0x7f = 1.0 in the other bytes)
    const int byte = opsel_mode & 3, osel = opsel_mode >> 2;
    const int sca = (0x7f7f7f7f & ~(0xff << (8 * byte))) | (sa[r * 4 + g] << (8 * byte));
    const int scb = (0x7f7f7f7f & ~(0xff << (8 * byte))) | (sb[r * 4 + g] << (8 * byte));
    switch (osel) {
        case 0: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 0, sca, 0, scb); break;
        case 1: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 1, sca, 1, scb); break;
        case 2: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 2, sca, 2, scb); break;
        default: c = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, c, 0, 0, 3, sca, 3, scb); break;
    }
    for (int j = 0; j < 4; ++j) D[(4 * g + j) * 16 + r] = c[j];
}

__global__ void k_cvt(const float* x, unsigned char* q, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n + 1) return;
    const int w = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    q[2 * i] = w & 255; q[2 * i + 1] = (w >> 8) & 255;
}

static float e4m3_to_f(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f;
    if (e == 15 && m == 7) f = NAN;
    else if (e == 0) f = ldexpf((float)m, -9);
    else f = ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -f : f;
}
static unsigned char f_to_e4m3_rne_sat(float x) {     // host reference: RNE, saturate to +-448
    unsigned char best = 0; float bd = 1e30f;
    const float ax = fminf(fabsf(x), 448.0f);
    for (int v = 0; v < 127; ++v) {                  // 0x7f is NaN
        const float f = e4m3_to_f((unsigned char)v), d = fabsf(f - ax);
        if (d < bd || (d == bd && (v & 1) == 0)) { bd = d; best = (unsigned char)v; }
    }
    return best | (std::signbit(x) ? 0x80 : 0);
}

int main() {
    std::vector<unsigned char> A(16 * 128), B(16 * 128), sa(64), sb(64);
    const unsigned char vals[] = {0x00, 0x30, 0x38, 0x3C, 0x40, 0x44, 0xB0, 0xB8, 0xBC, 0xC0, 0xC4};   // 0, +-0.5, 1, 1.5, 2, 3
    srand(7);
    int bad_total = 0;
    for (int mode = 0; mode < 16; ++mode) {
        for (auto& v : A) v = vals[rand() % 11];
        for (auto& v : B) v = vals[rand() % 11];
        for (auto& v : sa) v = 125 + rand() % 5;
        for (auto& v : sb) v = 124 + rand() % 7;
        unsigned char *dA, *dB, *dsa, *dsb; float* dD;
        hipMalloc(&dA, A.size()); hipMalloc(&dB, B.size()); hipMalloc(&dsa, 64); hipMalloc(&dsb, 64); hipMalloc(&dD, 1024);
        hipMemcpy(dA, A.data(), A.size(), hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size(), hipMemcpyHostToDevice);
        hipMemcpy(dsa, sa.data(), 64, hipMemcpyHostToDevice); hipMemcpy(dsb, sb.data(), 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mfma, dim3(1), dim3(64), 0, 0, dA, dB, dsa, dsb, dD, mode);
        std::vector<float> D(256);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                double ref = 0;
                for (int k = 0; k < 128; ++k)
                    ref += (double)e4m3_to_f(A[i * 128 + k]) * ldexp(1.0, sa[i * 4 + k / 32] - 127) * (double)e4m3_to_f(B[j * 128 + k]) * ldexp(1.0, sb[j * 4 + k / 32] - 127);
                if (fabs(ref - D[i * 16 + j]) > 1e-6 * (1 + fabs(ref))) { ++bad; }
            }
        printf("MFMA layout hypothesis H, OPSEL %d, scale in byte %d: %d of 256 outputs differ\n", mode >> 2, mode & 3, bad);
        if ((mode >> 2) == (mode & 3)) bad_total += bad;
    }
    // conversion: every e4m3 value, midpoints between neighbours, values around the saturation point, subnormals
    std::vector<float> xs;
    for (int v = 0; v < 127; ++v) {
        const float f = e4m3_to_f((unsigned char)v), f2 = e4m3_to_f((unsigned char)(v + 1 < 127 ? v + 1 : v));
        xs.push_back(f); xs.push_back(-f); xs.push_back(0.5f * (f + f2)); xs.push_back(-0.5f * (f + f2));
        xs.push_back(nextafterf(0.5f * (f + f2), 1e9f)); xs.push_back(nextafterf(0.5f * (f + f2), -1e9f));
    }
    for (float f : {448.f, 449.f, 463.9f, 464.f, 464.1f, 480.f, 500.f, 511.9f, 512.f, 1000.f, 1e-4f, 9.765625e-4f, 1.5e-3f, 2e-3f}) { xs.push_back(f); xs.push_back(-f); }
    if (xs.size() & 1) xs.push_back(0.f);
    float* dx; unsigned char* dq;
    hipMalloc(&dx, xs.size() * 4); hipMalloc(&dq, xs.size());
    hipMemcpy(dx, xs.data(), xs.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_cvt, dim3((xs.size() / 2 + 63) / 64), dim3(64), 0, 0, dx, dq, (int)xs.size());
    std::vector<unsigned char> q(xs.size());
    hipMemcpy(q.data(), dq, xs.size(), hipMemcpyDeviceToHost);
    int cb = 0;
    for (size_t i = 0; i < xs.size(); ++i) {
        const unsigned char ref = f_to_e4m3_rne_sat(xs[i]);
        if (q[i] != ref && !(xs[i] == 0.f)) { if (cb < 12) printf("cvt(%.9g) = 0x%02x (%g), RNE-saturating reference 0x%02x (%g)\n", xs[i], q[i], e4m3_to_f(q[i]), ref, e4m3_to_f(ref)); ++cb; }
    }
    printf("v_cvt_pk_fp8_f32 vs RNE + saturate(448): %d of %zu differ\n", cb, xs.size());
    return bad_total ? 1 : 0;
}
