// Second probe set for v_mfma_f32_16x16x32_f16: where are small products cut off next to large ones?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D) {
    const int t = blockIdx.x, l = threadIdx.x, m = l & 15, q = l >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = A[(size_t) t * 512 + m * 32 + 8 * q + j]; b[j] = B[(size_t) t * 512 + (8 * q + j) * 16 + m]; }
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[(size_t) t * 256 + (4 * q + r) * 16 + m];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(size_t) t * 256 + (4 * q + r) * 16 + m] = c[r];
}
struct Probe { char name[64]; double a[32], b[32], c; };
int main() {
    std::vector<Probe> P;
    auto add = [&](const char* fmt, int t, auto fill) { Probe p; for (int i = 0; i < 32; i++) p.a[i] = p.b[i] = 0; p.c = 0; snprintf(p.name, 64, fmt, t); fill(p); P.push_back(p); };
    auto split = [](double v, double& a, double& b) { int e; double m = frexp(v, &e); int ea = e / 2, eb = e - ea; a = ldexp(m, ea); b = ldexp(1.0, eb); };   // v = a * b, both f16-representable for moderate e
    for (int t = 20; t <= 28; t++) {
        add("same group: 1 - 1 + 2^-%d", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; p.a[1] = -1; p.b[1] = 1; split(ldexp(1.0, -t), p.a[2], p.b[2]); });
        add("other group: 1 - 1 + 2^-%d", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; p.a[8] = -1; p.b[8] = 1; split(ldexp(1.0, -t), p.a[16], p.b[16]); });
        add("other group(2): 1 + 2^-%d - 1", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; split(ldexp(1.0, -t), p.a[8], p.b[8]); p.a[16] = -1; p.b[16] = 1; });
        add("C = 1, products -1 + 2^-%d", t, [&](Probe& p) { p.c = 1; p.a[0] = -1; p.b[0] = 1; split(ldexp(1.0, -t), p.a[1], p.b[1]); });
        add("C = 1, products -1 | 2^-%d", t, [&](Probe& p) { p.c = 1; p.a[0] = -1; p.b[0] = 1; split(ldexp(1.0, -t), p.a[8], p.b[8]); });
        add("C = 2^-%d, products 1 - 1", t, [&](Probe& p) { p.c = ldexp(1.0, -t); p.a[0] = 1; p.b[0] = 1; p.a[1] = -1; p.b[1] = 1; });
        add("C = 2^-%d, products 1 | -1", t, [&](Probe& p) { p.c = ldexp(1.0, -t); p.a[0] = 1; p.b[0] = 1; p.a[8] = -1; p.b[8] = 1; });
    }
    // seven small terms next to a 1 in the same group, each (2^-23 - 2^-33): truncation loses almost 7 x 2^-23?
    for (int t = 22; t <= 26; t++) add("1 + 7 x 1.9990 x 2^-%d, same group", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; for (int i = 1; i < 8; i++) { p.a[i] = ldexp(2047.0 / 1024.0, -12); p.b[i] = ldexp(1.0, -(t - 12)); } });
    // is the cut a truncation or a rounding?  1 - 1 + 1.5 x 2^-t
    for (int t = 22; t <= 26; t++) add("1 - 1 + 1.5 x 2^-%d same group", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; p.a[1] = -1; p.b[1] = 1; p.a[2] = ldexp(1.5, -12); p.b[2] = ldexp(1.0, -(t - 12)); });
    for (int t = 22; t <= 26; t++) add("1 - 1 - 1.5 x 2^-%d same group", t, [&](Probe& p) { p.a[0] = 1; p.b[0] = 1; p.a[1] = -1; p.b[1] = 1; p.a[2] = -ldexp(1.5, -12); p.b[2] = ldexp(1.0, -(t - 12)); });
    const int T = (int) P.size();
    std::vector<_Float16> A((size_t) T * 512, (_Float16) 0.0f), B((size_t) T * 512, (_Float16) 0.0f); std::vector<float> C((size_t) T * 256, 0.0f), D((size_t) T * 256);
    for (int t = 0; t < T; t++) { for (int kk = 0; kk < 32; kk++) { A[(size_t) t * 512 + kk] = (_Float16) P[t].a[kk]; B[(size_t) t * 512 + kk * 16] = (_Float16) P[t].b[kk]; } C[(size_t) t * 256] = (float) P[t].c; }
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(T), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    for (int t = 0; t < T; t++) {
        double ex = P[t].c; for (int kk = 0; kk < 32; kk++) ex += (double) (_Float16) P[t].a[kk] * (double) (_Float16) P[t].b[kk];
        printf("%-40s D = %-14.8g exact = %-14.8g  D/2^-24 = %.4f\n", P[t].name, D[(size_t) t * 256], ex, D[(size_t) t * 256] / ldexp(1.0, -24));
    }
    return 0;
}
