// What does v_mfma_f32_16x16x32_f16 compute, numerically?  (the error-bounded FIR on the matrix cores relies on it)
//   D[m][n] = C[m][n] + sum_{k<32} A[m][k] B[k][n],  f16 inputs, f32 accumulate
// Probes: f16 subnormal inputs; rounding (nearest / truncation); one rounding of the exact sum or a chain of rounded
// additions; then random dot products against double: max |D - exact| / (u (|C| + sum |a b|)), u = 2^-24.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

// one tile per wavefront: A[16][32], B[32][16] (f16 bits), C[16][16] -> D[16][16]
__global__ void k(const _Float16* A, const _Float16* B, const float* C, float* D, int tiles) {
    const int t = blockIdx.x; if (t >= tiles) return;
    const int l = threadIdx.x, m = l & 15, q = l >> 4;
    f16x8 a, b;
    for (int j = 0; j < 8; j++) { a[j] = A[(size_t) t * 512 + m * 32 + 8 * q + j]; b[j] = B[(size_t) t * 512 + (8 * q + j) * 16 + m]; }
    f32x4 c;
    for (int r = 0; r < 4; r++) c[r] = C[(size_t) t * 256 + (4 * q + r) * 16 + m];
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; r++) D[(size_t) t * 256 + (4 * q + r) * 16 + m] = c[r];
}

int main() {
    const int TILES = 4096;
    std::vector<_Float16> A((size_t) TILES * 512), B((size_t) TILES * 512);
    std::vector<float> C((size_t) TILES * 256), D((size_t) TILES * 256);
    std::mt19937_64 rng(1);
    auto set_row = [&](int t, int m, int n, const double* a, const double* b, double c) {       // row m of A, column n of B, C[m][n]
        for (int kk = 0; kk < 32; kk++) { A[(size_t) t * 512 + m * 32 + kk] = (_Float16) a[kk]; B[(size_t) t * 512 + kk * 16 + n] = (_Float16) b[kk]; }
        C[(size_t) t * 256 + m * 16 + n] = (float) c;
    };
    for (auto& v : A) v = (_Float16) 0.0f; for (auto& v : B) v = (_Float16) 0.0f; for (auto& v : C) v = 0.0f;
    // ---- probes in tile 0 (element [p][p] of the tile is probe p: row p of A against column p of B)
    double a[32], b[32];
    auto clear = [&]() { for (int i = 0; i < 32; i++) a[i] = b[i] = 0; };
    clear(); a[0] = ldexp(1.0, -20); b[0] = 1.0; set_row(0, 0, 0, a, b, 0.0);                       // 0: subnormal A
    clear(); a[0] = 1.0; b[0] = ldexp(1.0, -20); set_row(0, 1, 1, a, b, 0.0);                       // 1: subnormal B
    clear(); a[0] = ldexp(1.0, -20); b[0] = ldexp(1.0, -20); set_row(0, 2, 2, a, b, 0.0);           // 2: subnormal x subnormal = 2^-40
    clear(); for (int i = 0; i < 32; i++) { a[i] = ldexp(1.0, -13); b[i] = ldexp(1.0, -12); } set_row(0, 3, 3, a, b, 1.0);   // 3: 1 + 32 x 2^-25: chain of RN adds -> 1, exact -> 1 + 2^-20
    clear(); a[0] = ldexp(3.0, -13); b[0] = ldexp(1.0, -12); set_row(0, 4, 4, a, b, 1.0);           // 4: 1 + 0.75 ulp: nearest -> 1 + 2^-23, truncation -> 1
    clear(); a[0] = ldexp(1.0, -12); b[0] = ldexp(1.0, -12); set_row(0, 5, 5, a, b, 1.0);           // 5: 1 + 0.5 ulp (tie): RNE -> 1
    clear(); a[0] = ldexp(1.0, -12); b[0] = ldexp(1.0, -12); a[1] = ldexp(1.0, -20); b[1] = ldexp(1.0, -20); set_row(0, 6, 6, a, b, 1.0);   // 6: tie + 2^-40: exact sum rounds up; limited internal width -> 1
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = -1.0; b[1] = 1.0; a[2] = ldexp(1.0, -12); b[2] = ldexp(1.0, -13); set_row(0, 7, 7, a, b, 0.0);      // 7: cancellation: 1 - 1 + 2^-25 -> 2^-25 if products are summed exactly
    clear(); for (int i = 0; i < 32; i++) { a[i] = ldexp(1.0, -13); b[i] = ldexp(1.0, -12); } a[0] = 1.0; b[0] = 1.0; set_row(0, 8, 8, a, b, 0.0);  // 8: like 3 with the 1 as product 0 (C = 0)
    clear(); for (int i = 0; i < 32; i++) { a[i] = ldexp(1.0, -13); b[i] = ldexp(1.0, -12); } a[31] = 1.0; b[31] = 1.0; set_row(0, 9, 9, a, b, 0.0); // 9: the 1 as the LAST product
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = ldexp(1.0, -12); b[1] = ldexp(1.0, -18); set_row(0, 10, 10, a, b, 0.0);   // 10: 1 + 2^-30: how many bits below the largest product survive alignment? (then x 2^k below)
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = -1.0; b[1] = 1.0; a[2] = ldexp(1.0, -14); b[2] = ldexp(1.0, -14); set_row(0, 11, 11, a, b, 0.0);    // 11: 1 - 1 + 2^-28
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = -1.0; b[1] = 1.0; a[2] = ldexp(1.0, -14); b[2] = ldexp(1.0, -20); set_row(0, 12, 12, a, b, 0.0);    // 12: 1 - 1 + 2^-34
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = -1.0; b[1] = 1.0; a[2] = ldexp(1.0, -20); b[2] = ldexp(1.0, -24); set_row(0, 13, 13, a, b, 0.0);    // 13: 1 - 1 + 2^-44
    clear(); a[0] = 1.0; b[0] = 1.0; a[1] = ldexp(1.0, -14); b[1] = ldexp(1.0, -14); set_row(0, 14, 14, a, b, -1.0);  // 14: C = -1: 1 + 2^-28 - 1
    // ---- random tiles 1..: every element an independent dot product (A rows x B columns)
    std::uniform_real_distribution<double> U(-1.0, 1.0);
    for (int t = 1; t < TILES; t++) {
        const int mode = t % 4;       // 0: uniform, 1: wide exponent spread, 2: one huge term, 3: FIR-like (taps x samples) with C of partial-sum size
        for (int i = 0; i < 512; i++) {
            double va = U(rng), vb = U(rng);
            if (mode == 1) { va *= ldexp(1.0, -(int) (rng() % 12)); vb *= ldexp(1.0, -(int) (rng() % 12)); }
            if (mode == 2 && (rng() % 32) == 0) va *= 64.0;
            A[(size_t) t * 512 + i] = (_Float16) va; B[(size_t) t * 512 + i] = (_Float16) vb;
        }
        for (int i = 0; i < 256; i++) C[(size_t) t * 256 + i] = mode == 0 ? 0.0f : (float) (U(rng) * (mode == 3 ? 8.0 : 1.0));
    }
    _Float16 *dA, *dB; float *dC, *dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, B.size() * 2); hipMalloc(&dC, C.size() * 4); hipMalloc(&dD, D.size() * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), C.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(TILES), dim3(64), 0, 0, dA, dB, dC, dD, TILES);
    hipMemcpy(D.data(), dD, D.size() * 4, hipMemcpyDeviceToHost);
    const char* what[15] = {"subnormal A (2^-20 x 1)", "subnormal B", "subn x subn (2^-40)", "1 + 32 x 2^-25 (C = 1)", "1 + 0.75 ulp", "1 + 0.5 ulp (tie)", "tie + 2^-40",
                            "1 - 1 + 2^-25", "1 (k=0) + 31 x 2^-25", "31 x 2^-25 + 1 (k=31)", "1 + 2^-30", "1 - 1 + 2^-28", "1 - 1 + 2^-34", "1 - 1 + 2^-44", "C=-1: 1 + 2^-28 - 1"};
    for (int p = 0; p < 15; p++) {
        const float d = D[p * 16 + p];
        int e; const double mant = frexp((double) d, &e);
        printf("probe %2d %-26s D = %.10g  (= %.6f x 2^%d;  D - 1 = %.6g ulp(1))\n", p, what[p], d, mant, e, ((double) d - 1.0) / ldexp(1.0, -23));
    }
    double worst[4] = {0, 0, 0, 0}, worst_rel_exact[4] = {0, 0, 0, 0};
    for (int t = 1; t < TILES; t++) for (int m = 0; m < 16; m++) for (int n = 0; n < 16; n++) {
        double ex = C[(size_t) t * 256 + m * 16 + n], mag = fabs(ex);
        for (int kk = 0; kk < 32; kk++) { const double pr = (double) A[(size_t) t * 512 + m * 32 + kk] * (double) B[(size_t) t * 512 + kk * 16 + n]; ex += pr; mag += fabs(pr); }
        const double err = fabs((double) D[(size_t) t * 256 + m * 16 + n] - ex);
        const double r = err / (ldexp(1.0, -24) * mag);
        if (r > worst[t % 4]) worst[t % 4] = r;
        const double ulp = ldexp(1.0, ilogb(fabs(ex) > 1e-300 ? fabs(ex) : 1e-300) - 23);
        if (err / ulp > worst_rel_exact[t % 4]) worst_rel_exact[t % 4] = err / ulp;
    }
    for (int i = 0; i < 4; i++) printf("random mode %d: max |D - exact| = %.3f u (|C| + sum|ab|);  = %.3f ulp(exact result)\n", i, worst[i], worst_rel_exact[i]);
    return 0;
}
