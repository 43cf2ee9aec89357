// How many independent vector instructions of the SAME wavefront hide behind one MFMA?  One wave per SIMD (256-thread
// workgroups, one per CU); a loop of 16 MFMAs (4 independent accumulators) with F fillers after each.
//   prints shader cycles per MFMA for F = 0..8, for the f32 (16x16x4) and f16 (16x16x32) forms and three filler kinds
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int MK, int F, int KIND>
__global__ __launch_bounds__(256) void k(float* out, int iters, long long* cycles) {
    f32x4 a[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
    f16x8 hx, hy;
    for (int j = 0; j < 8; j++) { hx[j] = (_Float16) (threadIdx.x * 1e-3f + j); hy[j] = (_Float16) (1.0f + j * 1e-2f); }
    float v[8]; for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 1e-3f + j;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 pv[8]; for (int j = 0; j < 8; j++) pv[j] = f2{threadIdx.x * 1e-3f + j, 1.0f};
    const float c = 1.0001f, d = 1e-3f; const f2 pc = {1.0001f, 1.0002f}, pd = {1e-3f, 2e-3f};
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 16; u++) {
            if (MK == 0) a[u & 3] = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a[u & 3], 0, 0, 0);
            else if (MK == 1) a[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(hx, hy, a[u & 3], 0, 0, 0);
#pragma unroll
            for (int f = 0; f < F; f++) {
                if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[(u * F + f) & 7]) : "v"(c), "v"(d));
                else if (KIND == 1) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[(u * F + f) & 7]) : "v"(threadIdx.x));
                else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(pv[(u * F + f) & 7]) : "v"(pc), "v"(pd));
            }
        }
    }
    const long long t1 = clock64();
    float r = a[0][0] + a[1][1] + a[2][2] + a[3][3];
    for (int j = 0; j < 8; j++) r += v[j] + pv[j].x;
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cycles = t1 - t0;
}

template <int MK, int F, int KIND> void one(float* out, long long* cyc, int iters) {
    hipLaunchKernelGGL((k<MK, F, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MK, F, KIND>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long h; hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf(" F=%d: %.1f clk64 / %.2f ns", F, (double) h / (iters * 16.0), ms * 1e6 / (iters * 16.0));
}
template <int MK, int KIND> void row(const char* name, float* out, long long* cyc) {
    printf("%-28s", name);
    const int iters = 20000;
    one<MK, 0, KIND>(out, cyc, iters); one<MK, 1, KIND>(out, cyc, iters); one<MK, 2, KIND>(out, cyc, iters); one<MK, 3, KIND>(out, cyc, iters);
    one<MK, 4, KIND>(out, cyc, iters); one<MK, 6, KIND>(out, cyc, iters); one<MK, 8, KIND>(out, cyc, iters);
    printf("\n");
}
int main() {
    float* out; hipMalloc(&out, 4096); long long* cyc; hipMalloc(&cyc, 8);
    row<2, 0>("no mfma + v_fma_f32", out, cyc);
    row<0, 0>("mfma f32 16x16x4 + v_fma_f32", out, cyc);
    row<0, 1>("mfma f32 16x16x4 + v_add_u32", out, cyc);
    row<0, 2>("mfma f32 16x16x4 + v_pk_fma", out, cyc);
    row<1, 0>("mfma f16 16x16x32 + v_fma_f32", out, cyc);
    row<1, 1>("mfma f16 16x16x32 + v_add_u32", out, cyc);
    row<1, 2>("mfma f16 16x16x32 + v_pk_fma", out, cyc);
    return 0;
}
