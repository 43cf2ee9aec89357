// Does v_mfma_f32_16x16x4_f32 of one wavefront overlap with the vector instructions of ANOTHER wavefront on the same SIMD?
// 512-thread workgroups = 8 waves = 2 per SIMD (waves w and w + 4 share a SIMD).  Waves 0..3 run MFMA chains, waves 4..7
// run a VALU stream; each role is timed alone and together (wall time of the whole launch, one workgroup per CU x 2).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/mfma_valu_overlap.hip -o /tmp/mvo && /tmp/mvo
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int MK = 0>
__global__ __launch_bounds__(512) void k(float* out, int iters, int do_mfma, int do_valu) {
    const int wave = threadIdx.x >> 6;
    float r = 0.0f;
    if (wave < 4) {
        if (do_mfma && MK == 1) {       // v_mfma_f32_16x16x32_f16
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            f16x8 x, y;
            for (int j = 0; j < 8; j++) { x[j] = (_Float16) (threadIdx.x * 1e-3f + j); y[j] = (_Float16) (1.0f + j * 1e-2f); }
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, y, a3, 0, 0, 0);
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        } else if (do_mfma) {
            f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
            float x = threadIdx.x * 1e-3f, y = 1.0f + threadIdx.x * 1e-4f;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
                    a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
                    a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
                    a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
                }
            }
            r = a0[0] + a1[1] + a2[2] + a3[3];
        }
    } else if (do_valu) {
        if (KIND == 0) {            // v_fma_f32, 8 independent chains
            float v[8]; for (int j = 0; j < 8; j++) v[j] = threadIdx.x * 1e-3f + j;
            const float c = 1.0001f, d = 1e-3f;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            }
            for (int j = 0; j < 8; j++) r += v[j];
        } else if (KIND == 1) {     // v_add_u32
            unsigned v[8]; for (int j = 0; j < 8; j++) v[j] = threadIdx.x + j;
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(threadIdx.x));
            }
            for (int j = 0; j < 8; j++) r += v[j];
        } else if (KIND == 2) {     // v_pk_fma_f32
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v[8]; for (int j = 0; j < 8; j++) v[j] = f2{threadIdx.x * 1e-3f + j, 1.0f};
            const f2 c = {1.0001f, 1.0002f}, d = {1e-3f, 2e-3f};
            for (int i = 0; i < iters; i++) {
#pragma unroll
                for (int u = 0; u < 8; u++)
#pragma unroll
                    for (int j = 0; j < 8; j++) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            }
            for (int j = 0; j < 8; j++) r += v[j].x + v[j].y;
        } else if (KIND == 3) {     // ds_read_b32 stream (LDS pipe)
            __shared__ float sh[4096];
            sh[threadIdx.x] = threadIdx.x; __syncthreads();      // (only waves 4..7 get here together? no: guard below)
        }
    }
    if (r == 12345.678f) out[blockIdx.x * 512 + threadIdx.x] = r;
}

template <int KIND, int MK = 0> void run(const char* name, float* out, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float t[3];
    const int modes[3][2] = {{1, 0}, {0, 1}, {1, 1}};
    for (int m = 0; m < 3; m++) {
        hipLaunchKernelGGL((k<KIND, MK>), dim3(512), dim3(512), 0, 0, out, iters / 8, modes[m][0], modes[m][1]);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, MK>), dim3(512), dim3(512), 0, 0, out, iters, modes[m][0], modes[m][1]);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&t[m], e0, e1);
    }
    // per SIMD: 2 workgroups per CU worth of work -> MFMA: iters * 16 per wave
    printf("%-14s mfma alone %.3f ms | valu alone %.3f ms | together %.3f ms  (sum %.3f, max %.3f)\n", name, t[0], t[1], t[2], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
}

int main() {
    float* out; hipMalloc(&out, 512 * 512 * 4);
    const int iters = 20000;
    run<0>("v_fma_f32", out, iters);
    run<1>("v_add_u32", out, iters);
    run<2>("v_pk_fma_f32", out, iters);
    run<0, 1>("f16: v_fma_f32", out, iters);
    run<1, 1>("f16: v_add_u32", out, iters);
    run<2, 1>("f16: v_pk_fma", out, iters);
    return 0;
}
