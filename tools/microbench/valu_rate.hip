// Issue cost of vector instructions on one SIMD of an MI355X, by instruction kind and by the number of wavefronts sharing the SIMD.
// One workgroup per CU of W x 4 wavefronts (so W per SIMD); every wavefront runs a loop of 64 INDEPENDENT instructions of one kind
// (8 chains, 8 deep per trip); prints SIMD cycles per instruction = cycles of the slowest wavefront x 1 / (instructions per wave x W).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/valu_rate.hip -o tools/microbench/bin_valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(1024) void k(float* out, int iters, long long* cycles) {
    float v[8]; f2 p[8];
    for (int j = 0; j < 8; j++) { v[j] = threadIdx.x * 1e-3f + j; p[j] = f2{ threadIdx.x * 1e-3f + j, 1.0f + j }; }
    const float c = 1.0001f, d = 1e-3f; const f2 pc = { 1.0001f, 1.0002f }, pd = { 1e-3f, 2e-3f };
    asm volatile("s_mov_b64 s[22:23], 0x5555" ::: "s22", "s23");
    asm volatile("s_mov_b64 vcc, 0x3333" ::: "vcc");
    __syncthreads();
    const long long t0 = clock64();
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 64; u++) {
            const int j = u & 7;
            if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(d));
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pd));
            else if (KIND == 3) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(pc), "v"(pd));
            else if (KIND == 4) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[j]) : "v"(pc));
            else if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(threadIdx.x));
            else if (KIND == 6) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[j]));
            else if (KIND == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[j]) : "v"(d));
            else if (KIND == 8) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 9) asm volatile("v_min_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(v[j]));
            else if (KIND == 10) asm volatile("v_cmp_gt_f32 vcc, %0, %1" :: "v"(v[j]), "v"(d) : "vcc");
            else if (KIND == 11) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(v[j]) : "v"(threadIdx.x));
            else if (KIND == 12) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(v[j]) : "v"(d));
            else if (KIND == 13) asm volatile("v_readlane_b32 s20, %0, 3" :: "v"(v[j]) : "s20");
            else if (KIND == 14) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
            else if (KIND == 15) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[22:23]" : "+v"(v[j]) : "v"(d));
            else if (KIND == 16) asm volatile("v_mov_b32 %0, %1" : "=v"(v[j]) : "v"(d));
            else if (KIND == 17) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[j]) : "v"(c));
            else if (KIND == 18) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[j]) : "v"(threadIdx.x));
            else if (KIND == 19) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[j]));
            else if (KIND == 20) asm volatile("v_cvt_f32_f16 %0, %0" : "+v"(v[j]));
            else if (KIND == 21) asm volatile("v_fma_mixlo_f16 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 22) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[j]) : "v"(c));
            else if (KIND == 23) asm volatile("v_max_f32 %0, %0, %1" : "+v"(v[j]) : "v"(d));
            else if (KIND == 24) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 25) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 26) asm volatile("v_add_f32_e64 %0, |%0|, %1" : "+v"(v[j]) : "v"(d));
            else if (KIND == 27) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[j]) : "v"(d));
            else if (KIND == 28) asm volatile("v_bfe_u32 %0, %0, 3, 5" : "+v"(v[j]));
            else if (KIND == 29) asm volatile("v_cmp_gt_f32_e64 s[24:25], %0, %1" :: "v"(v[j]), "v"(d) : "s24", "s25");
            else if (KIND == 30) asm volatile("v_mov_b32_dpp %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(v[j]));
            else if (KIND == 31) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[j]) : "v"(threadIdx.x));
        }
    }
    const long long t1 = clock64();
    float r = 0; for (int j = 0; j < 8; j++) r += v[j] + p[j].x + p[j].y;
    if (r == 12345.678f) out[threadIdx.x] = r;
    if ((threadIdx.x & 63) == 0) atomicMax((unsigned long long*) cycles, (unsigned long long) (t1 - t0));
}

template <int KIND> void one(const char* name, float* out, long long* cyc) {
    printf("%-18s", name);
    for (int W : { 1, 2, 4 }) {
        const int iters = 2000;
        hipMemset(cyc, 0, 8);
        hipLaunchKernelGGL((k<KIND>), dim3(256), dim3(256 * W), 0, 0, out, iters, cyc);
        hipDeviceSynchronize();
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        printf("  W=%d: %6.2f cyc/instr/SIMD", W, (double) c / ((double) iters * 64 * W));
    }
    printf("\n");
}

int main() {
    float* out; long long* cyc; hipMalloc(&out, 4096 * 4); hipMalloc(&cyc, 8);
    one<0>("v_add_f32", out, cyc); one<1>("v_fma_f32", out, cyc); one<2>("v_pk_add_f32", out, cyc); one<3>("v_pk_fma_f32", out, cyc);
    one<4>("v_pk_mul_f32", out, cyc); one<5>("v_add_u32", out, cyc); one<6>("v_cvt_f16_f32", out, cyc); one<7>("v_cndmask_b32", out, cyc);
    one<8>("v_max3_f32", out, cyc); one<9>("v_min_f32_dpp", out, cyc); one<10>("v_cmp_gt_f32", out, cyc); one<11>("v_lshl_add_u32", out, cyc);
    one<12>("v_cvt_pk_f16_f32", out, cyc); one<13>("v_readlane_b32", out, cyc); one<14>("s_add_u32", out, cyc);
    one<15>("v_cndmask sgpr", out, cyc); one<16>("v_mov_b32", out, cyc); one<17>("v_mul_f32", out, cyc); one<18>("v_and_b32", out, cyc);
    one<19>("v_lshlrev_b32", out, cyc); one<20>("v_cvt_f32_f16", out, cyc); one<21>("v_fma_mixlo_f16", out, cyc); one<22>("v_pk_mul_f16", out, cyc);
    one<23>("v_max_f32", out, cyc); one<24>("v_fmac_f32", out, cyc); one<25>("v_mad_u32_u24", out, cyc); one<26>("v_add_f32_e64 abs", out, cyc);
    one<27>("v_sub_f32", out, cyc); one<28>("v_bfe_u32", out, cyc); one<29>("v_cmp_e64 sgpr", out, cyc); one<30>("v_mov_dpp quad", out, cyc); one<31>("v_xor_b32", out, cyc);
    return 0;
}
