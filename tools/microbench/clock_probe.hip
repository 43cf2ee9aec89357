// What shader clock does an MI355X hold under different kinds of load?  Every wavefront of a full-chip launch (256 CUs x 16 wavefronts)
// runs one kind of work for a fixed number of iterations; wavefront 0 of each workgroup reads the shader clock (s_memtime, clock64) and the
// 100 MHz wall clock (s_memrealtime) before and after: f = d(clock64) / d(wall) x 100 MHz.
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/clock_probe.hip -o tools/microbench/bin_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, const float* in, int iters, unsigned long long* stamps) {
    __shared__ float lds[4096];
    float v[8]; f2 p[8]; f4 acc[4] = { { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 }, { 0, 0, 0, 0 } };
    for (int j = 0; j < 8; j++) { v[j] = threadIdx.x * 1e-3f + j; p[j] = f2{ threadIdx.x * 1e-3f + j, 1.0f + j }; }
    for (int j = threadIdx.x; j < 4096; j += 256) lds[j] = j * 0.37f;
    h8 a, b; for (int j = 0; j < 8; j++) { a[j] = (_Float16) (0.01f * (threadIdx.x + j)); b[j] = (_Float16) (0.02f * (threadIdx.x ^ j)); }
    __syncthreads();
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    const float c = 1.0001f, d = 1e-3f;
    const unsigned la = (unsigned) (size_t) (__attribute__((address_space(3))) float*) lds + 16u * threadIdx.x;
    const float* gp = in + ((size_t) blockIdx.x * 256 + threadIdx.x) * 4;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int u = 0; u < 32; u++) {
            const int j = u & 7;
            if (KIND == 0) asm volatile("s_sleep 1");
            else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[j]) : "v"(c), "v"(d));
            else if (KIND == 2) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j]) : "v"(f2{ c, c }), "v"(f2{ d, d }));
            else if (KIND == 3) acc[u & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[u & 3], 0, 0, 0);
            else if (KIND == 4) { f4 t; asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(la)); asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t)); v[j] += t.x; }
            else if (KIND == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(v[j]) : "v"(threadIdx.x));
            else if (KIND == 6) { if (u < 4) { f4 t = *reinterpret_cast<const f4*>(gp + (size_t) ((i * 4 + u) & 1023) * 262144); v[j] += t.x; } }      // streaming loads (HBM)
            else if (KIND == 7) asm volatile("v_cvt_f16_f32 %0, %0" : "+v"(v[j]));
            else if (KIND == 8) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
            else if (KIND == 9) asm volatile("s_cmp_eq_u32 s20, 0x12345\n\ts_cbranch_scc1 1f\n\ts_add_u32 s20, s20, 1\n1:" ::: "s20", "scc");            // compare + branch not taken + add
            else if (KIND == 10) asm volatile("s_cmp_lg_u32 s20, 0x12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:\n\ts_add_u32 s20, s20, 1" ::: "s20", "scc");         // compare + branch TAKEN (over one instruction) + add
            else if (KIND == 11) asm volatile("s_cmp_eq_u32 s20, 0x12345\n\ts_add_u32 s20, s20, 1" ::: "s20", "scc");                                     // compare + add (no branch)
            else if (KIND == 12) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n\ts_and_b64 vcc, exec, vcc\n\ts_cbranch_vccz 1f\n\ts_nop 0\n1:" :: "v"(v[j]), "v"(1e30f) : "vcc");   // vector compare + uniform branch taken? (v < 1e30: gt false -> vccz taken)
            else if (KIND == 13) asm volatile("v_readlane_b32 s20, %0, 3\n\ts_add_u32 s21, s20, 1" :: "v"(v[j]) : "s20", "s21");
            else if (KIND == 14) asm volatile("v_readfirstlane_b32 s20, %0\n\tv_mov_b32 %0, s20" : "+v"(v[j]) :: "s20");
        }
    }
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    float r = 0; for (int j = 0; j < 8; j++) r += v[j] + p[j].x + p[j].y;
    for (int j = 0; j < 4; j++) r += acc[j][0] + acc[j][3];
    if (r == 12345.678f) out[threadIdx.x] = r;
    if (threadIdx.x == 0) { stamps[2 * blockIdx.x] = c1 - c0; stamps[2 * blockIdx.x + 1] = w1 - w0; }
}

static int g_wgs_per_cu = 4;
template <int KIND> void one(const char* name, float* out, const float* in, unsigned long long* st, int iters) {
    const int G = 256 * g_wgs_per_cu;                   // g_wgs_per_cu workgroups of 4 wavefronts per CU: that many wavefronts per SIMD
    hipLaunchKernelGGL((k<KIND>), dim3(G), dim3(256), 0, 0, out, in, iters, st);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND>), dim3(G), dim3(256), 0, 0, out, in, iters, st);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * G); hipMemcpy(h.data(), st, 16 * G, hipMemcpyDeviceToHost);
    double fs = 0, cyc = 0; for (int i = 0; i < G; i++) { fs += 100.0 * (double) h[2 * i] / (double) h[2 * i + 1]; cyc += (double) h[2 * i]; }
    printf("%-22s %8.3f ms  shader clock %7.1f MHz   cycles per instruction per wavefront %.2f\n", name, ms, fs / G, cyc / G / ((double) iters * 32));
}

int main(int argc, char** argv) {
    if (argc > 1) g_wgs_per_cu = atoi(argv[1]);
    printf("== %d wavefront(s) per SIMD\n", g_wgs_per_cu);
    float* out; float* in; unsigned long long* st;
    hipMalloc(&out, 4096 * 4); hipMalloc(&in, (size_t) 1024 * 262144 * 4 + (1 << 22)); hipMalloc(&st, 16 * 4096);
    hipMemset(in, 0, (size_t) 1024 * 262144 * 4 + (1 << 22));
    for (int rep = 0; rep < 2; rep++) {
        one<0>("s_sleep", out, in, st, 20000);
        one<1>("v_fma_f32", out, in, st, 40000);
        one<2>("v_pk_fma_f32", out, in, st, 40000);
        one<5>("v_add_u32", out, in, st, 40000);
        one<7>("v_cvt_f16_f32", out, in, st, 40000);
        one<3>("mfma 16x16x32 f16", out, in, st, 20000);
        one<4>("ds_read_b128", out, in, st, 10000);
        one<6>("global 16 B loads", out, in, st, 4000);
        one<8>("s_add_u32", out, in, st, 40000);
        one<11>("s_cmp + s_add", out, in, st, 40000);
        one<9>("s_cmp + cbranch(not taken) + s_add", out, in, st, 40000);
        one<10>("s_cmp + cbranch(taken) + s_add", out, in, st, 40000);
        one<12>("v_cmp + s_and + cbranch_vccz", out, in, st, 40000);
        one<13>("v_readlane + s_add", out, in, st, 40000);
        one<14>("v_readfirstlane + v_mov", out, in, st, 40000);
    }
    return 0;
}
