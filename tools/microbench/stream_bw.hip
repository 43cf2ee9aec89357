// What does a plain streaming kernel reach on this MI355X?  Read-only (sum into a sink) and copy, 16 bytes per lane, over 2 GiB:
// workgroups per grid x loads in flight per lane x {plain, non-temporal}.  Prints GB/s (copy: read + write bytes).
//   hipcc --offload-arch=gfx950 -O3 tools/microbench/stream_bw.hip -o tools/microbench/bin_stream_bw && tools/microbench/bin_stream_bw
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));

template <int U, bool NT, bool COPY>
__global__ __launch_bounds__(256) void k(const u4* __restrict__ src, u4* __restrict__ dst, size_t n16, u4* sink) {
    const size_t stride = (size_t) gridDim.x * blockDim.x;
    size_t i = blockIdx.x * (size_t) blockDim.x + threadIdx.x;
    u4 acc = { 0, 0, 0, 0 };
    for (; i + (U - 1) * stride < n16; i += U * stride) {
        u4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) v[u] = NT ? __builtin_nontemporal_load(src + i + u * stride) : src[i + u * stride];
#pragma unroll
        for (int u = 0; u < U; u++) {
            if (COPY) { if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride); else dst[i + u * stride] = v[u]; }
            else acc ^= v[u];
        }
    }
    if (!COPY && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[threadIdx.x] = acc;
}

template <int U, bool NT, bool COPY> void run(const u4* a, u4* b, size_t n16, u4* sink, unsigned grid) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; w++) hipLaunchKernelGGL((k<U, NT, COPY>), dim3(grid), dim3(256), 0, 0, a, b, n16, sink);
    hipEventRecord(e0, 0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL((k<U, NT, COPY>), dim3(grid), dim3(256), 0, 0, a, b, n16, sink);
    hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("%-5s U=%d %-5s grid=%6u  %7.0f GB/s\n", COPY ? "copy" : "read", U, NT ? "nt" : "plain", grid, 5.0 * (COPY ? 2.0 : 1.0) * n16 * 16 / (ms * 1e-3) / 1e9);
}

int main() {
    const size_t bytes = (size_t) 2 << 30, n16 = bytes / 16;
    u4 *a, *b, *sink;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes); hipMalloc(&sink, 4096 * 16);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    for (unsigned grid : { 1024u, 2048u, 4096u, 8192u, 16384u, 65536u }) {
        run<4, true, false>(a, b, n16, sink, grid); run<4, false, false>(a, b, n16, sink, grid);
        run<8, true, false>(a, b, n16, sink, grid); run<2, false, false>(a, b, n16, sink, grid);
        run<4, true, true>(a, b, n16, sink, grid); run<4, false, true>(a, b, n16, sink, grid);
        run<8, true, true>(a, b, n16, sink, grid); run<2, false, true>(a, b, n16, sink, grid); run<1, false, true>(a, b, n16, sink, grid);
    }
    hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, 0); for (int r = 0; r < 5; r++) hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); hipEventRecord(e1, 0); hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    printf("hipMemcpyAsync D2D %7.0f GB/s\n", 5.0 * 2.0 * bytes / (ms * 1e-3) / 1e9);
    return 0;
}
